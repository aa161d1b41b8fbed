"""Ceiling check: our GEMM kernels against the vendor library (torch -> hipBLASLt / rocBLAS) on the C2 shapes.  Diagnostic only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from videoglamm_amd import ops
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [("c2 llm qkv", 3361, 6144, 4096), ("c2 llm o", 3361, 4096, 4096), ("c2 llm gate|up", 3361, 28672, 4096), ("c2 llm down", 3361, 4096, 14336),
          ("iv2 qkv", 4100, 4224, 1408), ("iv2 fc1", 4100, 6144, 1408), ("iv2 fc2", 4100, 1408, 6144),
          ("clip fc1", 9232, 4096, 1024), ("clip fc2", 9232, 1024, 4096),
          ("hiera s1 qkv", 1048576, 432, 144), ("hiera s1 fc1", 1048576, 576, 144), ("hiera s1 fc2", 1048576, 144, 576),
          ("hiera s2 qkv", 262144, 864, 288), ("hiera s2 fc1", 262144, 1152, 288), ("hiera s2 fc2", 262144, 288, 1152),
          ("hiera s3 qkv", 65536, 1728, 576), ("hiera s3 proj", 65536, 576, 576), ("hiera s3 fc1", 65536, 2304, 576), ("hiera s3 fc2", 65536, 576, 2304),
          ("hiera s4 qkv", 16384, 3456, 1152), ("hiera s4 proj", 16384, 1152, 1152), ("hiera s4 fc1", 16384, 4608, 1152), ("hiera s4 fc2", 16384, 1152, 4608),
          ("clip proj", 9232, 1024, 1024), ("clip qkv", 9232, 3072, 1024), ("iv2 proj", 4100, 1408, 1408),
          ("square 8k", 8192, 8192, 8192)]
if os.environ.get("VG_BENCH_SHAPES"):
    shapes = [s for s in shapes if any(t in s[0] for t in os.environ["VG_BENCH_SHAPES"].split(","))]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    # interleaved rounds, best of each (box drift between two single runs is larger than most kernel effects)
    ms, ml = 1e9, 1e9
    for _ in range(int(os.environ.get("VG_BENCH_ROUNDS", "3"))):
        ms = min(ms, t(lambda: ops.linear(a, w, out=out)))
        ml = min(ml, t(lambda: F.linear(a, w)))
    fl = 2 * M * N * K
    from videoglamm_amd import _lib
    name = f"{name} [r{_lib.load().vg_gemm_route(M, N, K, 1, 0, 0)}]"
    print(f"{name:19s} M={M:8d} N={N:6d} K={K:6d}  ours {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF/s | lib {ml*1e3:8.1f} us {fl/ml/1e9:7.1f} TF/s  ratio {ml/ms:.2f}", flush=True)
