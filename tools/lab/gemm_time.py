"""Time ops.linear on a few shapes (one process per library variant: VG_KERNELS_SO).  Best of N interleaved rounds, us."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videoglamm_amd import ops
SH = {"o": (3361, 4096, 4096, False), "gateup": (3361, 14336, 4096, True), "down": (3361, 4096, 14336, False), "8k": (8192, 8192, 8192, False),
      "4k": (4096, 4096, 4096, False), "s4fc1": (16384, 4608, 1152, False), "iv2fc1": (4100, 6144, 1408, False), "qkv": (3361, 6144, 4096, False),
      "clipfc2": (9232, 1024, 4096, False), "iv2fc2": (4100, 1408, 6144, False), "s4fc2": (16384, 1152, 4608, False),
      "s3fc1": (65536, 2304, 576, False), "s3qkv": (65536, 1728, 576, False), "s3proj": (65536, 576, 576, False), "s3fc2": (65536, 576, 2304, False)}
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
names = (os.environ.get("SHAPES") or "o,gateup,down,8k,4k,s4fc1,iv2fc1").split(",")
out = []
for nm in names:
    M, N, K, glu = SH[nm]
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn((2 if glu else 1) * N, K, device="cuda", dtype=torch.bfloat16)
    best = min(t(lambda: ops.linear(a, w, glu=glu)) for _ in range(3))
    out.append(f"{nm} {best:7.1f} ({2*M*N*(2 if glu else 1)*K/best/1e6:6.0f} TF)")
print(os.environ.get("TAG", ""), os.environ.get("VG_KERNELS_SO", "default").split("/")[-1], " | ".join(out), flush=True)
