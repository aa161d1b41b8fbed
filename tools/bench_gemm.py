"""GEMM / attention micro-benchmarks on the shapes of the C1 workload (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoglamm_amd import ops

def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

shapes = [("llm qkv/o", 1697, 4096, 4096), ("llm gate|up", 1697, 28672, 4096), ("llm down", 1697, 4096, 14336),
          ("iv2 qkv", 2050, 4224, 1408), ("iv2 fc1", 2050, 6144, 1408), ("iv2 fc2", 2050, 1408, 6144),
          ("clip qkv", 4616, 3072, 1024), ("clip fc1", 4616, 4096, 1024), ("clip fc2", 4616, 1024, 4096),
          ("hiera s1 qkv", 524288, 432, 144), ("hiera s1 fc1", 524288, 576, 144), ("hiera s2 qkv", 131072, 864, 288),
          ("hiera s3 qkv", 32768, 1728, 576), ("hiera s3 fc1", 32768, 2304, 576), ("hiera s3 fc2", 32768, 576, 2304),
          ("hiera s4 fc1", 8192, 4608, 1152), ("hiera s3 proj", 32768, 576, 576), ("hiera s4 qkv", 8192, 3456, 1152), ("hiera s4 fc2", 8192, 1152, 4608),
          ("hiera s4 proj", 8192, 1152, 1152), ("hiera s2 fc1", 131072, 1152, 288), ("hiera s2 fc2", 131072, 288, 1152), ("hiera s1 fc2", 524288, 144, 576),
          ("hiera s2 proj", 131072, 288, 288), ("square 4k", 4096, 4096, 4096), ("square 8k", 8192, 8192, 8192),
          ("sp8 qkv", 213, 6144, 4096), ("sp8 o", 213, 4096, 4096), ("sp8 gate|up", 213, 28672, 4096), ("sp8 down", 213, 4096, 14336),
          ("sp4 gate|up", 425, 28672, 4096), ("sp4 down", 425, 4096, 14336),
          ("sp2 o", 849, 4096, 4096), ("sp2 down", 849, 4096, 14336), ("sp2 qkv", 849, 6144, 4096),
          ("c2 s1 qkv", 1048576, 432, 144), ("c2 s1 fc1", 1048576, 576, 144), ("c2 s1 proj", 1048576, 144, 144), ("c2 s2 qkv", 262144, 864, 288),
          ("c2 s2 fc1", 262144, 1152, 288), ("c2 s2 proj", 262144, 288, 288), ("c2 patch", 1048576, 144, 152),
          ("c2 llm qkv", 3361, 6144, 4096), ("c2 llm o", 3361, 4096, 4096), ("c2 llm gate|up", 3361, 28672, 4096), ("c2 llm down", 3361, 4096, 14336)]
if os.environ.get("VG_BENCH_SHAPES"):
    shapes = [s for s in shapes if any(t in s[0] for t in os.environ["VG_BENCH_SHAPES"].split(","))]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    if os.environ.get("VG_BENCH_ACT"):      # the epilogue of Hiera's fc1 (bias + exact GELU) / qkv (bias)
        bias = torch.randn(N, device="cuda", dtype=torch.float32)
        act = ops.ACT_GELU if os.environ["VG_BENCH_ACT"] == "gelu" else ops.ACT_NONE
        ms = t(lambda: ops.linear(a, w, bias, act, out=out))
    else:
        ms = t(lambda: ops.linear(a, w, out=out))
    if os.environ.get("VG_BENCH_GEMM_ONLY"):
        print(f"gemm {name:14s} M={M:7d} N={N:6d} K={K:6d}  {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF/s  {(M*K+N*K+M*N)*2/ms/1e9:7.2f} TB/s algorithmic")
        continue
    print(f"gemm {name:14s} M={M:7d} N={N:6d} K={K:6d}  {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF/s")
if os.environ.get("VG_BENCH_GEMM_ONLY"):
    sys.exit(0)
if os.environ.get('VG_BENCH_ONLY') in ('gemm', 'f8'):
    for name, M, N, K, glu in [("llm qkv", 1697, 6144, 4096, False), ("llm o", 1697, 4096, 4096, False), ("llm gate|up+glu", 1697, 14336, 4096, True),
                               ("llm down", 1697, 4096, 14336, False), ("c2 gate|up+glu", 3361, 14336, 4096, True), ("c2 down", 3361, 4096, 14336, False),
                               ("square 8k", 8192, 8192, 8192, False)]:
        a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        w = torch.randn((2 * N if glu else N), K, device="cuda", dtype=torch.bfloat16)
        q, qs = ops.quantize_fp8(a)
        w8, ws = ops.quantize_fp8_rows(w)
        ms = t(lambda: ops.linear_f8(q, qs, w8, ws, glu=glu))
        mq = t(lambda: ops.quantize_fp8(a))
        mb = t(lambda: ops.linear(a, w, glu=glu))
        fl = 2 * M * (2 * N if glu else N) * K
        print(f"f8 gemm {name:16s} M={M:5d} N={N:6d} K={K:6d}  fp8 {ms*1e3:8.1f} us {fl/ms/1e9:8.1f} TF/s | bf16 {mb*1e3:8.1f} us {fl/mb/1e9:8.1f} TF/s | quantise A {mq*1e3:6.1f} us")
    sys.exit(0)
for name, B, H, Hkv, Sq, Skv, D, causal in [("llm prefill", 1, 32, 8, 1697, 1697, 128, True), ("llm decode", 1, 32, 8, 1, 1730, 128, True),
                                            ("iv2", 2, 16, 16, 1025, 1025, 88, False), ("clip", 8, 16, 16, 577, 577, 64, False),
                                            ("hiera win8", 8192, 2, 2, 64, 64, 72, False), ("hiera glob", 8, 8, 8, 4096, 4096, 72, False),
                                            ("memattn cross", 1, 1, 1, 4096, 28736, 256, False), ("dec tok->img", 8, 8, 8, 7, 4096, 16, False),
                                            ("dec img->tok", 8, 8, 8, 4096, 7, 16, False)]:
    q = torch.randn(B, Sq, H, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, Skv, Hkv, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, Skv, Hkv, D, device="cuda", dtype=torch.bfloat16)
    ms = t(lambda: ops.attention(q, k, v, D ** -0.5, causal))
    fl = 4.0 * B * H * Sq * Skv * D * (0.5 if causal and Sq == Skv else 1.0)
    print(f"attn {name:14s} B={B} H={H} Sq={Sq} Skv={Skv} D={D}  {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TF/s")
for rows, C in [(1, 4096), (1697, 4096), (2050, 1408), (524288, 144)]:
    x = torch.randn(rows, C, device="cuda", dtype=torch.bfloat16); w = torch.ones(C, device="cuda")
    ms = t(lambda: ops.rmsnorm(x, w, 1e-5))
    print(f"rmsnorm rows={rows} C={C} {ms*1e3:8.1f} us  {rows*C*4/ms/1e6:8.1f} GB/s")
for N, K in [(4096, 4096), (28672, 4096), (4096, 14336), (128257, 4096)]:
    a = torch.randn(1, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    ms = t(lambda: ops.linear(a, w), 20)
    print(f"gemv N={N} K={K} {ms*1e3:8.1f} us  {N*K*2/ms/1e6:8.1f} GB/s")
