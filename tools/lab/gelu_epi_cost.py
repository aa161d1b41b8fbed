"""what the exact-erf GELU costs inside the GEMM epilogues: the same shape with act none / GELU / ReLU (Hiera stage 3 / 4 fc1, InternVideo2 fc1, stage 2 fc1)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoglamm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, M, N, K in [("hiera s3 fc1", 65536, 2304, 576), ("hiera s4 fc1", 16384, 4608, 1152), ("iv2 fc1", 4100, 6144, 1408), ("hiera s2 fc1", 262144, 1152, 288)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16) * 0.3
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * K ** -0.5
    b = torch.randn(N, device=dev, dtype=torch.float32)
    r = {k: t(lambda: ops.linear(a, w, b, act=v)) for k, v in (("none", ops.ACT_NONE), ("gelu", ops.ACT_GELU), ("relu", ops.ACT_RELU), ("quick", ops.ACT_QUICK_GELU))}
    print(f"{name:14s} M={M:7d} N={N:5d} K={K:5d}: " + "  ".join(f"{k} {v:7.1f} us" for k, v in r.items()) + f"   GELU costs {r['gelu'] - r['none']:6.1f} us ({M * N / 1e6:.0f} M elements)", flush=True)
