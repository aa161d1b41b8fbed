"""row-register kernel (vg_gemm_rr.hip) on Hiera's stage 1-2 shapes.  Run once per VG_GEMM_RR setting (0 = the tile kernels, 1 = the rule)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoglamm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [("s1 qkv", 1048576, 432, 144, 0, 0), ("s1 proj+res", 1048576, 144, 144, 0, 1), ("s1->s2 proj", 1048576, 288, 144, 0, 0), ("s2b0 qkv", 1048576, 864, 144, 0, 0),
          ("s2 qkv", 262144, 864, 288, 0, 0), ("s2 proj+res", 262144, 288, 288, 0, 1), ("s2 fc1 gelu", 262144, 1152, 288, 1, 0), ("s2 fc1 none", 262144, 1152, 288, 0, 0),
          ("fpn l0", 1048576, 256, 144, 0, 0), ("fpn l1", 262144, 256, 288, 0, 0)]
knob = os.environ.get("VG_GEMM_RR", "1")
for name, M, N, K, act, res in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16) * 0.1
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.1
    b = torch.randn(N, device=dev, dtype=torch.float32)
    r = torch.randn(M, N, device=dev, dtype=torch.bfloat16) if res else None
    us = t(lambda: ops.linear(a, w, b, act=act, residual=r))
    gb = (M * K + M * N * (2 if res else 1)) * 2 / 1e9
    print(f"RR={knob} {name:12s} M={M:8d} N={N:5d} K={K:4d}: {us:7.1f} us = {2.0 * M * N * K / us / 1e6:7.1f} TF/s, {gb / us * 1e3:5.2f} TB/s (HBM roof {gb / 8e3 * 1e6:6.1f} us)", flush=True)
    del a, w, r
