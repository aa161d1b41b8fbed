"""vg_mlp_rows against the three launches it replaces, Hiera stage 1 / 2 shapes of a 16-frame chunk"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoglamm_amd import ops  # noqa: E402


def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


dev = "cuda"
for M, C in ((1048576, 144), (262144, 288)):
    H = 4 * C
    x = torch.randn(M, C, device=dev, dtype=torch.bfloat16)
    ln = (torch.ones(C, device=dev), torch.zeros(C, device=dev), 1e-6)
    w1, b1 = (torch.randn(H, C, device=dev) * C ** -0.5).bfloat16(), torch.zeros(H, device=dev)
    w2, b2 = (torch.randn(C, H, device=dev) * H ** -0.5).bfloat16(), torch.zeros(C, device=dev)
    fused = t(lambda: ops.mlp_rows(x, ln, w1, b1, w2, b2))
    n_ = t(lambda: ops.layernorm(x, ln[0], ln[1], ln[2]))
    xn = ops.layernorm(x, ln[0], ln[1], ln[2])
    f1 = t(lambda: ops.linear(xn, w1, b1, ops.ACT_GELU))
    h = ops.linear(xn, w1, b1, ops.ACT_GELU)
    f2 = t(lambda: ops.linear(h, w2, b2, residual=x))
    fl = 4.0 * M * C * H
    print(f"M={M} C={C}: fused {fused:.1f} us ({fl / fused / 1e6:.0f} TF/s) | separate norm {n_:.1f} + fc1 {f1:.1f} + fc2 {f2:.1f} = {n_ + f1 + f2:.1f} us")
