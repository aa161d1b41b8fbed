"""vg_gemm_rows (LayerNorm prologue, optional RoPE epilogue) against the separate launches, by row count: where does the fused short-row kernel stop paying?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoglamm_amd import ops  # noqa: E402


def t(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


dev, K = "cuda", 256
ang = torch.randn(4096, 128, device=dev)
cos, sin = ang.cos(), ang.sin()
ln = (torch.ones(K, device=dev), torch.zeros(K, device=dev), 1e-5)
for B in (1, 2, 4, 8):
    M = B * 4096
    x = torch.randn(B, 4096, K, device=dev, dtype=torch.bfloat16)
    for N, act, rope in ((2048, ops.ACT_RELU, None), (1024, ops.ACT_GELU, None), (768, 0, (cos, sin, 512, 256, 4096, 0, 4096, 4096)), (256, 0, (cos, sin, 256, 256, 4096, 0, 4096, 4096))):
        w, b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16(), torch.zeros(N, device=dev)
        fused = t(lambda: ops.linear_rows(x, w, b, act, ln=ln, rope=rope))

        def sep():
            y = ops.linear(ops.layernorm(x, ln[0], ln[1], ln[2]), w, b, act)
            if rope is not None:
                ops.rope_axial_heads_(y, rope[2] // rope[3], cos, sin, 4096, 4096)
            return y
        print(f"M={M:6d} N={N:5d} rope={rope is not None}: fused {fused:7.1f} us | separate {t(sep):7.1f} us")
