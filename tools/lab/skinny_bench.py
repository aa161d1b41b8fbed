"""vg_gemm's skinny route (M <= 16 rows) on the mask decoder's token-side shapes, graph-replayed (launch gaps as in the product)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videoglamm_amd import ops
from mlp3_bench import graph_time

for M, N, K, res in ((9, 256, 256, True), (9, 128, 256, False), (9, 2048, 256, False), (9, 256, 2048, True), (1, 256, 256, False), (1, 128256, 4096, False)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").bfloat16() if res else None
    t = graph_time(lambda: ops.linear(x, w, b, residual=r), n=20 if N > 100000 else 50)
    print(f"M={M} N={N} K={K} residual={res}: {t:.1f} us")
