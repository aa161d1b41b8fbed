"""the video branch's HBM-bound projections at 8 objects (C4): ours (per routing knob, one process each) beside the vendor library"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.nn.functional as F
from videoglamm_amd import ops, _lib
def t(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, N, K) in [(32768, 256, 256), (32768, 768, 256), (32768, 2048, 256), (32768, 256, 2048), (229888, 256, 64), (32768, 256, 64), (4096, 256, 256), (4096, 768, 256)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, device="cuda")
    ms = min(t(lambda: ops.linear(a, w, b)) for _ in range(2))
    ml = min(t(lambda: F.linear(a, w)) for _ in range(2))
    byt = (M * K + N * K + M * N) * 2
    print(f"M={M:7d} N={N:5d} K={K:5d} route {_lib.load().vg_gemm_route(M, N, K, 1, 0, 0)}  ours {ms*1e3:7.1f} us = {byt/ms/1e9:5.2f} TB/s | lib {ml*1e3:7.1f} us", flush=True)
