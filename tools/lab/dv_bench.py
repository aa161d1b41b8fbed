"""vg_attention_dv alone on the video branch's per-frame shapes (C2: one object, C4: eight)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoglamm_amd import ops
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for B in (1, 8):
    Sq, Skv, D = 4096, 7 * 4096 + 64, 256
    q = torch.randn(B, Sq, 1, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, Skv, 1, D, device="cuda", dtype=torch.bfloat16)
    v64 = torch.randn(B, Skv, 1, 64, device="cuda", dtype=torch.bfloat16)
    v256 = torch.randn(B, Skv, 1, D, device="cuda", dtype=torch.bfloat16)
    a = t(lambda: ops.attention_dv(q, k, v64, D ** -0.5))
    b = t(lambda: ops.attention(q, k, v256, D ** -0.5))
    s = t(lambda: ops.attention(q, k[:, :4096], v256[:, :4096], D ** -0.5))
    print(f"B={B}: dv(256|64) {a*1e3:8.1f} us = {2*B*Sq*Skv*320/a/1e9:6.1f} TF/s | full 256 {b*1e3:8.1f} us = {4*B*Sq*Skv*256/b/1e9:6.1f} TF/s | self 4096^2 {s*1e3:8.1f} us", flush=True)
