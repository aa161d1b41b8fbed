"""vg_attention_dv alone on the video branch's per-frame shapes (C2: one object, C4: eight): the LDS-DMA form (VG_ATTN_DMA=1) against r05's key-split kernel
(the library reads VG_ATTN_DMA once: run twice)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videoglamm_amd import ops

D, DV = 256, 64


def t(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


if __name__ != "__main__":
    SH = ()
else:
    SH = ((1, 4160), (1, 28736), (8, 28736), (8, 12000))
for B, Skv in SH:
    q = torch.randn(B, 4096, 1, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, Skv, 1, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, Skv, 1, DV, device="cuda", dtype=torch.bfloat16)
    us = t(lambda: ops.attention_dv(q, k, v, D ** -0.5))
    fl = 2.0 * B * 4096 * Skv * (D + DV)
    o = ops.attention_dv(q, k, v, D ** -0.5).float()
    print(f"VG_ATTN_DMA={os.environ.get('VG_ATTN_DMA', '1')} B={B} Skv={Skv}: {us:.1f} us ({fl / us / 1e6:.0f} TF/s)  checksum {float(o.abs().sum()):.4f}")
