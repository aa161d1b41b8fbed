"""vg_attention at D = DV = 256 on the memory self-attention's shapes (one / eight objects); VG_ATTN_DMA=0: r05's key-split kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videoglamm_amd import ops
from dv_bench import t

D = 256
for B in (1, 8):
    q, k, v = (torch.randn(B, 4096, 1, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    us = t(lambda: ops.attention(q, k, v, D ** -0.5))
    print(f"VG_ATTN_DMA={os.environ.get('VG_ATTN_DMA', '1')} self-attention B={B}: {us:.1f} us ({4.0 * B * 4096 * 4096 * D / us / 1e6:.0f} TF/s)")
