"""ops.attention alone on the clip's long-sequence shapes (LLM prefill, Hiera global blocks, the towers): TF/s per shape; VG_KERNELS_SO picks a variant build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videoglamm_amd import ops


def t(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


SHAPES = [("LLM prefill C2", 1, 32, 8, 3361, 128, True), ("Hiera global s3", 32, 8, 8, 4096, 72, False), ("Hiera global s4", 32, 16, 16, 1024, 72, False),
          ("CLIP", 32, 16, 16, 577, 64, False), ("InternVideo2", 4, 16, 16, 1025, 88, False), ("d128 dense 8k", 1, 16, 16, 8192, 128, False)]
for name, B, H, Hkv, S, D, causal in SHAPES:
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, S, Hkv, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, S, Hkv, D, device="cuda", dtype=torch.bfloat16)
    us = t(lambda: ops.attention(q, k, v, D ** -0.5, causal))
    fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
    print(f"{os.environ.get('VG_KERNELS_SO', 'default')[-14:]:>14s}  {name:18s} {us:9.1f} us  {fl / us / 1e6:6.0f} TF/s", flush=True)
