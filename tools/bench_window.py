"""Hiera stage-3 window attention (256-token windows, 8 heads of 72) on a 16-frame chunk: isolated time of vg_window_attention."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoglamm_amd import ops
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for frames in (16, 8):
    Bw, H, D = frames * 16, 8, 72
    qkv = torch.randn(Bw, 256, 3, H, D, device="cuda", dtype=torch.bfloat16)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    ms = t(lambda: ops.window_attention(q, k, v, D ** -0.5))
    fl = 4.0 * Bw * H * 256 * 256 * D
    print(f"window attention {frames} frames: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s  {(qkv.numel()*2 + Bw*256*H*D*2)/ms/1e9:6.2f} TB/s algorithmic", flush=True)
