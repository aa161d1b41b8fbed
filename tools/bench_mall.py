"""Is a decode GEMV faster when its weights sit in the 256 MB Infinity Cache?  Same kernel on (a) one weight matrix re-read every launch (cache-resident when it fits)
and (b) a ring of matrices larger than the cache (every launch streams from HBM)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoglamm_amd import ops
def t(fn, n):
    fn(0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, N, K, glu in [("o_proj", 4096, 4096, False), ("qkv", 6144, 4096, False), ("down", 4096, 14336, False), ("gate|up", 14336, 4096, True), ("gate|up first 96 MB", 5888, 4096, True)]:
    rows = 2 * N if glu else N
    mb = rows * K * 2 / 1e6
    ring = [torch.randn(rows, K, device="cuda", dtype=torch.bfloat16) for _ in range(max(2, int(1200 / mb)))]
    x = torch.randn(1, K, device="cuda", dtype=torch.bfloat16)
    hot = t(lambda i: ops.decode_gemv(x, ring[0], glu=glu), 40)
    cold = t(lambda i: ops.decode_gemv(x, ring[i % len(ring)], glu=glu), 40)
    print(f"{name:22s} {mb:6.1f} MB: cache-resident {hot*1e3:6.1f} us ({mb/hot/1e3:5.2f} TB/s) | from HBM {cold*1e3:6.1f} us ({mb/cold/1e3:5.2f} TB/s)")
