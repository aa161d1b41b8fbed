"""A/B timing of the head-dim-256 attention shapes of SAM2's memory attention (self: 4096 x 4096, cross: 4096 x up to 28736)."""
import sys, torch
sys.path.insert(0, ".")
from videoglamm_amd import ops
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for B in (1, 4):
    for Skv in (4096, 8256, 16448, 28736):
        q = torch.randn(B, 4096, 1, 256, device="cuda", dtype=torch.bfloat16); k = torch.randn(B, Skv, 1, 256, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k)
        ms = t(lambda: ops.attention(q, k, v, 1 / 16, False))
        fl = 4.0 * B * 4096 * Skv * 256
        print(f"B={B} Skv={Skv:6d} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TF/s", flush=True)
