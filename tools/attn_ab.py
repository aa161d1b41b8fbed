import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoglamm_amd import ops
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, B, H, Hkv, Sq, Skv, D, causal in [("llm prefill c2", 1, 32, 8, 3361, 3361, 128, True), ("hiera glob 16f", 16, 8, 8, 4096, 4096, 72, False),
                                            ("iv2", 4, 16, 16, 1025, 1025, 88, False), ("llm prefill c1", 1, 32, 8, 1697, 1697, 128, True),
                                            ("clip 16f", 16, 16, 16, 577, 577, 64, False), ("hiera glob s4", 16, 16, 16, 1024, 1024, 72, False)]:
    q = torch.randn(B, Sq, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn(B, Skv, Hkv, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k)
    ms = t(lambda: ops.attention(q, k, v, D ** -0.5, causal))
    fl = 4.0 * B * H * Sq * Skv * D * (0.5 if causal else 1.0)
    print(f"attn {name:16s} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TF/s")
