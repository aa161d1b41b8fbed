"""Isolated bandwidth of the row / pointwise kernels on the C2 shapes (HBM-bound: read + write of the activation)."""
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
for name, rows, C in [("hiera s1 ln", 1048576, 144), ("hiera s2 ln", 262144, 288), ("hiera s3 ln", 65536, 576), ("hiera s4 ln", 16384, 1152),
                      ("iv2 ln", 16400, 1408), ("clip ln", 9232, 1024), ("llm rms", 3361, 4096)]:
    x = torch.randn(rows, C, device="cuda", dtype=torch.bfloat16); w = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    ms = t(lambda: ops.layernorm(x, w, b, 1e-6)) if "rms" not in name else t(lambda: ops.rmsnorm(x, w, 1e-5))
    print(f"{name:12s} rows={rows:8d} C={C:5d} {ms*1e3:8.1f} us  {rows*C*4/ms/1e9:8.2f} TB/s", flush=True)
x = torch.randn(65536, 2304, device="cuda", dtype=torch.bfloat16); y = torch.randn_like(x)
ms = t(lambda: ops.axpby(x, y)); print(f"axpby 65536x2304 {ms*1e3:8.1f} us {x.numel()*6/ms/1e9:8.2f} TB/s")
ms = t(lambda: x.clone()); print(f"torch clone     {ms*1e3:8.1f} us {x.numel()*4/ms/1e9:8.2f} TB/s")
