"""Diagnostic (GPU box): first operator whose output for frame F differs between a batched and a single-frame Hiera pass.
Every ops.* call is logged; batch is the outermost index of every layout, so frame F of a batched output is a contiguous slice."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videoglamm_amd import ops, synth  # noqa: E402
from videoglamm_amd.params import Params  # noqa: E402
from videoglamm_amd.sam2 import SAM2  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
dt = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == "fp32") else torch.bfloat16
B, F = 4, 2
cfg = synth.SAM2_L
sd = synth.device_state_dict(synth.sam2_manifest(cfg), dev, torch.bfloat16)
if dt == torch.float32:
    sd = {k: v.float() for k, v in sd.items()}
img = torch.randn(B, 3, 1024, 1024, generator=torch.Generator().manual_seed(7)).to(dev)
m = SAM2(Params(sd, dev, dt), "", cfg)
log = []
for name in dir(ops):
    fn = getattr(ops, name)
    if isinstance(fn, types.FunctionType) and not name.startswith("_") and fn.__module__ == ops.__name__:
        def make(fn, name):
            def w(*a, **k):
                y = fn(*a, **k)
                t = y[0] if isinstance(y, tuple) else y
                if torch.is_tensor(t):
                    log.append((name, t, [tuple(x.shape) for x in a if torch.is_tensor(x)]))
                return y
            return w
        setattr(ops, name, make(fn, name))
m.forward_image(img)
big = log[:]
log.clear()
m.forward_image(img[F:F + 1])
small = log[:]
assert len(big) == len(small), (len(big), len(small))
bad = 0
for i, ((n1, t1, s1), (n2, t2, s2)) in enumerate(zip(big, small)):
    a = t1.reshape(B, -1)[F].float()
    b = t2.reshape(1, -1)[0].float()
    d = float((a - b).abs().max())
    if d > 1e-2 * max(1.0, float(b.abs().max())):
        print(f"op #{i} {n1}: out {tuple(t1.shape)} vs {tuple(t2.shape)} inputs {s1} max diff {d:.4f} (|ref| max {float(b.abs().max()):.3f})")
        bad += 1
        if bad >= 4:
            break
print("ops compared:", len(big), "first bad shown above" if bad else "no divergence")
