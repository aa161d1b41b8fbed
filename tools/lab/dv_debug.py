import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from videoglamm_amd import ops
import _cpu_ops as ref
torch.manual_seed(0)
D = 256
for dt in (torch.bfloat16, torch.float32):
    for (B, Sq, Skv, DV) in [(1, 64, n, 64) for n in (27, 28, 31, 32, 60)] + [(1, 64, 28, 256), (1, 64, 60, 256), (1, 128, 28, 64), (1, 32, 28, 64)]:
        q, k, v = torch.randn(B, Sq, 1, D).to(dt), torch.randn(B, Skv, 1, D).to(dt), torch.randn(B, Skv, 1, DV).to(dt)
        f = ops.attention_dv if DV != D else ops.attention
        o = f(q.cuda(), k.cuda(), v.cuda(), D ** -0.5).float().cpu()
        r = ref.attention(q, k, v, D ** -0.5).float()
        e = (o - r).abs()
        print(dt, (B, Sq, Skv, DV), "max err per batch", [round(float(e[b].max()), 6) for b in range(B)],
              "bad query rows per batch", [int((e[b].amax(dim=(1, 2)) > 1e-4).sum()) for b in range(B)],
              "first bad rows b-last", (e[B - 1].amax(dim=(1, 2)) > 1e-4).nonzero().flatten()[:8].tolist(), flush=True)
