import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoglamm_amd import ops
D, DV = 256, 64
torch.set_printoptions(linewidth=250, precision=3, sci_mode=False)
for Skv in (27, 28, 30, 60, 63, 700, 2100):
    DT = torch.bfloat16
    q = torch.zeros(1, 64, 1, D, dtype=DT); k = torch.zeros(1, Skv, 1, D, dtype=DT)
    v = torch.zeros(1, Skv, 1, DV, dtype=DT)
    for j in range(Skv):
        v[0, j, 0, j % DV] = 1.0
    o = ops.attention_dv(q.cuda(), k.cuda(), v.cuda(), D ** -0.5).float().cpu()[0, :, 0, :] * Skv
    print("Skv", Skv)
    # random q/k, one-hot v: o[q, d] = P[q, key d]
    q = torch.randn(1, 64, 1, D).to(DT); k = torch.randn(1, Skv, 1, D).to(DT)
    o = ops.attention_dv(q.cuda(), k.cuda(), v.cuda(), D ** -0.5).float().cpu()[0, :, 0, :]
    P = torch.softmax((q[0, :, 0].float() @ k[0, :, 0].float().t()) * D ** -0.5, dim=-1)
    Pm = torch.zeros(64, DV)
    for j in range(Skv): Pm[:, j % DV] += P[:, j]
    e = (o - Pm).abs()
    print("   max P error per column:", e.amax(dim=0).max().item(), "argmax col", e.amax(dim=0).argmax().item())
    
