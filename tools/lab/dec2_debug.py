"""per-position error of vg_decode_attention2 against the fp32 statement (debug aid)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import _cpu_ops as ref
from videoglamm_amd import ops
cuda = torch.device("cuda:0")
D, max_len = 128, 2048
for H, Hkv in ((8, 8), (32, 8)):
    g = torch.Generator().manual_seed(1)
    kc = torch.randn(max_len, Hkv, D, generator=g).to(torch.bfloat16)
    vc = torch.randn(max_len, Hkv, D, generator=g).to(torch.bfloat16)
    ws = ops.decode_attention_workspace(H, Hkv, D, max_len, cuda)
    for kpw in (128, 256):
        for pos in (0, 1, 3, 4, 15, 16, 31, 32, 33, 127, 128, 129, 255, 256, 257, 1000):
            q = torch.randn(1, H * D, generator=g).to(torch.bfloat16)
            pd = torch.tensor([pos], dtype=torch.int32)
            o = ops.decode_attention2(q.to(cuda), kc.to(cuda), vc.to(cuda), H, Hkv, D, pd.to(cuda), D ** -0.5, ws, keys_per_wg=kpw).float().cpu()
            want = ref.attention_decode(q.view(1, 1, H, D), kc, vc, pd, D ** -0.5, 0).view(1, H * D).float()
            err = (o - want).abs().view(H, D)
            print(f"H={H} Hkv={Hkv} kpw={kpw} pos={pos}: max err {float(err.max()):.3f}; per-head max {[round(float(e), 2) for e in err.max(dim=1).values[:8]]}; "
                  f"per-d16 max {[round(float(e), 2) for e in err.view(H, 8, 16).amax(dim=(0, 2))]}", flush=True)
