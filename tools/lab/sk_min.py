import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoglamm_amd import ops
torch.manual_seed(0)
for (M, N, K) in [(3361, 6144, 4096), (9232, 1024, 4096), (3361, 4096, 14336), (16384, 1152, 4608)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5
    print("launch", M, N, K, flush=True)
    y = ops.linear(a, w, out_dtype=torch.float32)
    torch.cuda.synchronize()
    ref = a.float() @ w.float().t()
    err = (y - ref).abs().max().item()
    print("max err", err, "bad rows", int(((y - ref).abs().amax(dim=1) > 0.05).sum()), flush=True)
    y2 = ops.linear(a, w, out_dtype=torch.float32)
    torch.cuda.synchronize()
    print("repeat equal", torch.equal(y, y2), flush=True)
