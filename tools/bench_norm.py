import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from videoglamm_amd import ops
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rows, C in [(32768, 576), (131072, 288), (524288, 144), (8192, 1152), (4616, 1024), (2050, 1408), (1697, 4096)]:
    x = torch.randn(rows, C, device="cuda", dtype=torch.bfloat16); w = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    ms = t(lambda: ops.layernorm(x, w, b, 1e-6))
    print(f"layernorm rows={rows} C={C} {ms*1e3:8.1f} us  {rows*C*4/ms/1e6:8.1f} GB/s")
