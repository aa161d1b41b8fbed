"""Feasibility probe: the MFMA flash kernel on the decode-step shape (GQA rows folded: 4 query rows per KV head) with FEW KV splits, next to the
decode step's own attention kernel — graph replays of 64 launches each, so the numbers are kernel time, not host launch time."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoglamm_amd import ops, _lib
from videoglamm_amd.ops import _p, _dt, _stream
lib = _lib.load()
H, Hkv, D, S, MAXL = 32, 8, 128, 3400, 3456
dev = "cuda"
q = torch.randn(1, 1, H, D, device=dev, dtype=torch.bfloat16)
kc = torch.randn(MAXL, Hkv, D, device=dev, dtype=torch.bfloat16)
vc = torch.randn(MAXL, Hkv, D, device=dev, dtype=torch.bfloat16)
pos = torch.tensor([S - 1], device=dev, dtype=torch.int32)
out = torch.empty(1, 1, H, D, device=dev, dtype=torch.bfloat16)

def run_graph(fn, n=64, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / n * 1e3

for ns in (1, 2, 4, 8, 16, 32, 54):
    ws = torch.empty(H * ns * (D + 2), device=dev, dtype=torch.float32)
    def f():
        rc = lib.vg_attention_splitkv(_p(q), _p(kc), _p(vc), _p(out), 1, H, Hkv, 1, MAXL, D, q.stride(0), q.stride(1), q.stride(2), 0, kc.stride(0), kc.stride(1),
                                      0, vc.stride(0), vc.stride(1), out.stride(0), out.stride(1), out.stride(2), D ** -0.5, 1, _dt(q), _p(ws), ws.numel(), ns, _p(pos), _stream())
        assert rc == 0
    print(f"flash fold, nsplit {ns:3d}: {run_graph(f):7.2f} us per call (attention + combine when nsplit > 1)", flush=True)
