"""Debug aid: every aten op torch itself launches during ONE inference step of the C2 workload (the product's arithmetic is all vg_* kernels:
what shows up here is data movement — copies, cats, casts, index ops), grouped by op and call site inside videoglamm_amd, with the bytes it moves.
usage: python tools/find_torch_ops.py [framewise|video] [bench.py arguments]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from videoglamm_amd import synth  # noqa: E402
from videoglamm_amd.model import VideoGLaMMForCausalLM  # noqa: E402

branch = sys.argv[1] if len(sys.argv) > 1 else "framewise"
sys.argv = sys.argv[:1] + sys.argv[2:]          # further arguments go to bench.parse() (e.g. --frames 64 --objects 8)
args = bench.parse()
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
cfg = synth.videoglamm_llama3_8b()
cfg["forced_tokens"] = {8: cfg["seg_token_idx"]} if args.objects == 1 else {4 + 3 * i: cfg["seg_token_idx"] for i in range(args.objects)}
sd = synth.device_state_dict(synth.manifest(cfg), dev, torch.bfloat16)
model = synth.install_forced_tokens(VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.bfloat16, device=dev))
images, context, sam, ids = bench.make_inputs(cfg, args, 1, dev)
step = lambda: model.inference([images], [context], [sam], ids, [(1024, 1024)], [(args.src, args.src)], max_new_tokens=args.max_new_tokens,  # noqa: E731
                               use_sam2_video_branch=branch == "video")
os.environ["VG_VIDEO_GRAPH"] = "0"
step()
step()
sites = collections.defaultdict(lambda: [0, 0])
SKIP = ("aten.view", "aten.as_strided", "aten.slice", "aten.select", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.transpose", "aten.permute",
        "aten._unsafe_view", "aten.detach", "aten.alias", "aten.t.", "aten.empty", "aten.unbind", "aten.split", "aten.chunk", "aten.narrow", "aten._local_scalar_dense",
        "aten.is_pinned", "aten.stride", "aten.sym_", "aten.reshape", "aten.lift_fresh", "aten.record_stream")


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, a=(), k=None):
        out = func(*a, **(k or {}))
        name = str(func)
        if not name.startswith(SKIP):
            ts = [t for t in (out if isinstance(out, (tuple, list)) else [out]) if isinstance(t, torch.Tensor)]
            if any(t.is_cuda for t in ts) or any(isinstance(x, torch.Tensor) and x.is_cuda for x in a):
                fr = [f for f in traceback.extract_stack() if "videoglamm_amd" in f.filename]
                where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-2:][::-1])
                e = sites[(name, where)]
                e[0] += 1
                e[1] += sum(t.numel() * t.element_size() for t in ts)
        return out


with Spy():
    step()
torch.cuda.synchronize()
print(f"{'launches':>8} {'MB out':>10}  op  <- call site")
for (name, where), (n, b) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{n:8d} {b / 1e6:10.1f}  {name}  <- {where}")
