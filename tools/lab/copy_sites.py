"""Which lines of the package make torch copy device tensors inside one clip (.contiguous() on a strided view, .clone(), .to(dtype), copy_, cat): patched
Tensor methods that record (call site, bytes) when a copy really happens.  The profiler cannot see these sites when they sit inside captured graphs or
when its stacks come back empty (tools/lab/aten_sites.py).  usage: python tools/lab/copy_sites.py [--branch video]"""
import argparse
import os
import sys
import traceback
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from videoglamm_amd import synth  # noqa: E402
from videoglamm_amd.model import VideoGLaMMForCausalLM  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--branch", default="framewise")
a = ap.parse_args()
sys.argv = [sys.argv[0]]
args = bench.parse()
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = synth.videoglamm_llama3_8b()
cfg["forced_tokens"] = {8: cfg["seg_token_idx"]}
sd = synth.device_state_dict(synth.manifest(cfg), dev, torch.bfloat16)
model = synth.install_forced_tokens(VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.bfloat16, device=dev))
images, context, sam, ids = bench.make_inputs(cfg, args, 1, dev)
step = lambda: model.inference([images], [context], [sam], ids, [(1024, 1024)], [(args.src, args.src)], max_new_tokens=args.max_new_tokens,  # noqa: E731
                               use_sam2_video_branch=a.branch == "video")
os.environ["VG_VIDEO_GRAPH"] = "0"
step(); step()
torch.cuda.synchronize()
sites = defaultdict(lambda: [0, 0])
on = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "videoglamm_amd" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:90]}"
    return "?"


def rec(kind, t):
    if on[0] and t.is_cuda:
        k = (kind, site())
        sites[k][0] += 1
        sites[k][1] += t.numel() * t.element_size()


from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

VIEWS = ("view", "reshape", "_unsafe_view", "as_strided", "expand", "permute", "transpose", "t.", "slice", "select", "unsqueeze", "squeeze", "alias", "detach", "empty", "unbind",
         "split", "_local_scalar_dense", "item", "is_", "sym_", "stride", "size", "numel", "dim", "_reshape_alias", "unfold", "narrow", "chunk", "lift_fresh", "record_stream")


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "")
        if on[0] and not name.startswith(VIEWS):
            t = out if isinstance(out, torch.Tensor) else next((x for x in args if isinstance(x, torch.Tensor)), None)
            if t is not None and t.is_cuda:
                k = (name, site())
                sites[k][0] += 1
                sites[k][1] += t.numel() * t.element_size()
        return out


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "videoglamm_amd" in fr.filename and "_python_dispatch" not in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:90]}"
    return "?"


with Spy():
    on[0] = True
    step()
    on[0] = False
torch.cuda.synchronize()
print(f"torch ops that launch kernels inside one clip ({a.branch}): {sum(v[0] for v in sites.values())} calls, {sum(v[1] for v in sites.values()) / 1e6:.1f} MB")
for (kind, where), (n, b) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{b / 1e6:9.2f} MB {n:5d} x  {kind:11s} {where}")
