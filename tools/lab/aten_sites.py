"""Which lines of the package launch torch's own kernels (cat / copy / contiguous / clone / fill / index) inside one C2 clip: torch.profiler with python stacks,
device time per call site.  usage: python tools/lab/aten_sites.py [--branch video]"""
import argparse
import os
import sys
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
os.environ["VG_HIERA_START"] = "serial"
step(); step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
sites = defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.name in ("aten::empty", "aten::view", "aten::as_strided", "aten::empty_strided", "aten::reshape", "aten::select", "aten::slice"):
        continue
    dt = getattr(ev, "device_time_total", None) or getattr(ev, "cuda_time_total", 0)
    if not dt or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    where = next((s for s in ev.stack if "videoglamm_amd" in s or "bench.py" in s), "?")
    k = (ev.name, where.strip()[-110:])
    sites[k][0] += 1
    sites[k][1] += dt
tot = sum(v[1] for v in sites.values())
print(f"torch-native device time inside one clip: {tot / 1e3:.2f} ms in {sum(v[0] for v in sites.values())} calls")
for (name, where), (n, us) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{us / 1e3:8.3f} ms {n:5d} x  {name:22s} {where}")
