"""Mask-decode stage alone (framewise branch on precomputed Hiera features): wall clock incl. host launch time, per configuration.
usage: python tools/lab/tw_stage.py FRAMES OBJECTS"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
T, NOBJ = int(sys.argv[1]), int(sys.argv[2])
sys.argv = sys.argv[:1] + ["--frames-per-gpu", str(T), "--objects", str(NOBJ)]
import bench
from videoglamm_amd import ops, synth
from videoglamm_amd.model import VideoGLaMMForCausalLM
args = bench.parse()
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
cfg = synth.videoglamm_llama3_8b()
sd = synth.device_state_dict(synth.manifest(cfg), dev, torch.bfloat16)
model = VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.bfloat16, device=dev)
images, context, sam, ids = bench.make_inputs(cfg, args, 1, dev)
feats = model.sam2.hiera_frames(sam, None)
emb = (torch.randn(NOBJ, 256, device=dev) * 0.5).to(torch.bfloat16)
def run():
    return model.sam2.framewise_branch(sam, emb, (args.src, args.src), frame_feats=feats, as_masks=True)[0]
for mode in ("1", "0", "1", "0"):
    os.environ["VG_TWOWAY_FUSED"] = mode
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): run()
    torch.cuda.synchronize()
    print(f"T={T} objects={NOBJ} VG_TWOWAY_FUSED={mode}: mask decode {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms", flush=True)
