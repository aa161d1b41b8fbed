"""Does the HBM-bound decode loop co-run with the MFMA-bound Hiera pass?  Times (a) the decode loop alone, (b) Hiera + FPN alone, (c) both at once
on two streams (same model and inputs as bench.py's default workload).  usage: python tools/bench_corun.py [reps=2]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from videoglamm_amd import synth  # noqa: E402
from videoglamm_amd.model import VideoGLaMMForCausalLM  # noqa: E402
from videoglamm_amd.vlm import generate  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sys.argv = sys.argv[:1]
args = bench.parse()
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
cfg = synth.videoglamm_llama3_8b()
sd = synth.device_state_dict(synth.manifest(cfg), dev, torch.bfloat16)
model = synth.install_forced_tokens(VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.bfloat16, device=dev))
images, context, sam, ids = bench.make_inputs(cfg, args, 1, dev)
visual = torch.zeros(208 * args.te, cfg["llm"]["hidden"], dtype=torch.bfloat16, device=dev)
generate(model.P, model.cfg, model.towers, images, context, ids[0].cpu(), 2, visual=visual)      # prefill + graph capture
dec = model.P._decoder
side = torch.cuda.Stream()
S0 = dec.pos - 1                       # a position whose KV rows are all written (the prefill's) — uninitialised rows could hold NaNs
tok0 = dec.tok_dev.clone()


def decode(n=31):
    for _ in range(n):
        dec.pos_dev.fill_(S0)
        dec.tok_dev.copy_(tok0)
        dec.pos = S0
        dec.decode_step()


def hiera():
    return model.sam2.hiera_frames(sam, None)


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        f = hiera()
    e = torch.cuda.Event(enable_timing=False)
    decode()
    e.record()
    torch.cuda.current_stream().wait_stream(side)
    return f


decode(2); hiera(); both()
for _ in range(reps):
    a, b, c = wall(decode), wall(hiera), wall(both)
    print(f"decode alone {a:7.1f} ms | hiera alone {b:7.1f} ms | both {c:7.1f} ms  (sum {a + b:.1f}, max {max(a, b):.1f})")
