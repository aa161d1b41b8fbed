"""Serial stage budget of the C1 workload (same synthetic model and inputs as bench.py): each stage timed alone with
HIP events on the launch stream, then the overlapped end-to-end step for comparison.
usage: python tools/bench_parts.py [reps=3] [hiera|towers|prefill]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from videoglamm_amd import ops, synth  # noqa: E402
from videoglamm_amd.model import VideoGLaMMForCausalLM  # noqa: E402
from videoglamm_amd.vlm import generate  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
only = sys.argv[2] if len(sys.argv) > 2 else ""      # "hiera" / "towers" / "prefill": run just that stage (for rocprofv3)
sys.argv = sys.argv[:1]
args = bench.parse()
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
cfg = synth.videoglamm_llama3_8b()
cfg["forced_tokens"] = {8: cfg["seg_token_idx"]}
sd = synth.device_state_dict(synth.manifest(cfg), dev, torch.bfloat16)
model = synth.install_forced_tokens(VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.bfloat16, device=dev))
images, context, sam, ids = bench.make_inputs(cfg, args, 1, dev)


def timed(fn, n=reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


if only:
    fn = {"hiera": lambda: model.sam2.hiera_frames(sam, None),
          "towers": lambda: model.towers.encode(images, context),
          "prefill": lambda: generate(model.P, model.cfg, model.towers, images, context, ids[0].cpu(), 0,
                                      visual=torch.zeros(208 * args.te, cfg["llm"]["hidden"], dtype=torch.bfloat16, device=dev))}[only]
    ms, _ = timed(fn)
    print(f"{only}: {ms:.2f} ms")
    sys.exit(0)
e2e, _ = timed(lambda: model.inference([images], [context], [sam], ids, [(1024, 1024)], [(args.src, args.src)], max_new_tokens=args.max_new_tokens))
t_enc, visual = timed(lambda: model.towers.encode(images, context))
t_iv2, _ = timed(lambda: model.towers.iv2(images.view(images.shape[0] // 4, 4, *images.shape[1:])))
t_clip, _ = timed(lambda: model.towers.clip(context))
t_gen0, _ = timed(lambda: generate(model.P, model.cfg, model.towers, images, context, ids[0].cpu(), 0, visual=visual))
t_gen, (out_ids, emb) = timed(lambda: generate(model.P, model.cfg, model.towers, images, context, ids[0].cpu(), args.max_new_tokens,
                                                visual=visual, token_hook=synth.forced_tokens_hook(cfg["forced_tokens"])))
t_hiera, feats = timed(lambda: model.sam2.hiera_frames(sam, None))
t_fw, _ = timed(lambda: ops.threshold(model.sam2.framewise_branch(sam, emb, (args.src, args.src), frame_feats=feats)[0]).cpu())
print(f"towers.encode {t_enc:7.2f} ms  (iv2 {t_iv2:.2f}, clip {t_clip:.2f})")
print(f"llm prefill   {t_gen0:7.2f} ms  (S = {visual.shape[0] + ids.shape[1] - args.te})")
print(f"llm decode    {t_gen - t_gen0:7.2f} ms  ({args.max_new_tokens} tokens -> {(t_gen - t_gen0) / max(args.max_new_tokens - 1, 1):.3f} ms/token)")
print(f"hiera+fpn     {t_hiera:7.2f} ms  ({sam.shape[0]} frames)")
print(f"mask decode   {t_fw:7.2f} ms  (framewise, incl. threshold + D2H)")
print(f"serial sum    {t_enc + t_gen + t_hiera + t_fw:7.2f} ms;  end-to-end step (hiera overlapped) {e2e:7.2f} ms")
