"""Diagnostic (GPU box): where does the bf16 run of the bench workload part from the fp32 parity run?  Records the inputs and the
choice of every vg_multimask_select call (4 candidate masks, predicted IoUs, chosen index) and the Hiera features for both modes.
usage: python tools/diag_bf16_c2.py [--frames 8] [--te 8]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--te", type=int, default=8)
ap.add_argument("--src", type=int, default=1024)
args = ap.parse_args()

from videoglamm_amd import ops, synth  # noqa: E402
from videoglamm_amd.model import VideoGLaMMForCausalLM  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = synth.videoglamm_llama3_8b()
cfg["forced_tokens"] = {8: cfg["seg_token_idx"]}
g = torch.Generator().manual_seed(1234)
images = torch.randn(args.te, 3, 224, 224, generator=g).to(dev)
context = torch.randn(args.te, 3, 336, 336, generator=g).to(dev)
sam = torch.randn(args.frames, 3, 1024, 1024, generator=g).to(dev)
ids = torch.cat([torch.tensor([1, 5, 6]), torch.full((args.te,), -200), torch.randint(3, cfg["llm"]["vocab"] - 2, (30,), generator=g)])[None]

rec = {}
orig = ops.multimask_select


def spy(masks, ious, tokens, mode, *a, **k):
    out = orig(masks, ious, tokens, mode, *a, **k)
    rec.setdefault(cur[0], []).append((masks.float().cpu(), ious.float().cpu(), out[3].cpu()))
    return out


ops.multimask_select = spy
cur = [None]
outs = {}
sd16 = synth.device_state_dict(synth.manifest(cfg), dev, torch.bfloat16)
ids16 = None
for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
    cur[0] = name
    c = dict(cfg)
    if ids16 is not None:
        c["forced_tokens"] = {i: t for i, t in enumerate(ids16)}
    sd = sd16 if dt == torch.bfloat16 else {k: v.float() for k, v in sd16.items()}
    m = synth.install_forced_tokens(VideoGLaMMForCausalLM(sd, c, torch_dtype=dt, device=dev))
    cap = m.capture = {}
    out_ids, _ = m.inference([images], [context], [sam], ids, [(1024, 1024)], [(args.src, args.src)], max_new_tokens=32)
    if ids16 is None:
        ids16 = out_ids[0].tolist()[ids.shape[1]:]
    outs[name] = (cap["logits"].float().cpu(), cap["emb"].float().cpu())
    del m, sd
    torch.cuda.empty_cache()

lb, l32 = outs["bf16"][0], outs["fp32"][0]
print("emb cosine", float(torch.nn.functional.cosine_similarity(outs["bf16"][1], outs["fp32"][1]).min()))
print("final mask fraction bf16 / fp32:", float((lb > 0).float().mean()), float((l32 > 0).float().mean()))
for (mb, ib, xb), (mf, if_, xf) in zip(rec["bf16"], rec["fp32"]):
    N = mb.shape[0]
    for n in range(N):
        iou_tok = [float(((mb[n, t] > 0) & (mf[n, t] > 0)).sum() / ((mb[n, t] > 0) | (mf[n, t] > 0)).sum().clamp_min(1)) for t in range(4)]
        def stab(m):
            x = m[n, 0]
            return float((x > 0.05).sum() / (x > -0.05).sum().clamp_min(1))
        print(f"item {n}: chosen bf16 {int(xb[n])} fp32 {int(xf[n])} | stability tok0 bf16 {stab(mb):.4f} fp32 {stab(mf):.4f} | pred iou bf16 {[round(float(v), 3) for v in ib[n]]} "
              f"fp32 {[round(float(v), 3) for v in if_[n]]} | per-token mask IoU bf16 vs fp32 {[round(v, 4) for v in iou_tok]} | logit corr tok0 "
              f"{float(torch.corrcoef(torch.stack([mb[n, 0].flatten(), mf[n, 0].flatten()]))[0, 1]):.5f}")
