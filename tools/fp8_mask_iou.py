"""Masks of the fp8 LLM path (fp8 MFMA prefill + fp8 decode weights) against the bf16 path on the full-size synthetic C1
model (Llama-3-8B + towers + SAM2-L, random weights), teacher-forced: the fp8 run is made to emit the ids of the bf16 run, so
the [SEG] hidden states differ by the fp8 arithmetic only.  Prints the cosine of the [SEG] embeddings seen by SAM2 (through the
masks) and the per-mask IoU.  usage: python tools/fp8_mask_iou.py [objects]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoglamm_amd import synth  # noqa: E402
from videoglamm_amd.model import VideoGLaMMForCausalLM  # noqa: E402

torch.set_grad_enabled(False)
objects = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
cfg = synth.videoglamm_llama3_8b()
seg = cfg["seg_token_idx"]
sd = synth.device_state_dict(synth.manifest(cfg), dev, torch.bfloat16)
g = torch.Generator().manual_seed(0)
T, te, S, src, new = 4, 8, 1024, 512, 16
images, context = torch.randn(te, 3, 224, 224, generator=g).to(dev), torch.randn(te, 3, 336, 336, generator=g).to(dev)
sam = torch.randn(T, 3, S, S, generator=g).to(dev)
ids = torch.cat([torch.tensor([1, 5, 6]), torch.full((te,), -200), torch.randint(3, cfg["llm"]["vocab"] - 2, (30,), generator=g)])[None]
forced = {3 + 4 * i: seg for i in range(objects)}
base = synth.install_forced_tokens(VideoGLaMMForCausalLM(sd, dict(cfg, forced_tokens=forced), torch_dtype=torch.bfloat16, device=dev))
out_ids, segs = base.inference([images], [context], [sam], ids, [(S, S)], [(src, src)], max_new_tokens=new)
emitted = out_ids[0, ids.shape[1]:].tolist()
ref = np.stack([np.stack([segs[0][t][k] for k in sorted(segs[0][t])]) for t in sorted(segs[0])])
del base
for name, llm in (("fp8 decode weights", dict(cfg["llm"], decode_weights="fp8")), ("fp8 prefill + decode weights", dict(cfg["llm"], decode_weights="fp8", prefill_gemm="fp8"))):
    sd["_"] = None
    sd.pop("_")
    m = synth.install_forced_tokens(VideoGLaMMForCausalLM(sd, dict(cfg, llm=llm, forced_tokens={i: t for i, t in enumerate(emitted)}), torch_dtype=torch.bfloat16, device=dev))
    if hasattr(m.P, "_decoder"):
        del m.P._decoder
    o2, s2 = m.inference([images], [context], [sam], ids, [(S, S)], [(src, src)], max_new_tokens=new)
    assert o2[0].tolist() == out_ids[0].tolist()
    got = np.stack([np.stack([s2[0][t][k] for k in sorted(s2[0][t])]) for t in sorted(s2[0])])
    iou = [[float((got[t, n] & ref[t, n]).sum() / max((got[t, n] | ref[t, n]).sum(), 1)) for n in range(ref.shape[1])] for t in range(T)]
    flat = np.array(iou).ravel()
    print(f"{name}: {ref.shape[1]} objects x {T} frames, mask IoU vs bf16: mean {flat.mean():.4f} median {np.median(flat):.4f} min {flat.min():.4f}")
    print("   per-mask [t][n]:", np.round(np.array(iou), 3).tolist())
    del m
