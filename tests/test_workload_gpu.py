"""Whole-workload checks (-m gpu) of the configurations BASELINE.json names, at full architecture size, built exactly the way bench.py
builds them (synthetic weights of the exact architectures, the same inputs):

  C2  32-frame 1024^2 clip, Te = 16, Llama-3-8B bf16 + InternVideo2-1B + CLIP-L/336 + SAM2-L, one [SEG] object, 32 greedy tokens — the
      configuration the headline is quoted on: the bf16 run against the SAME clip in fp32 parity mode on the GPU (the exact-fp32 MFMA path
      the fixture tests pin to the reference within 1e-3), teacher-forced to the bf16 ids — bench.py's `quality` object as a test.
  C4  (shape) 8 [SEG] objects on a 16-frame clip, bf16 and the fp8 LLM path (fp8 MFMA prefill + fp8 decode weights), framewise branch:
      the fp8 run teacher-forced to the bf16 ids; masks of all 8 objects against the bf16 run's.
The reference itself cannot run at these sizes here (the oracle needs 17 s of CPU for ONE SAM2-L frame)."""
import os
import sys

import pytest
import torch

from videoglamm_amd import synth  # noqa: E402  (harness helpers: synthetic weights, the forced-[SEG] token hook)

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench_args(argv):
    import bench
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        return bench, bench.parse()
    finally:
        sys.argv = old


def _build(bench, args, cuda, llm_extra=None):
    from videoglamm_amd import synth
    from videoglamm_amd.model import VideoGLaMMForCausalLM
    cfg = synth.videoglamm_llama3_8b()
    if llm_extra:
        cfg["llm"] = dict(cfg["llm"], **llm_extra)
    cfg["forced_tokens"] = {8: cfg["seg_token_idx"]} if args.objects == 1 else {4 + 3 * i: cfg["seg_token_idx"] for i in range(args.objects)}
    sd = synth.device_state_dict(synth.manifest(cfg), cuda, torch.bfloat16)
    model = synth.install_forced_tokens(VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.bfloat16, device=cuda))
    del sd
    images, context, sam, ids = bench.make_inputs(cfg, args, 1, cuda)

    def step():
        return model.inference([images], [context], [sam], ids, [(1024, 1024)], [(args.src, args.src)], max_new_tokens=args.max_new_tokens,
                               use_sam2_video_branch=args.branch == "video")
    return cfg, model, step, ids


def test_c2_workload_bf16_vs_fp32_mode(cuda):
    bench, args = _bench_args([])
    cfg, model, step, ids = _build(bench, args, cuda)
    q = bench.quality(cfg, args, model, step, cuda)
    print("C2 whole workload:", q)
    out_ids, segs = step()
    assert out_ids.shape[1] == ids.shape[1] + args.max_new_tokens and sorted(segs[0]) == list(range(32)) and segs[0][0][0].shape == (1024, 1024)
    assert q["finite"] and q["seg_objects"] == 1
    assert q["mask_miou_vs_fp32"] > 0.99 and q["min_frame_iou_vs_fp32"] > 0.98, q
    assert q["ids_top1_agree"] >= 0.9 and q["seg_emb_cosine"] > 0.999, q
    assert 0.05 < q["mask_fraction"] < 0.95          # (a mask that is all on / all off would make the IoU meaningless)
    del model
    torch.cuda.empty_cache()


def test_c4_shape_fp8_llm_path_vs_bf16(cuda):
    bench, args = _bench_args(["--frames", "16", "--objects", "8"])
    cfg, model, step, ids = _build(bench, args, cuda)
    cap = model.capture = {}
    out16, _ = step()
    model.capture = None
    gen = out16[0].tolist()[ids.shape[1]:]
    assert sum(t == cfg["seg_token_idx"] for t in gen) == 8 and cap["emb"].shape == (8, 256) and cap["logits"].shape == (16, 8, 1024, 1024)
    m16, e16 = (cap["logits"] > 0), cap["emb"].float()
    del model, cap
    torch.cuda.empty_cache()
    cfg8, model8, _, _ = _build(bench, args, cuda, llm_extra=dict(prefill_gemm="fp8", decode_weights="fp8"))
    synth.install_forced_tokens(model8, {i: t for i, t in enumerate(gen)})            # teacher-forced to the bf16 run's ids
    cap8 = model8.capture = {}
    images, context, sam, ids2 = bench.make_inputs(cfg8, args, 1, cuda)
    out8, _ = model8.inference([images], [context], [sam], ids2, [(1024, 1024)], [(args.src, args.src)], max_new_tokens=args.max_new_tokens)
    assert out8[0].tolist() == out16[0].tolist()
    m8 = cap8["logits"] > 0
    iou = (m16 & m8).sum(dim=(0, 2, 3)).double() / (m16 | m8).sum(dim=(0, 2, 3)).double().clamp_min(1)
    cos = torch.nn.functional.cosine_similarity(e16, cap8["emb"].float()).min().item()
    free = [i for i in range(len(gen)) if i not in cfg["forced_tokens"]]
    print(f"C4 shape, fp8 LLM path vs bf16: mask IoU per object {[round(float(v), 4) for v in iou]}, [SEG] embedding cosine {cos:.4f}")
    assert torch.isfinite(cap8["logits"]).all() and cos > 0.99 and iou.mean() > 0.98 and iou.min() > 0.95, (cos, iou.tolist())
    assert len(free) > 0


def test_c1_workload_bf16_vs_fp32_mode(cuda):
    """C1's exact shape (BASELINE.json configs[0], the reference's own demo shape): 8-frame 512^2-source clip, Te = 8 -> a 1697-row prompt (other
    GEMM tile routes than C2's 3361 rows) and 512^2 output masks (vg_bilinear_mask 256^2 -> 512^2), one [SEG]."""
    bench, args = _bench_args(["--frames", "8", "--te", "8", "--src", "512"])
    cfg, model, step, ids = _build(bench, args, cuda)
    assert ids.shape[1] - 8 + 208 * 8 == 1697
    q = bench.quality(cfg, args, model, step, cuda)
    print("C1 whole workload:", q)
    out_ids, segs = step()
    assert out_ids.shape[1] == ids.shape[1] + args.max_new_tokens and sorted(segs[0]) == list(range(8)) and segs[0][0][0].shape == (512, 512)
    assert q["finite"] and q["seg_objects"] == 1
    assert q["mask_miou_vs_fp32"] > 0.99 and q["min_frame_iou_vs_fp32"] > 0.98, q
    assert q["ids_top1_agree"] >= 0.9 and q["seg_emb_cosine"] > 0.999, q
    assert 0.05 < q["mask_fraction"] < 0.95
    del model
    torch.cuda.empty_cache()


def test_c4_clip_64_frames_8_objects(cuda):
    """C4's real clip on one GPU (64 frames, 8 [SEG] objects, framewise branch): 512 (frame, object) mask-decoder instances through the fused
    two-way path in four 128-pair launch groups.  Properties: the framewise branch treats frames independently, so (i) the clip's first 16 frames
    equal a 16-frame clip of the same pixels, (ii) the clip with its frames reversed gives the reversed masks; the LLM side does not see the SAM
    frames, so the ids are those of the 16-frame run."""
    bench, args = _bench_args(["--frames", "64", "--objects", "8"])
    cfg, model, step, ids = _build(bench, args, cuda)
    images, context, sam, ids = bench.make_inputs(cfg, args, 1, cuda)

    def run(frames):
        cap = model.capture = {}
        out, segs = model.inference([images], [context], [frames], ids, [(1024, 1024)], [(1024, 1024)], max_new_tokens=args.max_new_tokens)
        model.capture = None
        return out[0].tolist(), cap["logits"], segs

    ids64, lg64, segs = run(sam)
    gen = ids64[ids.shape[1]:]
    assert sum(t == cfg["seg_token_idx"] for t in gen) == 8 and lg64.shape == (64, 8, 1024, 1024) and torch.isfinite(lg64).all()
    assert sorted(segs[0]) == list(range(64)) and len(segs[0][63]) == 8 and segs[0][63][7].shape == (1024, 1024)
    m64 = lg64 > 0
    frac = m64.float().mean().item()
    assert 0.02 < frac < 0.98, frac
    del lg64
    ids16, lg16, _ = run(sam[:16])
    assert ids16 == ids64
    m16 = lg16 > 0
    inter, union = (m16 & m64[:16]).sum().item(), (m16 | m64[:16]).sum().item()
    assert inter / max(union, 1) > 0.9995, inter / max(union, 1)            # (another chunk / pair-group composition: bf16 summation orders differ)
    del lg16, m16
    _, lgr, _ = run(sam.flip(0))
    mr = (lgr > 0).flip(0)
    inter, union = (mr & m64).sum().item(), (mr | m64).sum().item()
    assert inter / max(union, 1) > 0.9995, inter / max(union, 1)
    del model
    torch.cuda.empty_cache()
