"""Host-side LLM-side graph (videoglamm_amd/vlm.py, model.py) vs reference outputs (golden fixtures).
CPU variant runs on tests/_cpu_ops.py; the -m gpu variant runs the same graph on the HIP kernels (fp32 parity mode)."""
import numpy as np
import pytest
import torch

import _golden as G
from oracle import seeded

torch.set_grad_enabled(False)


def towers(device, sd, cfg):
    from videoglamm_amd.params import Params
    from videoglamm_amd.vlm import VisionTowers

    return VisionTowers(Params(sd, device, torch.float32), cfg)


def ops_decode_row(dec, xrow):
    """one decode step on a given embedding row through the fused decode kernels (what the captured graph runs)."""
    from videoglamm_amd import ops
    h = dec._layers_decode(xrow.contiguous()) if dec.fused_decode else dec._layers(xrow, 0, dec.pos_dev)
    ops.add_int_(dec.pos_dev, 1)
    dec.pos += 1
    return h


def check_modules(device, tol):
    fx = G.fixture("vlm_tiny.npz")
    c = G.configs.IV2_TINY
    sd = {"model.vision_tower.vision_encoder." + k: v for k, v in G.weights("iv2_tiny_manifest.json", 2).items()}
    t = towers(device, sd, dict(iv2=dict(depth=c["depth"], num_heads=c["num_heads"], patch_size=c["patch_size"])))
    out = t.iv2(G.rnd((2, 4, 3, c["img_size"], c["img_size"]), 31).to(device))
    torch.testing.assert_close(out.float().cpu(), fx["iv2_out"][:, 1:], **tol)

    c = G.configs.CLIP_TINY
    sd = {"model.image_vision_tower.vision_tower." + k: v for k, v in G.weights("clip_tiny_manifest.json", 3).items()}
    t = towers(device, sd, dict(clip=dict(num_layers=c["num_layers"], num_heads=c["num_heads"], patch_size=c["patch_size"])))
    out = t.clip(G.rnd((3, 3, c["img_size"], c["img_size"]), 32).to(device))
    torch.testing.assert_close(out.float().cpu(), fx["clip_out"], **tol)

    from videoglamm_amd.params import Params
    from videoglamm_amd.vlm import LlamaDecoder
    c = G.configs.LLAMA_TINY
    sd = {"model." + k: v for k, v in G.weights("llama_tiny_manifest.json", 4).items()}
    x = G.rnd((1, 45, c["hidden"]), 33)[0].to(device)
    dec = LlamaDecoder(Params(sd, device, torch.float32), c, 64)
    torch.testing.assert_close(dec.forward(x).cpu(), fx["llama_out"], **tol)
    # prefill + token-by-token decode through the KV cache must give the same rows
    dec = LlamaDecoder(Params(sd, device, torch.float32), c, 64)
    rows = [dec.forward(x[:40])] + [dec.forward(x[i:i + 1]) for i in range(40, 45)]
    torch.testing.assert_close(torch.cat(rows).cpu(), fx["llama_out"], **tol)

    # Phi-3 layout (the released checkpoint's LLM): fused qkv_proj / gate_up_proj tensors, MHA — vs HF Phi3Model
    c = G.configs.PHI3_TINY
    sd = {"model." + k: v for k, v in G.weights("phi3_tiny_manifest.json", 5).items()}
    x = G.rnd((1, 45, c["hidden"]), 34)[0].to(device)
    ref = G.fixture("phi3_tiny.npz")["phi3_out"]
    dec = LlamaDecoder(Params(sd, device, torch.float32), c, 64)
    torch.testing.assert_close(dec.forward(x).cpu(), ref, **tol)
    dec = LlamaDecoder(Params(sd, device, torch.float32), c, 64)
    rows = [dec.forward(x[:40])]
    if device.type == "cuda":      # the graph-replayed fused decode kernels (norm+GEMV, rope+append+attention+merge)
        for i in range(40, 45):
            hid = ops_decode_row(dec, x[i:i + 1])
            rows.append(hid)
    else:
        rows += [dec.forward(x[i:i + 1]) for i in range(40, 45)]
    torch.testing.assert_close(torch.cat(rows).cpu(), ref, **tol)
    # sliding window crossed by the sequence (position i sees [i - 11, i]): prefill at once, prefill in chunks (Sq < Skv),
    # and decode steps through the windowed decode attention — vs HF Phi3Model with the same mask (phi3_win_out)
    c = G.configs.PHI3_TINY_WIN
    ref = G.fixture("phi3_tiny.npz")["phi3_win_out"]
    dec = LlamaDecoder(Params(sd, device, torch.float32), c, 64)
    torch.testing.assert_close(dec.forward(x).cpu(), ref, **tol)
    dec = LlamaDecoder(Params(sd, device, torch.float32), c, 64)
    rows = [dec.forward(x[:9]), dec.forward(x[9:30])]
    if device.type == "cuda":
        rows += [ops_decode_row(dec, x[i:i + 1]) for i in range(30, 45)]
    else:
        rows += [dec.forward(x[i:i + 1]) for i in range(30, 45)]
    torch.testing.assert_close(torch.cat(rows).cpu(), ref, **tol)


def check_e2e(device, branch, eight=False):
    """eight: C4's shape — eight [SEG] ids in the prompt + the six the model emits = 14 objects (fixture keys *8)."""
    from test_oracle_e2e import check_seam, e2e_setup
    from videoglamm_amd.model import VideoGLaMMForCausalLM

    fx, sd, cfg, inp = e2e_setup()
    key = ("video" if branch else "framewise") + ("8" if eight else "")
    ids_in = (fx["input_ids8"] if eight else inp["input_ids"]).long()[None]
    m = VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.float32, device=device)

    def run():
        out_ids, segs = m.inference([inp["images"]], [inp["context_images"]], [inp["images_for_sam"]], ids_in,
                                    [(1024, 1024)], [inp["original_size"]], max_new_tokens=inp["max_new_tokens"],
                                    use_sam2_video_branch=branch)
        assert out_ids[0].tolist() == fx[f"{key}_output_ids"].long().tolist()        # token ids bit-exact
        seg = segs[0]
        got = np.stack([np.stack([seg[t][k] for k in sorted(seg[t])]) for t in sorted(seg)])
        ref = fx[f"{key}_masks"].numpy() > 0.5
        assert got.shape == ref.shape and (not eight or ref.shape[1] >= 8)
        iou = (got & ref).sum() / (got | ref).sum()
        assert iou > 0.999, iou

    if device.type == "cuda":
        run()                   # the product default: masks straight from the low-res logits (vg_bilinear_mask), graph-replayed propagation
    m.capture = {}              # the same clip with the seam exposed: [SEG] embeddings and logits before the threshold, within 1e-3
    run()
    check_seam(fx, key, m.capture["emb"], m.capture["logits"])


def check_e2e_min_blob(device):
    """min_blob_size: the device-side remove_small_blobs of the thresholded masks == the reference's host-side
    post-processing (eval_gcg_infer.py:182) applied to the reference's own masks."""
    from oracle import postproc as OP
    from test_oracle_e2e import e2e_setup
    from videoglamm_amd.model import VideoGLaMMForCausalLM

    fx, sd, cfg, inp = e2e_setup()
    m = VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.float32, device=device, min_blob_size=20)
    _, segs = m.inference([inp["images"]], [inp["context_images"]], [inp["images_for_sam"]], inp["input_ids"][None],
                          [(1024, 1024)], [inp["original_size"]], max_new_tokens=inp["max_new_tokens"])
    seg = segs[0]
    got = np.stack([np.stack([seg[t][k] for k in sorted(seg[t])]) for t in sorted(seg)])
    ref = fx["framewise_masks"].numpy() > 0.5
    want = np.stack([np.stack([OP.remove_small_blobs(x, 20) for x in fr]) for fr in ref])
    assert got.shape == want.shape
    assert (got & want).sum() / max((got | want).sum(), 1) > 0.999
    assert not (got & ~(ref | (got ^ want))).any()          # nothing is added by the blob filter


def test_e2e_min_blob_cpu(cpu_ops, monkeypatch):
    from videoglamm_amd import _lib
    monkeypatch.setattr(_lib, "load", lambda: None)
    check_e2e_min_blob(torch.device("cpu"))


@pytest.mark.gpu
def test_e2e_min_blob_hip_fp32(cuda):
    check_e2e_min_blob(cuda)


def check_chunked_prefill(device, tol):
    """prefill in two chunks == prefill at once: the second chunk's rows attend causally to keys [0, own position] with
    Sq < Skv — on the MI355X this goes through the split-KV + merge path of vg_attention (few query tiles, long KV), which
    the sequence-parallel multi-GPU prefill (LlamaDecoder.forward_sharded) relies on."""
    from videoglamm_amd.params import Params
    from videoglamm_amd.vlm import LlamaDecoder
    c = G.configs.LLAMA_TINY
    sd = {"model." + k: v for k, v in G.weights("llama_tiny_manifest.json", 4).items()}
    x = G.rnd((1, 600, c["hidden"]), 35)[0].to(device)
    full = LlamaDecoder(Params(sd, device, torch.float32), c, 1024, use_graph=False).forward(x)
    dec = LlamaDecoder(Params(sd, device, torch.float32), c, 1024, use_graph=False)
    parts = torch.cat([dec.forward(x[:300]), dec.forward(x[300:520]), dec.forward(x[520:])])
    torch.testing.assert_close(parts.cpu(), full.cpu(), **tol)


def test_chunked_prefill_cpu(cpu_ops):
    check_chunked_prefill(torch.device("cpu"), dict(rtol=1e-4, atol=1e-4))


@pytest.mark.gpu
def test_chunked_prefill_hip_fp32(cuda):
    check_chunked_prefill(cuda, dict(rtol=1e-3, atol=1e-3))


@pytest.mark.gpu
def test_decode_fp8_weights_hip(cuda):
    """decode step with fp8 (e4m3) weights + row scales vs the bf16 decode step on a 2-layer decoder of Llama-3-8B width: the
    hidden state stays within the quantisation noise of e4m3 weights (cosine > 0.998, relative error < 6 %)."""
    from videoglamm_amd import synth
    from videoglamm_amd.params import Params
    from videoglamm_amd.vlm import LlamaDecoder
    c = dict(synth.LLAMA3_8B, num_layers=2, vocab=4096)
    man = {k: v for k, v in synth.vlm_manifest(dict(synth.videoglamm_llama3_8b(), llm=c)).items()
           if k.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))}
    sd = synth.device_state_dict(man, cuda, torch.bfloat16)
    x = (torch.randn(46, c["hidden"], generator=torch.Generator().manual_seed(3)) * 0.5).to(cuda, torch.bfloat16)
    outs = []
    for mode in ("bf16", "fp8"):
        dec = LlamaDecoder(Params(sd, cuda, torch.bfloat16), dict(c, decode_weights=mode), 1024, use_graph=False)
        dec.forward(x[:39])
        rows = [ops_decode_row(dec, x[39 + i:40 + i]) for i in range(7)]         # seven steps on the same inputs: the error does not build up
        dec.next_token(rows[-1])
        outs.append((torch.cat(rows).float().cpu(), int(dec.tok_dev[0])))
    (hb, tb), (h8, t8) = outs
    cos = torch.nn.functional.cosine_similarity(hb, h8).min().item()
    rel = ((hb - h8).norm(dim=1) / hb.norm(dim=1)).max().item()
    assert cos > 0.998 and rel < 0.06, (cos, rel)     # e4m3 has 3 mantissa bits: ~3.6 % rms per weight, it does not average out of a dot product


@pytest.mark.gpu
def test_prefill_fp8_hip(cuda):
    """fp8 MFMA prefill (per-token activation scales, per-channel weight scales) vs the bf16 prefill on a 2-layer decoder of
    Llama-3-8B width: final-norm states within e4m3's noise, and the KV cache it leaves behind serves a bf16 decode step."""
    from videoglamm_amd import synth
    from videoglamm_amd.params import Params
    from videoglamm_amd.vlm import LlamaDecoder
    c = dict(synth.LLAMA3_8B, num_layers=2, vocab=4096)
    man = {k: v for k, v in synth.vlm_manifest(dict(synth.videoglamm_llama3_8b(), llm=c)).items()
           if k.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))}
    sd = synth.device_state_dict(man, cuda, torch.bfloat16)
    x = (torch.randn(300, c["hidden"], generator=torch.Generator().manual_seed(4)) * 0.5).to(cuda, torch.bfloat16)
    hs = []
    for mode in ("bf16", "fp8"):
        dec = LlamaDecoder(Params(sd, cuda, torch.bfloat16), dict(c, prefill_gemm=mode), 1024, use_graph=False)
        h = dec.forward(x[:299])
        hs.append(torch.cat([h, ops_decode_row(dec, x[299:300])]).float().cpu())
    cos = torch.nn.functional.cosine_similarity(hs[0], hs[1]).min().item()
    rel = ((hs[0] - hs[1]).norm(dim=1) / hs[0].norm(dim=1)).max().item()
    assert cos > 0.995 and rel < 0.10, (cos, rel)


def test_modules_cpu(cpu_ops):
    check_modules(torch.device("cpu"), dict(rtol=1e-4, atol=1e-4))


@pytest.mark.parametrize("branch", [False, True])
def test_e2e_cpu(cpu_ops, branch, monkeypatch):
    from videoglamm_amd import _lib
    monkeypatch.setattr(_lib, "load", lambda: None)
    check_e2e(torch.device("cpu"), branch)


def test_e2e_8_objects_cpu(cpu_ops, monkeypatch):
    from videoglamm_amd import _lib
    monkeypatch.setattr(_lib, "load", lambda: None)
    check_e2e(torch.device("cpu"), False, eight=True)     # (the 14-object video branch on the CPU twins takes minutes: -m gpu only)


@pytest.mark.gpu
def test_modules_hip_fp32(cuda):
    check_modules(cuda, dict(rtol=1e-3, atol=1e-3))


@pytest.mark.gpu
@pytest.mark.parametrize("branch", [False, True])
def test_e2e_hip_fp32(cuda, branch):
    check_e2e(cuda, branch)


@pytest.mark.gpu
@pytest.mark.parametrize("branch", [False, True])
def test_e2e_8_objects_hip_fp32(cuda, branch):
    """C4's object count on the HIP kernels, both branches, vs the reference's own inference(): ids exact, [SEG] embeddings and
    mask logits within 1e-3, masks IoU > 0.999."""
    check_e2e(cuda, branch, eight=True)


def check_e2e_image(device):
    """single-image prompt (context_images=None) vs the reference's own inference() — tests/golden/e2e_image.npz"""
    from test_oracle_e2e import image_setup
    from videoglamm_amd.model import VideoGLaMMForCausalLM

    fx, sd, cfg, inp = image_setup()
    m = VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.float32, device=device)
    out_ids, segs = m.inference([inp["images"]], None, [inp["images_for_sam"]], inp["input_ids"][None], [(1024, 1024)],
                                [inp["original_size"]], max_new_tokens=inp["max_new_tokens"])
    assert out_ids[0].tolist() == fx["output_ids"].long().tolist()
    seg = segs[0]
    got = np.stack([np.stack([seg[t][k] for k in sorted(seg[t])]) for t in sorted(seg)])
    ref = fx["masks"].numpy() > 0.5
    assert got.shape == ref.shape and (got & ref).sum() / (got | ref).sum() > 0.999


def test_e2e_image_cpu(cpu_ops, monkeypatch):
    from videoglamm_amd import _lib
    monkeypatch.setattr(_lib, "load", lambda: None)
    check_e2e_image(torch.device("cpu"))


@pytest.mark.gpu
def test_e2e_image_hip_fp32(cuda):
    check_e2e_image(cuda)
