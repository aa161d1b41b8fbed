"""Host-side SAM2 graph (videoglamm_amd/sam2.py) vs the reference's outputs (golden fixtures).

CPU variant: operators swapped for tests/_cpu_ops.py (checks the graph logic / layouts / weight packing).
GPU variant (-m gpu): the same graph on the HIP kernels, fp32 parity mode, tolerance 1e-3 on mask logits.
"""
import pytest
import torch

import _golden as G
from oracle import seeded

torch.set_grad_enabled(False)


def build(device, dtype=torch.float32):
    from videoglamm_amd.params import Params
    from videoglamm_amd.sam2 import SAM2

    sd = G.weights("sam2_micro_manifest.json", 1, seeded.sam2_overrides())
    return SAM2(Params(sd, device, dtype), "", G.sam2_cfg())


def run_checks(device, tol):
    fx = G.fixture("sam2_micro.npz")
    T, N, H, W = [int(v) for v in fx["meta"]]
    m = build(device)
    S = m.S
    images = G.rnd((T, 3, S, S), 11).to(device)
    text = G.rnd((N, 256), 12, 0.5).to(device)
    nchw = lambda x, h: x.view(x.shape[0], h, h, -1).permute(0, 3, 1, 2).float().cpu()  # noqa: E731
    fpn = m.forward_image(images[0:1])
    torch.testing.assert_close(nchw(fpn[0], S // 4), fx["fpn0"], **tol)
    torch.testing.assert_close(nchw(fpn[1], S // 8), fx["fpn1"], **tol)
    torch.testing.assert_close(nchw(fpn[2], S // 16), fx["fpn2"], **tol)
    torch.testing.assert_close(m.dense_pe().float().cpu().view(16, 16, 256).permute(2, 0, 1)[None], fx["dense_pe"], **tol)
    # mask decoder, all 4 tokens, last frame (the fixture's frame)
    fpn = m.forward_image(images[T - 1:T])
    from videoglamm_amd import ops
    emb = ops.add(fpn[2].view(1, 256, 256), m.P.t("no_mem_embed").view(-1))
    masks, iou, toks, obj = m.mask_decoder(emb, m.sparse_prompt(N, text.unsqueeze(1), False), (fpn[0], fpn[1]), True)
    torch.testing.assert_close(masks.cpu(), fx["dec_masks4"], **tol)
    torch.testing.assert_close(iou.cpu(), fx["dec_iou4"], **tol)
    torch.testing.assert_close(toks.float().cpu(), fx["dec_tokens4"], **tol)
    torch.testing.assert_close(obj.cpu(), fx["dec_obj"], **tol)
    # framewise branch
    logits, low = m.framewise_branch(images, text, (H, W))
    torch.testing.assert_close(low.cpu(), fx["framewise_low"], **tol)
    torch.testing.assert_close(logits.cpu(), fx["framewise_logits"], **tol)
    # memory attention / encoder in isolation
    hw = 256
    tr = lambda x: x.transpose(0, 1).contiguous().to(device)  # noqa: E731
    curr, cpos = G.rnd((hw, N, 256), 21), G.rnd((hw, N, 256), 22)
    mem, mpos = G.rnd((2 * hw + 8, N, 64), 23), G.rnd((2 * hw + 8, N, 64), 24)
    m.vision_pos = lambda: cpos[:, 0].contiguous().to(device)  # the isolated fixture used a random curr_pos
    assert torch.equal(cpos[:, 0], cpos[:, 0])  # (pos differs per batch column in the fixture -> compare column 0 only)
    out = m.memory_attention(tr(curr)[:1], tr(mem)[:1], tr(mpos)[:1], 8)
    torch.testing.assert_close(out[0].float().cpu(), fx["memattn_out"][:, 0], **tol)
    del m.vision_pos
    pix, msk = G.rnd((N, 256, 16, 16), 25), G.rnd((N, 1, S, S), 26, 3.0)
    feat = m.memory_encoder(pix.permute(0, 2, 3, 1).contiguous().to(device), (torch.sigmoid(msk) * 20 - 10).view(N, S, S, 1).to(device))
    torch.testing.assert_close(feat.float().cpu().view(N, 16, 16, 64).permute(0, 3, 1, 2), fx["memenc_feat"], **tol)
    torch.testing.assert_close(m.maskmem_pos().float().cpu().view(16, 16, 64).permute(2, 0, 1), fx["memenc_pos"][0], **tol)
    # video branch
    trace = {}
    vid = m.video_branch(images, text, (H, W), trace)
    assert (trace["frame0_obj_logits"] > 0).all()
    t2 = dict(rtol=max(tol["rtol"], 1e-3), atol=max(tol["atol"], 1e-3))
    torch.testing.assert_close(trace["low_res"].cpu(), fx["video_low_res"], **t2)
    torch.testing.assert_close(vid.cpu(), fx["video_logits"][:, :, 0], **t2)
    assert ((vid.cpu() > 0) == (fx["video_logits"][:, :, 0] > 0)).float().mean() > 0.9995


def run_long(device, tol):
    """T = 18 / T = 9 video branch vs the reference (tests/golden/sam2_video_long.npz): memory-bank roll-over, pointer window."""
    from make_golden_keys import LONG_MEM_FRAMES
    fx = G.fixture("sam2_video_long.npz")
    T, N, H, W = [int(v) for v in fx["meta"]]
    m = build(device)
    images, text = G.rnd((T, 3, m.S, m.S), 41).to(device), G.rnd((N, 256), 42, 0.5).to(device)
    trace = {}
    vid = m.video_branch(images, text, (H, W), trace)
    torch.testing.assert_close(trace["low_res"].cpu(), fx["low_res"], **tol)
    torch.testing.assert_close(trace["obj_ptr"].float().cpu(), fx["obj_ptr"], **tol)
    torch.testing.assert_close(vid.cpu(), fx["video_logits"], **tol)
    for t in LONG_MEM_FRAMES:       # [N, es*es, 64] token-major -> the reference's [N,64,es,es]; bf16-stored: one rounding step of slack
        got = trace["maskmem"][t].float().cpu().view(N, 16, 16, 64).permute(0, 3, 1, 2)
        torch.testing.assert_close(got, fx[f"maskmem_{t}"], rtol=1e-2, atol=2e-3)
    vid9 = m.video_branch(images[:9], text, (H, W))
    torch.testing.assert_close(vid9.cpu(), fx["video_logits"][:9], **tol)


def run_noobj(device, tol):
    """objects that disappear and come back (tests/golden/sam2_noobj.npz): where_rows row selection, the NO_OBJ_SCORE fill through
    the bilinear upsample + memory encoder, the no_obj_ptr mix — R/.../sam2_base.py:355-364,390-401."""
    from videoglamm_amd.params import Params
    from videoglamm_amd.sam2 import SAM2, NO_OBJ_SCORE
    fx = G.fixture("sam2_noobj.npz")
    T, N, H, W = [int(v) for v in fx["meta"]]
    sd = G.weights("sam2_micro_manifest.json", 1, seeded.sam2_noobj_overrides(float(fx["score_c"]), float(fx["score_k"])))
    m = SAM2(Params(sd, device, torch.float32), "", G.sam2_cfg())
    images, text = G.rnd((T, 3, m.S, m.S), 71).to(device), G.rnd((N, 256), 72, 0.5).to(device)
    trace = {}
    vid = m.video_branch(images, text, (H, W), trace)
    scores = torch.stack([trace["frame0_obj_logits"].view(-1)] + [trace[f"obj_logits_{t}"].view(-1) for t in range(1, T)]).cpu()
    pres = fx["obj_scores"] > 0
    assert pres.any() and (~pres).any() and (pres[:, 0] != pres[:, 1]).any()
    assert torch.equal(scores > 0, pres)
    torch.testing.assert_close(scores, fx["obj_scores"], rtol=1e-3, atol=2e-3)
    low = trace["low_res"].cpu()
    torch.testing.assert_close(low, fx["low_res"], **tol)
    assert (low[~pres] == NO_OBJ_SCORE).all()
    torch.testing.assert_close(trace["obj_ptr"].float().cpu(), fx["obj_ptr"], **tol)
    torch.testing.assert_close(vid.cpu(), fx["video_logits"], **tol)
    for t in range(T):
        got = trace["maskmem"][t].float().cpu().view(N, 16, 16, 64).permute(0, 3, 1, 2)
        torch.testing.assert_close(got, fx[f"maskmem_{t}"], rtol=1e-2, atol=2e-3)
    return m, images, text, (H, W), vid


def test_video_noobj_cpu(cpu_ops):
    run_noobj(torch.device("cpu"), dict(rtol=1e-3, atol=1e-3))


@pytest.mark.gpu
def test_video_noobj_hip_fp32(cuda):
    m, images, text, hw, vid = run_noobj(cuda, dict(rtol=1e-3, atol=1e-3))
    # the graph-replayed propagation (the product default) takes the same present / absent decisions on the device
    feats = m.hiera_frames(images)
    assert torch.equal(m.video_branch_graphed(images, text, hw, feats), vid)
    masks = m.video_branch_graphed(images, text, hw, feats, as_masks=True)
    assert torch.equal(masks.bool(), vid > 0)


def test_video_long_cpu(cpu_ops):
    run_long(torch.device("cpu"), dict(rtol=1e-3, atol=1e-3))


@pytest.mark.gpu
def test_video_long_hip_fp32(cuda):
    run_long(cuda, dict(rtol=1e-3, atol=1e-3))


def test_host_graph_cpu(cpu_ops):
    run_checks(torch.device("cpu"), dict(rtol=1e-4, atol=2e-4))


def test_host_graph_fused_two_way_cpu(cpu_ops, monkeypatch):
    """the fused form of the two-way transformer's image side (sam2.py: _two_way_fused — the token -> image attention as ONE head-dim-256
    attention over 8 nt derived queries, the image -> token attention as two small GEMMs + per-head softmax + LayerNorm) against the SAME reference
    fixtures: the algebra is exact, only the order of the fp32 roundings differs."""
    monkeypatch.setenv("VG_TWOWAY_FUSED", "2")
    run_checks(torch.device("cpu"), dict(rtol=1e-3, atol=1e-3))


def test_host_graph_grouped_heads_cpu(cpu_ops, monkeypatch):
    """r06: the mask decoder's three-layer heads through ops.mlp3_grouped (stacked parameters, the heads' inputs as strided views of the output tokens) —
    the bf16 mode's route, forced here in fp32 on the CPU twins against the reference fixtures of the framewise graph and of the no-object video clip."""
    from videoglamm_amd.sam2 import SAM2
    monkeypatch.setattr(SAM2, "_heads_grouped", lambda self: True)
    run_checks(torch.device("cpu"), dict(rtol=1e-4, atol=2e-4))
    run_noobj(torch.device("cpu"), dict(rtol=1e-3, atol=1e-3))


def test_video_reference_order_of_operations_cpu(cpu_ops, monkeypatch):
    """r04's / r05's reformulations of the video branch are algebra, not approximation: with every one of them switched OFF (v-projection applied to
    the memory rows in front of the attention, the mask downsampler as im2col + GEMM + LayerNorm + GELU launches, separate q / k / v projections,
    memories and pointers concatenated per frame in the reference's key order instead of living in the bank, separate W_v and W_out) the
    host graph is the reference's own order of operations — and the no-object clip (T = 9, N = 2, the reference's outputs) must come out the same,
    within the same 1e-3, as with them switched ON (the default, exercised by test_video_noobj_cpu / test_video_long_cpu)."""
    from videoglamm_amd import sam2
    for name in ("_MEMATTN_LOWRANK", "_MEMENC_FUSED", "_SELFATTN_FUSED", "_MEMBANK"):
        assert getattr(sam2, name) is True
        monkeypatch.setattr(sam2, name, False)
    run_noobj(torch.device("cpu"), dict(rtol=1e-3, atol=1e-3))


@pytest.mark.gpu
def test_hip_parity_fp32(cuda):
    """mask logits within 1e-3 of the reference in fp32 mode (BASELINE.md target)."""
    run_checks(cuda, dict(rtol=1e-3, atol=1e-3))


@pytest.mark.gpu
def test_hip_bf16_mask_miou(cuda):
    """bf16 performance mode: mask IoU vs the reference (IoU = sum(and)/sum(or), R/eval_gcg_metrics.py:26-35) on the
    framewise and video branches of the micro fixture.  Reported, and required to stay high.

    The framewise branch ends in a DISCRETE choice per mask (multimask_output=False + dynamic stability: token 0 or the
    best-IoU token, sam/mask_decoder.py:257-295).  The fp32 mode reproduces every choice (test_hip_parity_fp32); at bf16
    a stability score next to its 0.98 threshold can land on either side, and on this random-weight fixture a flipped
    choice swaps a whole mask (IoU ~0.85 for that one).  So the gate is the typical mask (median per-mask IoU), a cap on
    flipped choices, and the mean."""
    from videoglamm_amd.host import mask_iou
    from videoglamm_amd.params import Params
    from videoglamm_amd.sam2 import SAM2

    fx = G.fixture("sam2_micro.npz")
    T, N, H, W = [int(v) for v in fx["meta"]]
    sd = G.weights("sam2_micro_manifest.json", 1, seeded.sam2_overrides())
    m = SAM2(Params(sd, cuda, torch.bfloat16), "", G.sam2_cfg())
    images = G.rnd((T, 3, m.S, m.S), 11).to(cuda)
    text = G.rnd((N, 256), 12, 0.5).to(cuda)
    fw, _ = m.framewise_branch(images, text, (H, W))
    vid = m.video_branch(images, text, (H, W))
    iou_fw = mask_iou((fw.cpu() > 0).numpy(), (fx["framewise_logits"] > 0).numpy())
    iou_vid = mask_iou((vid.cpu() > 0).numpy(), (fx["video_logits"][:, :, 0] > 0).numpy())
    a, b = (fw.cpu() > 0).numpy(), (fx["framewise_logits"] > 0).numpy()
    per = sorted(float((a[t, n] & b[t, n]).sum() / max((a[t, n] | b[t, n]).sum(), 1)) for t in range(T) for n in range(N))
    med, flipped = per[len(per) // 2], sum(v < 0.95 for v in per)
    print(f"bf16 mask mIoU vs reference: framewise {iou_fw:.4f} (median per-mask {med:.4f}, {flipped}/{len(per)} choices flipped), "
          f"video branch {iou_vid:.4f}")
    assert med > 0.99 and flipped <= len(per) // 3 and iou_fw > 0.90 and iou_vid > 0.95, (per, iou_vid)


@pytest.mark.gpu
def test_hip_bf16_fused_two_way_equals_unfused(cuda, monkeypatch):
    """bf16: the fused image side of the mask decoder's two-way transformer (r04 default: head-dim-256 attention for token -> image,
    vg_twoway_image_update for image -> token) against the unfused kernel chain on the same weights — mask logits of all four tokens, IoU heads
    and output tokens within bf16 noise of each other, and both equally far from the fp32 reference fixture."""
    from videoglamm_amd import ops
    from videoglamm_amd.params import Params
    from videoglamm_amd.sam2 import SAM2

    fx = G.fixture("sam2_micro.npz")
    T, N, H, W = [int(v) for v in fx["meta"]]
    sd = G.weights("sam2_micro_manifest.json", 1, seeded.sam2_overrides())
    m = SAM2(Params(sd, cuda, torch.bfloat16), "", G.sam2_cfg())
    images, text = G.rnd((T, 3, m.S, m.S), 11).to(cuda), G.rnd((N, 256), 12, 0.5).to(cuda)
    fpn = m.forward_image(images[T - 1:T])
    emb = ops.add(fpn[2].view(1, 256, 256), m.P.t("no_mem_embed").view(-1))
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VG_TWOWAY_FUSED", mode)
        for with_pts in (False, True):          # 7 tokens (framewise) / 9 tokens (video branch: two not-a-point tokens)
            masks, iou, toks, obj = m.mask_decoder(emb, m.sparse_prompt(N, text.unsqueeze(1), with_pts), (fpn[0], fpn[1]), True)
            out[mode, with_pts] = (masks.float().cpu(), iou.float().cpu(), toks.float().cpu())
    ref = fx["dec_masks4"]
    for with_pts in (False, True):
        (m0, i0, t0), (m1, i1, t1) = out["0", with_pts], out["1", with_pts]
        scale = m0.abs().max().item()
        d01 = (m0 - m1).abs().max().item()
        print(f"two-way fused vs unfused (bf16, {'9' if with_pts else '7'} tokens): max |d logit| {d01:.4f} at |logit| <= {scale:.2f}; "
              f"sign agreement {((m0 > 0) == (m1 > 0)).float().mean().item():.5f}")
        assert d01 < 0.06 * scale + 0.05
        assert ((m0 > 0) == (m1 > 0)).float().mean() > 0.995
        torch.testing.assert_close(i1, i0, rtol=5e-2, atol=5e-2)
        torch.testing.assert_close(t1, t0, rtol=5e-2, atol=8e-2)
    e0, e1 = (out["0", False][0] - ref).abs().max().item(), (out["1", False][0] - ref).abs().max().item()
    print(f"distance to the fp32 reference: unfused {e0:.4f}, fused {e1:.4f}")
    assert e1 < 2.0 * e0 + 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_video_branch_graph_replay(cuda, dtype):
    """The HIP-graph replay of the propagation (the default on the GPU: videoglamm_amd/model.py) issues the same launches as the eager
    loop: its logits / uint8 masks are bit-identical, on the capturing call, on a replay with new inputs, and across the
    least-recently-used cache of configurations."""
    from videoglamm_amd.params import Params
    from videoglamm_amd.sam2 import SAM2

    fx = G.fixture("sam2_micro.npz")
    T, N, H, W = [int(v) for v in fx["meta"]]
    sd = G.weights("sam2_micro_manifest.json", 1, seeded.sam2_overrides())
    m = SAM2(Params(sd, cuda, dtype), "", G.sam2_cfg())
    for seed in (11, 31):                                   # first call captures, second replays with other inputs
        images = G.rnd((T, 3, m.S, m.S), seed).to(cuda)
        text = G.rnd((N, 256), seed + 1, 0.5).to(cuda).to(dtype)
        feats = m.hiera_frames(images)
        eager = m.video_branch(images, text, (H, W), frame_feats=feats)
        graphed = m.video_branch_graphed(images, text, (H, W), feats)
        assert torch.equal(eager, graphed)
        em = m.video_branch(images, text, (H, W), frame_feats=feats, as_masks=True)
        gm = m.video_branch_graphed(images, text, (H, W), feats, as_masks=True)
        assert gm.dtype == torch.uint8 and torch.equal(em, gm)
    for n in (1, N):                                        # more configurations than the cache keeps: evictions, re-captures
        for hw in ((H, W), (H + 8, W), (H, W + 8)):
            out = m.video_branch_graphed(images, text[:n], hw, feats)
            assert torch.equal(out, m.video_branch(images, text[:n], hw, frame_feats=feats))
    assert len(m._video_graphs) <= 4


def test_mlp3_pack_layout_cpu():
    """vg_mlp3_grouped's weight packing (ops.mlp3_pack; the layout stated in include/vg_kernels.h): element (g, t, s, lane, e) = W_g[32 t + lane % 32][16 s + 8 (lane / 32) + e],
    rows past `out` zero — checked element-wise on the product's packer, and as a round trip through the CPU twin's unpacker."""
    import _cpu_ops
    from videoglamm_amd import ops
    G, n_out, n_in = 2, 40, 48
    w = torch.arange(G * n_out * n_in, dtype=torch.float32).view(G, n_out, n_in)
    p = ops.mlp3_pack(w)
    assert p.shape == (G, 2, n_in // 16, 64, 8)
    for g, t, s, lane, e in ((0, 0, 0, 0, 0), (1, 1, 2, 37, 5), (0, 1, 1, 7, 7), (1, 0, 2, 63, 3)):
        r, c = 32 * t + lane % 32, 16 * s + 8 * (lane // 32) + e
        assert float(p[g, t, s, lane, e]) == (float(w[g, r, c]) if r < n_out else 0.0)
    assert torch.equal(_cpu_ops._mlp3_unpack(p, n_out), w) and torch.equal(_cpu_ops.mlp3_pack(w), p)
