"""The SAM2 oracle (oracle/sam2.py) vs outputs of the REFERENCE itself (tests/golden/sam2_micro.npz,
made by tests/golden/make_golden.py from /root/reference on the same name-seeded weights/inputs)."""
import torch

import _golden as G
from oracle import sam2 as O
from oracle import seeded

torch.set_grad_enabled(False)
TOL = dict(rtol=1e-4, atol=2e-4)


def setup_module(m):
    m.fx = G.fixture("sam2_micro.npz")
    m.sd = G.weights("sam2_micro_manifest.json", 1, seeded.sam2_overrides())
    m.cfg = G.sam2_cfg()
    T, N, H, W = [int(v) for v in m.fx["meta"]]
    m.T, m.N, m.H, m.W = T, N, H, W
    S = m.cfg["image_size"]
    m.images = G.rnd((T, 3, S, S), 11)
    m.text = G.rnd((N, 256), 12, 0.5)


def test_forward_image():
    fpn, pos = O.forward_image(sd, "", cfg, images[0:1])
    for i in range(3):
        torch.testing.assert_close(fpn[i], fx[f"fpn{i}"], **TOL)
    torch.testing.assert_close(pos[2], fx["pos2"], **TOL)
    torch.testing.assert_close(O.dense_pe(sd, "", (16, 16)), fx["dense_pe"], **TOL)


def test_mask_decoder_all_tokens():
    fpn, _ = O.forward_image(sd, "", cfg, images[T - 1:T])
    emb = fpn[-1] + sd["no_mem_embed"].view(1, 256, 1, 1)
    sparse, dense = O.prompt_encoder(sd, "", cfg, N, text.unsqueeze(1), False)
    masks, iou, tok, obj = O.mask_decoder_predict(sd, "", emb, O.dense_pe(sd, "", (16, 16)), sparse, dense, True, fpn[:-1])
    torch.testing.assert_close(masks, fx["dec_masks4"], **TOL)
    torch.testing.assert_close(iou, fx["dec_iou4"], **TOL)
    torch.testing.assert_close(tok, fx["dec_tokens4"], **TOL)
    torch.testing.assert_close(obj, fx["dec_obj"], **TOL)
    assert (obj > 0).all(), "fixture must exercise the object-present path"


def test_framewise_branch():
    logits, low = O.framewise_branch(sd, "", cfg, images, text, (H, W))
    torch.testing.assert_close(low, fx["framewise_low"], **TOL)
    torch.testing.assert_close(torch.stack(logits), fx["framewise_logits"], **TOL)


def test_memory_attention_and_encoder():
    hw = 256
    out = O.memory_attention(sd, "", G.rnd((hw, N, 256), 21), G.rnd((hw, N, 256), 22), G.rnd((2 * hw + 8, N, 64), 23),
                             G.rnd((2 * hw + 8, N, 64), 24), 8)
    torch.testing.assert_close(out, fx["memattn_out"], **TOL)
    S = cfg["image_size"]
    feat, pos = O.memory_encoder(sd, "", G.rnd((N, 256, 16, 16), 25), torch.sigmoid(G.rnd((N, 1, S, S), 26, 3.0)) * 20 - 10)
    torch.testing.assert_close(feat, fx["memenc_feat"], **TOL)
    torch.testing.assert_close(pos, fx["memenc_pos"], **TOL)


def test_video_branch():
    vid, trace = O.video_branch(sd, "", cfg, images, text, (H, W))
    assert (trace["frame0_obj_logits"] > 0).all()
    torch.testing.assert_close(trace["low_res"], fx["video_low_res"], rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(torch.stack(vid), fx["video_logits"], rtol=1e-3, atol=1e-3)
    # the masks the user finally sees
    assert ((torch.stack(vid) > 0) == (fx["video_logits"] > 0)).float().mean() > 0.9999


def test_video_branch_long():
    """T = 18, N = 2 (tests/golden/sam2_video_long.npz): the 7-slot memory bank rolls over from frame 8 and the 16-pointer window
    from frame 17 (R/.../sam2_base.py:536-633); every frame's low-res logits, object pointers and bf16-rounded memories vs the
    reference, and the T = 9 clip as a prefix (the generator checks the same property on the reference itself)."""
    from make_golden_keys import LONG_MEM_FRAMES
    fxl = G.fixture("sam2_video_long.npz")
    Tl, Nl, Hl, Wl = [int(v) for v in fxl["meta"]]
    S = cfg["image_size"]
    imgs, txt = G.rnd((Tl, 3, S, S), 41), G.rnd((Nl, 256), 42, 0.5)
    vid, trace = O.video_branch(sd, "", cfg, imgs, txt, (Hl, Wl))
    tol = dict(rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(trace["low_res"], fxl["low_res"], **tol)
    torch.testing.assert_close(trace["obj_ptr"], fxl["obj_ptr"], **tol)
    torch.testing.assert_close(torch.stack(vid)[:, :, 0], fxl["video_logits"], **tol)
    for t in LONG_MEM_FRAMES:
        # memories are stored in bf16 (sam2_video_predictor.py:967,1011): a value on a rounding boundary may land one bf16 step apart
        torch.testing.assert_close(trace["maskmem"][t], fxl[f"maskmem_{t}"], rtol=1e-2, atol=2e-3)
    vid9, _ = O.video_branch(sd, "", cfg, imgs[:9], txt, (Hl, Wl))
    torch.testing.assert_close(torch.stack(vid9)[:, :, 0], fxl["video_logits"][:9], **tol)


def test_video_branch_no_object():
    """objects that disappear and come back (tests/golden/sam2_noobj.npz, T = 9, N = 2; presence pattern per frame
    [[1,0],[0,1],[0,0],[0,0],[1,1],[0,0],[0,0],[1,1],[0,1]]): NO_OBJ_SCORE fill, no_obj_ptr mix, the -1024 mask through the
    upsample + memory encoder — R/.../sam2_base.py:355-364,390-401, sam2_video_predictor.py:571-612."""
    fxn = G.fixture("sam2_noobj.npz")
    Tn, Nn, Hn, Wn = [int(v) for v in fxn["meta"]]
    sdn = G.weights("sam2_micro_manifest.json", 1, seeded.sam2_noobj_overrides(float(fxn["score_c"]), float(fxn["score_k"])))
    S = cfg["image_size"]
    imgs, txt = G.rnd((Tn, 3, S, S), 71), G.rnd((Nn, 256), 72, 0.5)
    vid, trace = O.video_branch(sdn, "", cfg, imgs, txt, (Hn, Wn))
    scores = torch.stack([trace["frame0_obj_logits"].view(-1)] + [trace[f"obj_logits_{t}"].view(-1) for t in range(1, Tn)])
    pres = fxn["obj_scores"] > 0
    assert pres.any() and (~pres).any() and (pres[:, 0] != pres[:, 1]).any(), "fixture must mix present and absent objects"
    assert torch.equal(scores > 0, pres)
    tol = dict(rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(scores, fxn["obj_scores"], rtol=1e-3, atol=2e-3)
    torch.testing.assert_close(trace["low_res"], fxn["low_res"], **tol)
    assert (trace["low_res"][~pres] == O.NO_OBJ_SCORE).all()
    torch.testing.assert_close(trace["obj_ptr"], fxn["obj_ptr"], **tol)
    torch.testing.assert_close(torch.stack(vid)[:, :, 0], fxn["video_logits"], **tol)
    for t in range(Tn):
        torch.testing.assert_close(trace["maskmem"][t], fxn[f"maskmem_{t}"], rtol=1e-2, atol=2e-3)
