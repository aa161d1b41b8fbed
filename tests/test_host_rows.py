"""Host rows H1/H2/H4 (videoglamm_amd/host.py) vs the reference's own helper functions (tests/golden/host_rows.npz)."""
import numpy as np
import torch

import _golden as G
from videoglamm_amd import host


class ToyTokenizer:
    bos_token_id = 1

    def __init__(self):
        self.vocab = {}

    def __call__(self, text):
        ids = [self.bos_token_id]
        for w in text.replace("\n", " \n ").split(" "):
            if w == "":
                continue
            ids.append(self.vocab.setdefault(w, 10 + len(self.vocab)))
        return type("Enc", (), {"input_ids": ids})()


def test_sam_preprocess_matches_reference():
    fx = G.fixture("host_rows.npz")
    frame = np.random.RandomState(7).randint(0, 256, size=(60, 80, 3)).astype(np.uint8)
    x, shape = host.sam_preprocess(frame)
    assert list(shape) == fx["sam_pre_shape"].long().tolist()
    torch.testing.assert_close(x[:, ::16, ::16], fx["sam_pre_sub"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(x.mean(dim=(1, 2)), fx["sam_pre_mean"], rtol=1e-5, atol=1e-5)


def test_clip_preprocess_matches_reference():
    """CLIP stream of H1 vs the reference's EncPreprocessor_VideoGPTPlus.preprocess run on transformers' CLIPImageProcessor
    (constructed offline with the hub checkpoint's preprocessor values, make_golden.py:gen_host): up-scaling, down-scaling with a
    centre crop on both axes, identity, pad-by-repeat to num_frames.  1e-6: fp32 rounding of (x/255 - mean)/std only."""
    fx = G.fixture("host_rows.npz")
    g = np.random.RandomState(7)
    g.randint(0, 256, size=(60, 80, 3))                                   # (the SAM frame drawn first by the generator)
    frames = [g.randint(0, 256, size=s).astype(np.uint8) for s in ((60, 80, 3), (500, 400, 3), (336, 336, 3))]
    ctx = torch.stack([host.clip_preprocess(f) for f in host.pad_or_truncate(frames, 4)])
    assert ctx.shape == (4, 3, 336, 336)
    tol = dict(rtol=0, atol=1e-6)
    torch.testing.assert_close(ctx[:, :, ::7, ::7], fx["clip_pre_sub"], **tol)
    torch.testing.assert_close(ctx[:, :, 100:132, 200:232], fx["clip_pre_patch"], **tol)
    torch.testing.assert_close(ctx.mean(dim=(2, 3)), fx["clip_pre_mean"], rtol=0, atol=1e-6)


def test_prompt_and_image_tokens_match_reference():
    fx = G.fixture("host_rows.npz")
    tok = ToyTokenizer()
    for key in ("phi3", "llama3_1"):
        ids = host.apply_for_chat("Please segment the red car .", tok, num_frames=4, base_type=key)
        assert ids.shape[0] == 1
        assert ids[0].tolist() == fx[f"ids_{key}"].long().tolist()
        assert (ids == host.IMAGE_TOKEN_INDEX).sum() == 4


def test_frame_sampling_and_padding():
    frames = list(range(40))
    assert host.subsample_frames(frames, 16) == [frames[i] for i in np.linspace(0, 39, 16, dtype=int)]
    assert host.pad_or_truncate([1, 2, 3], 6) == [1, 2, 3, 3, 3, 3]
    assert host.pad_or_truncate(list(range(9)), 4) == [0, 1, 2, 3]


def test_preprocess_vision_matches_reference_positions():
    """The five return values of preprocess_vision, position by position, against the reference's own chat.preprocess_vision called the way
    R/chat.py:540-553 calls it (tests/golden/host_pv.npz; make_golden.py:gen_host_pv): (enc_image, enc_context_image, image_sam,
    original_size_list, resize_list) — r04 returned the last two swapped.  6 frames > NUM_FRAMES = 4: sub-sampled for the encoders, all for SAM."""
    fx = G.fixture("host_pv.npz")
    frames = [np.random.RandomState(20 + i).randint(0, 256, size=(48, 64, 3)).astype(np.uint8) for i in range(6)]
    cg = host.ConvGenerator_VideoGPTPlus(False, "phi3", num_frames=4)
    ret = host.preprocess_vision([list(frames)], type="video", enc_preprocessor=host.EncPreprocessor_VideoGPTPlus(4),
                                 sam_preprocessor=host.SAM_v2_Preprocess(), conv_generator=cg, precision="fp32")
    assert len(ret) == 5
    enc_image, enc_context_image, image_sam, original_size_list, resize_list = (r if r is None or not torch.is_tensor(r[0]) else [r[0].cpu()] for r in ret)
    assert list(enc_image[0].shape) == fx["video_pos0_shape"].long().tolist() and enc_image[0].dtype == torch.float32
    torch.testing.assert_close(enc_context_image[0][:, :, ::7, ::7], fx["video_pos1_sub"], rtol=0, atol=1e-6)
    torch.testing.assert_close(enc_context_image[0].mean(dim=(2, 3)), fx["video_pos1_mean"], rtol=0, atol=1e-6)
    torch.testing.assert_close(image_sam[0][:, :, ::16, ::16], fx["video_pos2_sub"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(image_sam[0].mean(dim=(2, 3)), fx["video_pos2_mean"], rtol=1e-5, atol=1e-5)
    assert [list(x) for x in original_size_list] == fx["video_pos3"].long().tolist() == [[48, 64]]
    assert [list(x) for x in resize_list] == fx["video_pos4"].long().tolist() == [[768, 1024]]
    # default arguments = the reference's objects; bf16 / fp16 cast like R/chat.py:437 (fp16 -> bf16 on this build)
    d = host.preprocess_vision([list(frames)], conv_generator=cg, precision="bf16")
    assert d[0][0].dtype == d[1][0].dtype == d[2][0].dtype == torch.bfloat16 and d[3] == [(48, 64)] and d[4] == [(768, 1024)]
    assert torch.equal(d[2][0].cpu(), image_sam[0].bfloat16())
    # type="image" (R/chat.py:458-487): the CLIP tensor at position 0, None at position 1, one SAM frame
    image = np.random.RandomState(31).randint(0, 256, size=(60, 80, 3)).astype(np.uint8)
    ret = host.preprocess_vision([[image]], type="image", conv_generator=cg, precision="fp32")
    assert len(ret) == 5 and ret[1] is None
    torch.testing.assert_close(ret[0][0].cpu()[:, :, ::7, ::7], fx["image_pos0_sub"], rtol=0, atol=1e-6)
    torch.testing.assert_close(ret[0][0].cpu().mean(dim=(2, 3)), fx["image_pos0_mean"], rtol=0, atol=1e-6)
    torch.testing.assert_close(ret[2][0].cpu()[:, :, ::16, ::16], fx["image_pos2_sub"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ret[2][0].cpu().mean(dim=(2, 3)), fx["image_pos2_mean"], rtol=1e-5, atol=1e-5)
    assert [list(x) for x in ret[3]] == fx["image_pos3"].long().tolist() and [list(x) for x in ret[4]] == fx["image_pos4"].long().tolist()
    import pytest
    with pytest.raises(AssertionError):
        host.preprocess_vision([list(frames), list(frames)])            # "Batch size must be 1"
    with pytest.raises(AssertionError):
        host.preprocess_vision([list(frames)], type="image")            # "Time dimension must be 1"


def test_conv_generator_matches_reference():
    """ConvGenerator_VideoGPTPlus.apply_for_chat, both types, with and without <im_start>/<vid_start> wrapping, vs the reference's class."""
    fx = G.fixture("host_pv.npz")
    for base in ("phi3", "llama3_1"):
        for mm in (False, True):
            cg = host.ConvGenerator_VideoGPTPlus(use_mm_start_end=mm, base_type=base, num_frames=4)
            for kind in ("video", "image"):
                ids = cg.apply_for_chat("Please segment the red car .", type=kind, tokenizer=ToyTokenizer()).cpu()
                assert ids.tolist() == fx[f"chat_ids_{base}_{int(mm)}_{kind}"].long().tolist(), (base, mm, kind)
                assert (ids == host.IMAGE_TOKEN_INDEX).sum() == (4 if kind == "video" else 1)


def test_preprocess_vision_shapes_and_write_masks(tmp_path):
    frames = [np.random.RandomState(i).randint(0, 256, size=(48, 64, 3)).astype(np.uint8) for i in range(5)]
    images, context, sam, orig, resize = host.preprocess_vision([frames], conv_generator=host.ConvGenerator_VideoGPTPlus(num_frames=4), precision="fp32")
    assert images[0].shape == (4, 3, 224, 224) and context[0].shape == (4, 3, 336, 336)
    assert sam[0].shape == (5, 3, 1024, 1024) and orig == [(48, 64)] and resize == [(768, 1024)]
    segs = {0: {0: np.zeros((48, 64), bool)}, 1: {0: np.ones((48, 64), bool)}}
    host.write_masks(segs, np.stack(frames), str(tmp_path))
    assert (tmp_path / "pred_masks_0" / "mask_1.png").exists() and (tmp_path / "masked_images" / "masked_img_0_0.jpg").exists()
    assert host.mask_iou(segs[1][0], segs[1][0]) == 1.0 and host.mask_iou(segs[0][0], segs[1][0]) == 0.0
