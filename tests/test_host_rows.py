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


def test_preprocess_vision_shapes_and_write_masks(tmp_path):
    frames = [np.random.RandomState(i).randint(0, 256, size=(48, 64, 3)).astype(np.uint8) for i in range(5)]
    images, context, sam, resize, orig = host.preprocess_vision(frames, num_frames=4)
    assert images[0].shape == (4, 3, 224, 224) and context[0].shape == (4, 3, 336, 336)
    assert sam[0].shape == (5, 3, 1024, 1024) and orig == [(48, 64)] and resize == [(768, 1024)]
    segs = {0: {0: np.zeros((48, 64), bool)}, 1: {0: np.ones((48, 64), bool)}}
    host.write_masks(segs, np.stack(frames), str(tmp_path))
    assert (tmp_path / "pred_masks_0" / "mask_1.png").exists() and (tmp_path / "masked_images" / "masked_img_0_0.jpg").exists()
    assert host.mask_iou(segs[1][0], segs[1][0]) == 1.0 and host.mask_iou(segs[0][0], segs[1][0]) == 0.0
