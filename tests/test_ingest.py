"""Checkpoint ingest: the tiny end-to-end model written in the RELEASED layout — HF directory with
.bin shards and an HF-style config.json, InternVideo2 .pt with foreign tensors beside vision_encoder.*, CLIP directory
with a text tower, stand-alone SAM2 checkpoint — loads into the same state dict / config and reproduces the reference's
ids and masks (tests/golden/e2e_tiny.npz)."""
import json
import os

import numpy as np
import pytest
import torch

torch.set_grad_enabled(False)


def write_released(tmp, sd, cfg, fmt):
    from safetensors.torch import save_file

    model_dir, clip_dir = tmp / "VideoGLaMM-tiny", tmp / "clip-vit"
    model_dir.mkdir()
    clip_dir.mkdir()
    iv2 = {k[len("model.vision_tower."):]: v for k, v in sd.items() if k.startswith("model.vision_tower.")}
    clip = {"vision_model." + k[len("model.image_vision_tower.vision_tower."):]: v for k, v in sd.items() if k.startswith("model.image_vision_tower.")}
    sam2 = {k[len("model.visual_model."):]: v for k, v in sd.items() if k.startswith("model.visual_model.")}
    rest = {k: v.contiguous() for k, v in sd.items() if not k.startswith(("model.vision_tower.", "model.image_vision_tower.", "model.visual_model."))}
    iv2["text_encoder.embeddings.weight"] = torch.zeros(3, 3)              # the stage-2 checkpoint also holds the text side
    clip["text_model.embeddings.token_embedding.weight"] = torch.zeros(3, 3)
    torch.save({"module": iv2}, tmp / "InternVideo2-stage2_1b-224p-f4.pt")
    torch.save({"model": sam2}, tmp / "sam2_hiera.pt")
    names = sorted(rest)
    if fmt == "bin":
        torch.save(clip, clip_dir / "pytorch_model.bin")
        torch.save({k: rest[k] for k in names[::2]}, model_dir / "pytorch_model-00001-of-00002.bin")
        torch.save({k: rest[k] for k in names[1::2]}, model_dir / "pytorch_model-00002-of-00002.bin")
    else:
        save_file({k: v.contiguous() for k, v in clip.items()}, str(clip_dir / "model.safetensors"))
        save_file({k: rest[k] for k in names[::2]}, str(model_dir / "model-00001-of-00002.safetensors"))
        save_file({k: rest[k] for k in names[1::2]}, str(model_dir / "model-00002-of-00002.safetensors"))
    c = cfg["llm"]
    hf = dict(architectures=["VideoGLaMMForCausalLM"], hidden_size=c["hidden"], num_hidden_layers=c["num_layers"], num_attention_heads=c["num_heads"],
              num_key_value_heads=c["num_kv_heads"], rms_norm_eps=c["rms_eps"], rope_theta=c["rope_theta"], vocab_size=c["vocab"],
              intermediate_size=c["ffn"], mm_vision_tower=str(tmp / "InternVideo2-stage2_1b-224p-f4.pt"), image_mm_vision_tower=str(clip_dir),
              sam2=cfg["sam2"])             # a non-preset trunk: its layout rides in config.json
    with open(model_dir / "config.json", "w") as fh:
        json.dump(hf, fh)
    return model_dir


@pytest.mark.parametrize("fmt", ["bin", "safetensors"])
def test_released_layout_round_trip(tmp_path, fmt, cpu_ops, monkeypatch):
    from test_oracle_e2e import e2e_setup
    from videoglamm_amd import _lib, ingest
    from videoglamm_amd.model import VideoGLaMMForCausalLM

    monkeypatch.setattr(_lib, "load", lambda: None)
    fx, sd, cfg, inp = e2e_setup()
    model_dir = write_released(tmp_path, sd, cfg, fmt)
    got, hf = ingest.load_state_dict(str(model_dir), sam2_checkpoint=str(tmp_path / "sam2_hiera.pt"))
    # foreign tensors dropped, every name mapped back (CLIP under the transformers-4.41 module path ...vision_tower.vision_model.*,
    # which the fixture — dumped with transformers 5.x — spells without the vision_model level; vlm.py accepts both)
    norm = {k.replace("vision_tower.vision_model.", "vision_tower."): v for k, v in got.items()}
    assert set(norm) == set(sd), set(norm) ^ set(sd)
    assert all(torch.equal(norm[k], sd[k]) for k in sd)
    derived = ingest.derive_config(got, hf, seg_token_idx=cfg["seg_token_idx"])
    for part in ("llm", "sam2"):
        for k, v in cfg[part].items():
            assert derived[part][k] == v, (part, k, derived[part][k], v)
    for part in ("iv2", "clip"):
        for k, v in cfg[part].items():
            assert derived[part][k] == v, (part, k, derived[part][k], v)
    m = VideoGLaMMForCausalLM.from_pretrained(str(model_dir), sam2_checkpoint=str(tmp_path / "sam2_hiera.pt"), seg_token_idx=cfg["seg_token_idx"],
                                              torch_dtype=torch.float32, device="cpu")
    out_ids, segs = m.inference([inp["images"]], [inp["context_images"]], [inp["images_for_sam"]], inp["input_ids"][None], [(1024, 1024)],
                                [inp["original_size"]], max_new_tokens=inp["max_new_tokens"])
    assert out_ids[0].tolist() == fx["framewise_output_ids"].long().tolist()
    seg = segs[0]
    got_m = np.stack([np.stack([seg[t][k] for k in sorted(seg[t])]) for t in sorted(seg)])
    ref = fx["framewise_masks"].numpy() > 0.5
    assert (got_m & ref).sum() / (got_m | ref).sum() > 0.999


def test_missing_pieces_fail_loudly(tmp_path, monkeypatch):
    from test_oracle_e2e import e2e_setup
    from videoglamm_amd import ingest

    fx, sd, cfg, inp = e2e_setup()
    model_dir = write_released(tmp_path, sd, cfg, "bin")
    with pytest.raises(FileNotFoundError):                              # SAM2 neither in the directory nor given
        ingest.load_state_dict(str(model_dir))
    os.remove(tmp_path / "InternVideo2-stage2_1b-224p-f4.pt")
    with pytest.raises(FileNotFoundError):
        ingest.load_state_dict(str(model_dir), sam2_checkpoint=str(tmp_path / "sam2_hiera.pt"))
    with pytest.raises(FileNotFoundError):
        ingest.load_state_dict(str(tmp_path / "nowhere"))
    got = {k: v for k, v in sd.items()}
    with pytest.raises(KeyError):                                       # head count is not recoverable from shapes
        ingest.derive_config(got, {})


def test_iv2_pos_embed_interpolation_matches_reference():
    """interpolate_iv2_pos_embed vs the reference's interpolate_pos_embed_internvideo2_new (tests/golden/ingest.npz)"""
    import _golden as G
    from videoglamm_amd import ingest
    fx = G.fixture("ingest.npz")
    torch.testing.assert_close(ingest.interpolate_iv2_pos_embed(fx["pos_in"], 8, 4, 6), fx["pos_t_and_s"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ingest.interpolate_iv2_pos_embed(fx["pos_in"], 8, 4, 4), fx["pos_t_only"], rtol=1e-6, atol=1e-6)
    assert torch.equal(ingest.interpolate_iv2_pos_embed(fx["pos_in"], 8, 8, 4), fx["pos_in"])          # the released config: identity


def _write_lora(tmp, sd, cfg, r=4, alpha=8.0):
    """a LoRA training output over the released layout: adapters on two LLM projections + non_lora_trainables.bin (text_hidden_fcs and
    an lm_head row block retrained), with the key prefixes the reference strips (train_ds_with_videogptplus.py:168-170)."""
    d = tmp / "lora_out"
    d.mkdir()
    g = torch.Generator().manual_seed(3)
    targets = ["model.layers.0.self_attn.q_proj", "model.layers.1.mlp.down_proj"]
    ad, merged = {}, {}
    for name in targets:
        w = sd[name + ".weight"]
        a, b = torch.randn(r, w.shape[1], generator=g) * 0.1, torch.randn(w.shape[0], r, generator=g) * 0.1
        ad[f"base_model.model.{name}.lora_A.weight"], ad[f"base_model.model.{name}.lora_B.weight"] = a, b
        merged[name + ".weight"] = w + (alpha / r) * (b @ a)
    torch.save(ad, d / "adapter_model.bin")
    with open(d / "adapter_config.json", "w") as fh:
        json.dump(dict(r=r, lora_alpha=alpha, target_modules=["q_proj", "down_proj"], peft_type="LORA"), fh)
    fc = "model.text_hidden_fcs.0.2.weight"
    merged[fc] = sd[fc] * 1.5 + 0.01
    torch.save({"base_model.model." + fc: merged[fc], "base_model.model.model.not_in_the_model.weight": torch.zeros(2, 2)}, d / "non_lora_trainables.bin")
    return d, merged


def test_lora_and_non_lora_merge(tmp_path):
    from test_oracle_e2e import e2e_setup
    from videoglamm_amd import ingest
    fx, sd, cfg, inp = e2e_setup()
    model_dir = write_released(tmp_path, sd, cfg, "bin")
    lora_dir, merged = _write_lora(tmp_path, sd, cfg)
    got, _ = ingest.load_state_dict(str(model_dir), sam2_checkpoint=str(tmp_path / "sam2_hiera.pt"), lora_dir=str(lora_dir))
    base, _ = ingest.load_state_dict(str(model_dir), sam2_checkpoint=str(tmp_path / "sam2_hiera.pt"))
    for k, v in got.items():
        if k in merged:
            torch.testing.assert_close(v, merged[k], rtol=1e-6, atol=1e-6)
            assert not torch.equal(v, base[k])
        else:
            assert torch.equal(v, base[k]), k
    assert "model.not_in_the_model.weight" not in got                    # strict=False: unknown names are ignored
    os.remove(lora_dir / "non_lora_trainables.bin")
    with pytest.raises(FileNotFoundError):
        ingest.load_state_dict(str(model_dir), sam2_checkpoint=str(tmp_path / "sam2_hiera.pt"), lora_dir=str(lora_dir))


@pytest.mark.gpu
def test_released_layout_on_the_gpu(tmp_path, cuda):
    """the released layout -> from_pretrained -> inference on the HIP kernels (fp32 parity mode): ids exact, masks IoU > 0.999"""
    from test_oracle_e2e import e2e_setup
    from videoglamm_amd.model import VideoGLaMMForCausalLM
    fx, sd, cfg, inp = e2e_setup()
    model_dir = write_released(tmp_path, sd, cfg, "safetensors")
    m = VideoGLaMMForCausalLM.from_pretrained(str(model_dir), sam2_checkpoint=str(tmp_path / "sam2_hiera.pt"), seg_token_idx=cfg["seg_token_idx"],
                                              torch_dtype=torch.float32, device=cuda)
    for branch, key in ((False, "framewise"), (True, "video")):
        out_ids, segs = m.inference([inp["images"]], [inp["context_images"]], [inp["images_for_sam"]], inp["input_ids"][None], [(1024, 1024)],
                                    [inp["original_size"]], max_new_tokens=inp["max_new_tokens"], use_sam2_video_branch=branch)
        assert out_ids[0].tolist() == fx[f"{key}_output_ids"].long().tolist()
        seg = segs[0]
        got_m = np.stack([np.stack([seg[t][k] for k in sorted(seg[t])]) for t in sorted(seg)])
        ref = fx[f"{key}_masks"].numpy() > 0.5
        assert (got_m & ref).sum() / (got_m | ref).sum() > 0.999
