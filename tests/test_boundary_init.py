"""Start-up surface of the drop-in boundary: the call sequence of the reference's initialize_model_videogptplus
(R/chat.py:225-369 — from_pretrained, tokenizer.add_tokens("[SEG]"), resize_token_embeddings, config ids,
get_model().initialize_vision_modules / get_vision_tower / get_image_vision_tower + .to(), .bfloat16()/.float()/.cuda(), .eval())
executed against the façade on a checkpoint written in the released layout; afterwards inference() reproduces the reference's
ids and masks (tests/golden/e2e_tiny.npz)."""
import numpy as np
import pytest
import torch

from test_ingest import write_released

torch.set_grad_enabled(False)


class ToyTokenizer:
    """the tokenizer surface R/chat.py:286-311 touches."""

    def __init__(self, n_vocab, seg_id=None, eos=2, bos=1, unk=0, pad=None):
        self.n, self.added = n_vocab, {} if seg_id is None else {"[SEG]": seg_id}
        self.eos_token_id, self.bos_token_id, self.unk_token_id, self.pad_token_id = eos, bos, unk, pad
        self.unk_token, self.eos_token, self._pad = "<unk>", "</s>", None

    pad_token = property(lambda self: self._pad, lambda self, v: (setattr(self, "_pad", v), setattr(self, "pad_token_id", {"<unk>": self.unk_token_id, "</s>": self.eos_token_id}.get(v)))[0])

    def add_tokens(self, tok, special_tokens=False):
        if tok in self.added:
            return 0
        self.added[tok] = self.n
        self.n += 1
        return 1

    def convert_tokens_to_ids(self, tok):
        return self.added[tok]

    def __len__(self):
        return self.n


def _released(tmp_path):
    from test_oracle_e2e import e2e_setup
    fx, sd, cfg, inp = e2e_setup()
    model_dir = write_released(tmp_path, sd, cfg, "safetensors")
    return fx, cfg, inp, str(model_dir), str(tmp_path / "sam2_hiera.pt")


def _run(m, inp):
    out_ids, segs = m.inference([inp["images"]], [inp["context_images"]], [inp["images_for_sam"]], inp["input_ids"][None], [(1024, 1024)],
                                [inp["original_size"]], max_new_tokens=inp["max_new_tokens"])
    seg = segs[0]
    return out_ids[0].tolist(), np.stack([np.stack([seg[t][k] for k in sorted(seg[t])]) for t in sorted(seg)])


@pytest.mark.parametrize("grow", [0, 5])
def test_reference_init_sequence(tmp_path, cpu_ops, monkeypatch, grow):
    from videoglamm_amd import _lib
    from videoglamm_amd.chat import initialize_model_videogptplus

    monkeypatch.setattr(_lib, "load", lambda: None)
    fx, cfg, inp, model_dir, sam2 = _released(tmp_path)
    vocab = cfg["llm"]["vocab"]
    # grow = 0: the released layout ([SEG] already inside the table); grow = 5: a tokenizer that is 5 tokens longer than the table
    tok = ToyTokenizer(vocab + grow, seg_id=cfg["seg_token_idx"], eos=10 ** 6)
    model, tok2 = initialize_model_videogptplus(model_dir, precision="fp32", local_rank=0, use_sam2_video_branch=False, base_llm_type="llama3_1",
                                                tokenizer=tok, sam2_checkpoint=sam2, device="cpu")
    assert tok2 is tok and tok.pad_token_id == tok.unk_token_id
    assert model.config.seg_token_idx == cfg["seg_token_idx"] and model.config.eos_token_id == 10 ** 6 and model.config.bos_token_id == 1
    assert model.P.sd["model.embed_tokens.weight"].shape[0] == vocab + grow == model.P.sd["lm_head.weight"].shape[0]
    assert model.get_model().get_vision_tower().to(dtype=torch.float32, device="cpu") is model.get_model().get_vision_tower()
    assert model.get_model().config is model.config and model.eval() is model and model.float() is model
    ids, masks = _run(model, inp)
    assert ids == fx["framewise_output_ids"].long().tolist()
    ref = fx["framewise_masks"].numpy() > 0.5
    assert (masks & ref).sum() / (masks | ref).sum() > 0.999


def test_config_ids_are_live_and_refusals(tmp_path, cpu_ops, monkeypatch):
    from videoglamm_amd import _lib
    from videoglamm_amd.chat import initialize_model_videogptplus
    from videoglamm_amd.model import VideoGLaMMForCausalLM

    monkeypatch.setattr(_lib, "load", lambda: None)
    fx, cfg, inp, model_dir, sam2 = _released(tmp_path)
    tok = ToyTokenizer(cfg["llm"]["vocab"], seg_id=cfg["seg_token_idx"])
    model, _ = initialize_model_videogptplus(model_dir, "fp32", 0, False, False, False, "llama3_1", tokenizer=tok, sam2_checkpoint=sam2, device="cpu")
    first = fx["framewise_output_ids"].long().tolist()[inp["input_ids"].numel()]
    model.config.eos_token_id = first                    # R/chat.py:305 assigns it after construction: generation must stop on it
    ids, _ = model.inference([inp["images"]], [inp["context_images"]], [inp["images_for_sam"]], inp["input_ids"][None], [(1024, 1024)],
                             [inp["original_size"]], max_new_tokens=inp["max_new_tokens"], use_sam2_video_branch=True)
    assert ids[0].tolist() == inp["input_ids"].tolist() + [first]
    model.config.eos_token_id = None
    model.config.seg_token_idx = 10 ** 6                 # a [SEG] id the LLM never emits: the video branch returns no segments
    _, segs = model.inference([inp["images"]], [inp["context_images"]], [inp["images_for_sam"]], inp["input_ids"][None], [(1024, 1024)],
                              [inp["original_size"]], max_new_tokens=inp["max_new_tokens"], use_sam2_video_branch=True)
    assert segs == [{}]
    with pytest.raises(NotImplementedError):
        model.half()
    for kw in (dict(load_in_8bit=True), dict(load_in_4bit=True), dict(torch_dtype=torch.float16)):
        with pytest.raises(NotImplementedError):
            VideoGLaMMForCausalLM.from_pretrained(model_dir, sam2_checkpoint=sam2, seg_token_idx=cfg["seg_token_idx"], device="cpu", **kw)
    with pytest.raises(ValueError):
        initialize_model_videogptplus(model_dir, "fp32", base_llm_type="vicuna", tokenizer=tok, sam2_checkpoint=sam2, device="cpu")
    # --precision fp16 (the reference's default) is never a silent bf16 swap: a UserWarning, or a refusal under VG_FP16_STRICT=1
    with pytest.warns(UserWarning, match="runs in bfloat16"):
        m16, _ = initialize_model_videogptplus(model_dir, "fp16", base_llm_type="llama3_1", tokenizer=tok, sam2_checkpoint=sam2, device="cpu")
    assert m16.dtype == torch.bfloat16
    monkeypatch.setenv("VG_FP16_STRICT", "1")
    with pytest.raises(NotImplementedError):
        initialize_model_videogptplus(model_dir, "fp16", base_llm_type="llama3_1", tokenizer=tok, sam2_checkpoint=sam2, device="cpu")
    monkeypatch.delenv("VG_FP16_STRICT")
    bad = ToyTokenizer(cfg["llm"]["vocab"], seg_id=None)
    bad.added["[SEG]"] = 10 ** 6                          # an id outside the (resized) table must not pass silently
    with pytest.raises(ValueError):
        initialize_model_videogptplus(model_dir, "fp32", base_llm_type="llama3_1", tokenizer=bad, sam2_checkpoint=sam2, device="cpu")


def test_eos_id_sets(tmp_path):
    """HF generate()'s precedence: generation_config.json's eos ids when that file names them (config.json's are then not consulted, and
    a later assignment on model.config is ignored), config.json's otherwise (and then an assignment on model.config replaces them)"""
    import json
    from videoglamm_amd import ingest
    assert ingest.eos_ids(2, [32000, 32001, 32007], None) == [2, 32000, 32001, 32007]
    from test_oracle_e2e import e2e_setup
    fx, sd, cfg, inp = e2e_setup()
    model_dir = write_released(tmp_path, sd, cfg, "bin")
    hf = json.load(open(model_dir / "config.json"))
    hf["eos_token_id"] = 2
    json.dump(hf, open(model_dir / "config.json", "w"))
    json.dump({"eos_token_id": [7, 9]}, open(model_dir / "generation_config.json", "w"))
    got, hf2 = ingest.load_state_dict(str(model_dir), sam2_checkpoint=str(tmp_path / "sam2_hiera.pt"))
    c = ingest.derive_config(got, hf2, seg_token_idx=300)
    assert c["eos_token_id"] == [7, 9] and c["eos_from_generation_config"]
    hf2.pop("_generation_eos_token_id")
    c2 = ingest.derive_config(got, hf2, seg_token_idx=300)
    assert c2["eos_token_id"] == [2] and not c2["eos_from_generation_config"]
    from videoglamm_amd.model import VideoGLaMMForCausalLM
    m = VideoGLaMMForCausalLM.__new__(VideoGLaMMForCausalLM)
    m.cfg, m.config = c, type("C", (), {"eos_token_id": 5})()
    assert m._eos() == [7, 9]
    m.cfg = c2
    assert m._eos() == [5]
    m.config.eos_token_id = None
    assert m._eos() == [2]
