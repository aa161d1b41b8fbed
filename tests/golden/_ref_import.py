"""Import recipe for the READ-ONLY reference tree (/root/reference/VideoGLaMM) on CPU.

Only used by tests/golden/make_golden.py in the build container: the reference is
public untrusted code that we *run* to produce golden input/output vectors, never copy.  It needs a
handful of absent third-party packages stubbed (hydra/omegaconf/timm/flash_attn/cv2/torchvision),
``Tensor.cuda`` neutralised and SAM2's hydra instantiation replaced by a tiny YAML instantiator.
Nothing here travels to the GPU box logic: tests only read the .npz fixtures this recipe produced.
"""
import importlib
import os
import sys
import types

R = "/root/reference/VideoGLaMM"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave like a package
    sys.modules[name] = m
    return m


def install():
    # transformers (and its cached *_available() probes) must be fully imported BEFORE the stubs exist
    import transformers  # noqa: F401
    import transformers.generation  # noqa: F401
    import transformers.models.clip.modeling_clip  # noqa: F401
    import transformers.models.llama.modeling_llama  # noqa: F401
    import transformers.models.phi3.modeling_phi3  # noqa: F401
    from transformers import AutoConfig, AutoModelForCausalLM, CLIPImageProcessor  # noqa: F401
    import torch

    if "hydra" not in sys.modules:
        _mod("hydra", compose=lambda *a, **k: None, initialize_config_module=lambda *a, **k: None)
        _mod("hydra.utils", instantiate=lambda *a, **k: None)
        _mod("hydra.core")
        _mod("hydra.core.global_hydra", GlobalHydra=types.SimpleNamespace(instance=lambda: types.SimpleNamespace(is_initialized=lambda: True)))
        _mod("omegaconf", OmegaConf=types.SimpleNamespace(resolve=lambda c: None))

        def to_2tuple(x):
            return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

        class DropPath(torch.nn.Identity):
            def __init__(self, *a, **k):
                super().__init__()

        _mod("timm")
        _mod("timm.models")
        _mod("timm.models.layers", DropPath=DropPath, to_2tuple=to_2tuple, trunc_normal_=torch.nn.init.trunc_normal_)
        _mod("flash_attn")
        _mod("flash_attn.flash_attn_interface", flash_attn_varlen_qkvpacked_func=None)
        _mod("flash_attn.bert_padding", unpad_input=None, pad_input=None)
        _mod("flash_attn.modules")
        _mod("flash_attn.modules.mlp", FusedMLP=None)
        _mod("flash_attn.ops")
        _mod("flash_attn.ops.rms_norm", DropoutAddRMSNorm=None)
        _mod("cv2")
        tv = _mod("torchvision")
        tvt = _mod("torchvision.transforms", Normalize=None, Resize=None, ToTensor=None, Compose=None, InterpolationMode=None)
        tvf = _mod("torchvision.transforms.functional", resize=None, to_pil_image=None, InterpolationMode=None)
        _mod("torchvision.transforms.v2")
        tvo = _mod("torchvision.ops")
        _mod("torchvision.ops.boxes", batched_nms=None, box_area=None)
        tv.transforms, tvt.functional, tv.ops = tvt, tvf, tvo

    if R not in sys.path:
        sys.path.insert(0, R)
    torch.Tensor.cuda = lambda self, *a, **k: self
    return R


def _coerce(v):
    if isinstance(v, str):
        try:
            return float(v) if any(c in v for c in ".e") else v
        except ValueError:
            return v
    return v


def instantiate(cfg, **extra):
    """Recursive `_target_` instantiation over a plain dict (replaces hydra.utils.instantiate)."""
    if isinstance(cfg, dict):
        cfg = dict(cfg)
        cfg.update(extra)
        tgt = cfg.pop("_target_", None)
        kw = {k: instantiate(v) for k, v in cfg.items()}
        if tgt is None:
            return kw
        modname, clsname = tgt.rsplit(".", 1)
        return getattr(importlib.import_module(modname), clsname)(**kw)
    if isinstance(cfg, list):
        return [instantiate(v) for v in cfg]
    return _coerce(cfg)


def sam2_model_cfg(name="sam2_hiera_l.yaml", video_predictor=True, trunk_override=None, neck_channels=None, image_size=None):
    """The dict build_sam2 / build_sam2_video_predictor would compose (R/.../sam2/build_sam.py:23-66)."""
    import yaml

    with open(os.path.join(R, "model/segment_anything_2/sam2_configs", name)) as f:
        cfg = yaml.safe_load(f)["model"]
    cfg["sam_mask_decoder_extra_args"] = dict(
        dynamic_multimask_via_stability=True,
        dynamic_multimask_stability_delta=0.05,
        dynamic_multimask_stability_thresh=0.98,
    )
    if video_predictor:
        cfg["_target_"] = "model.segment_anything_2.sam2.sam2_video_predictor.SAM2VideoPredictor"
        cfg["binarize_mask_from_pts_for_mem_enc"] = True
        cfg["fill_hole_area"] = 8
    if trunk_override:
        cfg["image_encoder"]["trunk"].update(trunk_override)
    if neck_channels:
        cfg["image_encoder"]["neck"]["backbone_channel_list"] = list(neck_channels)
    if image_size:
        cfg["image_size"] = image_size
    return cfg


def build_sam2(cfg):
    install()
    m = instantiate(cfg)
    m.eval()
    return m


def functions(relpath, names, namespace):
    """Run selected top-level function definitions of a reference script WITHOUT executing the script's module-level
    code (the eval scripts parse arguments / import absent packages at import time).  The definitions are compiled from
    the file where it lies and executed in `namespace`; nothing of them is written anywhere."""
    import ast

    path = os.path.join(R, relpath)
    with open(path) as f:
        tree = ast.parse(f.read(), path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    missing = set(names) - {n.name for n in picked}
    if missing:
        raise KeyError(f"{relpath}: no top-level function(s) {sorted(missing)}")
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), namespace)
    return [namespace[n] for n in names]
