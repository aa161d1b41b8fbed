"""Drop-in façade for the reference's ``VideoGLaMMForCausalLM`` inference surface
(R/model/VideoGLaMM.py:560-596 inference, :598-768 inference_framewise, :770-879 inference_video_branch,
:325-508 model_forward(inference=True), :897-900 forward) on the MI355X kernel library.

Same argument meaning, return structure and error behaviour as the reference for this path; batch size 1 is
asserted exactly like R/model/VideoGLaMM.py:252-253.
"""
import os

import torch

from . import ops
from .params import Params
from .sam2 import SAM2
from .vlm import VisionTowers, generate, stage_mark


class _Cfg:
    """model.config: the attributes R/chat.py:296-331 reads and assigns (seg_token_idx, eos/bos/pad ids, use_sam2, ...)."""

    def __init__(self, d):
        self.__dict__.update(d)


class _Tower:
    """What get_vision_tower() / get_image_vision_tower() hand to R/chat.py:316-319,335-351: an object whose dtype / device
    movers are no-ops — the towers' weights live in the model's packed parameter store and follow the model."""

    def __init__(self, name):
        self.name, self.is_loaded = name, True

    def to(self, *a, **k):
        return self

    half = bfloat16 = float = cuda = eval = lambda self, *a, **k: self

    def __repr__(self):
        return f"<{self.name} (weights packed inside VideoGLaMMForCausalLM)>"


class _Inner:
    """model.get_model() / model.model: config + the vision-module initialisers of VideoGPTPlusMetaModel
    (R/model/videogpt_plus/model/arch.py:14-85).  The reference builds and loads the towers there; here they arrive with the
    checkpoint (ingest.load_state_dict), so the initialiser only checks that they are present."""

    def __init__(self, outer):
        self._outer = outer
        self.config = outer.config
        self.vision_tower = _Tower("InternVideo2 video tower")
        self.image_vision_tower = _Tower("CLIP image tower")

    def initialize_vision_modules(self, model_args=None, fsdp=None):
        sd = self._outer.P.sd
        for prefix in ("model.vision_tower.vision_encoder.", "model.image_vision_tower.vision_tower."):
            if not any(k.startswith(prefix) for k in sd):
                raise RuntimeError(f"no {prefix}* tensors in the checkpoint: pass vision_tower / image_vision_tower to from_pretrained")

    def get_vision_tower(self):
        return self.vision_tower

    def get_image_vision_tower(self):
        return self.image_vision_tower


class VideoGLaMMForCausalLM:
    def __init__(self, state_dict, config, torch_dtype=torch.bfloat16, device="cuda", use_sam2_video_branch=False,
                 comm=None, **kwargs):
        """state_dict: the merged checkpoint ({name: tensor}, reference naming); config: dict with keys
        seg_token_idx, iv2{depth,num_heads,patch_size}, clip{num_layers,num_heads,patch_size},
        llm{hidden,num_layers,num_heads,num_kv_heads,rms_eps,rope_theta}, sam2{image_size,trunk{...}},
        optional eos_token_id.  comm: optional videoglamm_amd.dist.FrameSharder for multi-GPU runs."""
        from . import _lib

        _lib.load()  # fail loudly if the HIP extension is missing: there is no fallback path
        self.cfg = dict(config)
        self.config = _Cfg(dict(seg_token_idx=config["seg_token_idx"], use_sam2=True, eos_token_id=config.get("eos_token_id"),
                                bos_token_id=config.get("bos_token_id"), pad_token_id=config.get("pad_token_id"),
                                mm_use_im_start_end=False, mm_use_im_patch_token=False))
        if torch_dtype == torch.float16:
            raise NotImplementedError("the MI355X kernels compute in bfloat16 or float32 (same 16-bit storage as fp16, wider exponent): "
                                      "pass torch_dtype=torch.bfloat16")
        for k in ("load_in_8bit", "load_in_4bit", "quantization_config"):
            if kwargs.get(k):
                raise NotImplementedError(f"{k}: bitsandbytes quantisation (R/chat.py:247-272) is CUDA-only; the fp8 LLM path of this build is "
                                          "cfg['llm']['decode_weights'] = cfg['llm']['prefill_gemm'] = 'fp8'")
        self.dtype = torch_dtype
        self.device = torch.device(device)
        if self.device.type == "cuda":
            # the C-ABI launches go to HIP's CURRENT device and torch's current stream of it (ops._stream): bind this process to the
            # model's device (one process per GPU) instead of launching on device 0 against device-k pointers
            self.device = torch.device("cuda", torch.cuda.current_device() if self.device.index is None else self.device.index)
            torch.cuda.set_device(self.device)
        self.use_sam2_video_branch = use_sam2_video_branch
        self._build(state_dict)
        self.model = _Inner(self)
        self.comm = comm
        # > 0: clear 4-connected blobs smaller than this from the thresholded masks ON THE DEVICE, before they cross PCIe —
        # what the reference's evaluation does on the host afterwards (remove_small_blobs(min_size=20),
        # R/eval_gcg_infer.py:20-29,182).  0 = the reference's inference() behaviour.
        self.min_blob_size = int(kwargs.get("min_blob_size", 0))
        # diagnostics only: when set to a dict, inference() leaves the fp32 mask logits ("logits", device), the [SEG] embeddings
        # ("emb") and the model's own per-step argmaxes ("argmax") in it (bench.py's self-check, tests)
        self.capture = None
        # optional callable (step, emitted id) -> replacement id | None, applied after the lm_head + argmax of every decode step (vlm.generate):
        # teacher forcing in the parity tests, synth.install_forced_tokens for synthetic-weight runs.  None = plain greedy decoding.
        self.token_hook = None
        # diagnostics only: when set to a list, the stages of inference() append (name, host time) after a device sync
        # (bench.py's per-stage replicated / sharded split; meaningful with VG_HIERA_START=serial)
        self.stages = None

    @classmethod
    def from_pretrained(cls, path, config=None, vision_tower=None, image_vision_tower=None, sam2_checkpoint=None,
                        seg_token_idx=None, lora_dir=None, iv2_origin_num_frames=None, **kwargs):
        """The released artefact layout (R/chat.py:277-319): an HF directory (*.safetensors or pytorch_model*.bin shards +
        config.json) with the LLM, the projectors, text_hidden_fcs and SAM2; the InternVideo2 .pt and the CLIP directory
        the config names under mm_vision_tower / image_mm_vision_tower (or given here); optionally a stand-alone SAM2
        checkpoint.  The architecture config is derived from config.json + tensor shapes (videoglamm_amd/ingest.py) unless
        `config` — or a "videoglamm_amd" section in config.json — gives it explicitly."""
        from . import ingest

        sd, hf = ingest.load_state_dict(path, vision_tower, image_vision_tower, sam2_checkpoint, lora_dir=lora_dir,
                                        iv2_origin_num_frames=iv2_origin_num_frames)
        if config is None:
            config = (hf or {}).get("videoglamm_amd") or ingest.derive_config(sd, hf, seg_token_idx)
        return cls(sd, config, **kwargs)

    def _build(self, sd):
        """(re)create the packed parameter store and the graphs over it for self.device / self.dtype (packing is lazy)."""
        self.P = Params(sd, self.device, self.dtype)
        self.towers = VisionTowers(self.P, self.cfg)
        self.sam2 = SAM2(self.P, "model.visual_model.", self.cfg["sam2"])

    # ------------------------------------------------------------------ start-up surface (R/chat.py:277-350)
    def eval(self):
        return self

    def get_model(self):
        return self.model

    def resize_token_embeddings(self, new_num_tokens=None):
        """HF PreTrainedModel.resize_token_embeddings as R/chat.py:300 uses it after tokenizer.add_tokens("[SEG]"): the input
        embedding and the lm_head grow (or shrink) to new_num_tokens rows.  A released checkpoint already has the [SEG] row, so
        this is the identity there.  New rows are ZERO here (HF draws them from N(0, initializer_range): values no checkpoint
        pins) — a zero lm_head row can never win the argmax, i.e. an untrained token is never emitted."""
        sd = self.P.sd
        rows = sd["model.embed_tokens.weight"].shape[0]
        if new_num_tokens is None or new_num_tokens == rows:
            return self
        for name in ("model.embed_tokens.weight", "lm_head.weight"):
            w = sd[name]
            if new_num_tokens < rows:
                sd[name] = w[:new_num_tokens].contiguous()
            else:
                sd[name] = torch.cat([w, torch.zeros(new_num_tokens - w.shape[0], w.shape[1], dtype=w.dtype, device=w.device)])
        self.cfg["llm"] = dict(self.cfg["llm"], vocab=int(new_num_tokens))
        self._build(sd)
        return self

    def _cast(self, dtype=None, device=None):
        dtype = self.dtype if dtype is None else dtype
        device = self.device if device is None else torch.device(device)
        if dtype == torch.float16:
            raise NotImplementedError("float16 is not a compute type of this build: use .bfloat16() or .float()")
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if (dtype, device) != (self.dtype, self.device):
            self.dtype, self.device = dtype, device
            if device.type == "cuda":
                torch.cuda.set_device(device)
            self._build(self.P.sd)
        return self

    def bfloat16(self):
        return self._cast(dtype=torch.bfloat16)

    def float(self):
        return self._cast(dtype=torch.float32)

    def half(self):
        return self._cast(dtype=torch.float16)

    def cuda(self, device=None):
        return self._cast(device="cuda" if device is None else (f"cuda:{device}" if isinstance(device, int) else device))

    def to(self, *args, **kwargs):
        dtype, device = kwargs.get("dtype"), kwargs.get("device")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            else:
                device = f"cuda:{a}" if isinstance(a, int) else a
        return self._cast(dtype, device)

    def _live_cfg(self):
        """the architecture config with the ids the caller may have (re)assigned on model.config (R/chat.py:303-307)."""
        c = self.cfg
        seg = getattr(self.config, "seg_token_idx", c["seg_token_idx"])
        return c if seg == c["seg_token_idx"] else dict(c, seg_token_idx=seg)

    def _eos(self):
        """ids generation stops on, with HF generate()'s precedence (transformers 4.41 GenerationMixin._prepare_generation_config): a
        generation_config.json that names eos ids wins; without one the generation config is derived from model.config, so ids the
        caller assigned on model.config.eos_token_id after load (R/chat.py:305-307) REPLACE the checkpoint's config.json ids."""
        from .ingest import eos_ids
        live = getattr(self.config, "eos_token_id", None)
        if self.cfg.get("eos_from_generation_config") or live is None:
            return eos_ids(self.cfg.get("eos_token_id")) or None
        return eos_ids(live) or None

    # ------------------------------------------------------------------ forward surface
    def forward(self, **kwargs):
        """R/model/VideoGLaMM.py:897-900: LM forward when past_key_values is passed (not part of this path),
        otherwise model_forward."""
        if "past_key_values" in kwargs:
            raise NotImplementedError("the bare LM forward is outside the accelerated path (SURVEY §8)")
        return self.model_forward(**kwargs)

    __call__ = forward

    def model_forward(self, images_for_sam, images, context_images, input_ids, labels=None, attention_masks=None,
                      offset=None, masks_list=None, label_list=None, resize_list=None, inference=False, **kwargs):
        """inference=True branch of R/model/VideoGLaMM.py:325-508: teacher-forced ids -> [SEG] rows -> framewise
        decode; returns {"pred_masks": B x T x [N,H,W] logits, "gt_masks": masks_list}."""
        if not inference:
            raise NotImplementedError("training losses are out of scope (SURVEY §2 rows 10-11)")
        assert len(images) == 1 and input_ids.shape[0] == 1  # batch size is 1 (VideoGLaMM.py:252-253)
        hw = tuple(label_list[0].shape[-2:])
        ids = input_ids[0].cpu()
        _, emb = generate(self.P, self._live_cfg(), self.towers, images[0].to(self.device), context_images[0].to(self.device), ids, 0)
        if emb.shape[0] == 0:
            return {"pred_masks": [[torch.zeros(0, *hw, device=self.device) for _ in range(len(images_for_sam[0]))]], "gt_masks": masks_list}
        logits, _ = self.sam2.framewise_branch(images_for_sam[0].to(self.device), emb, hw)
        return {"pred_masks": [[logits[t] for t in range(logits.shape[0])]], "gt_masks": masks_list}

    # ------------------------------------------------------------------ inference surface
    def inference(self, images, context_images, images_for_sam, input_ids, resize_list, original_size_list,
                  max_new_tokens=32, use_sam2_video_branch=False):
        """R/model/VideoGLaMM.py:560-596."""
        if self.device.type == "cuda" and torch.cuda.current_device() != self.device.index:
            torch.cuda.set_device(self.device)
        if use_sam2_video_branch:
            if self.config.use_sam2:
                return self.inference_video_branch(images, context_images, images_for_sam, input_ids, resize_list,
                                                   original_size_list, max_new_tokens)
            raise ValueError("use_sam2_video_branch is True, but model is not configured to use SAM2")
        return self.inference_framewise(images, context_images, images_for_sam, input_ids, resize_list,
                                        original_size_list, max_new_tokens)

    def _text_side(self, images, context_images, input_ids, max_new_tokens, after_prefill=None):
        assert len(images) == 1 and input_ids.shape[0] == 1  # batch size is 1 (VideoGLaMM.py:252-253)
        # context_images=None: single-image prompt (CLIP -> image_mm_projector without pooling, arch.py:243-245,393-397)
        ctx = context_images[0] if context_images is not None else None
        out_ids, emb = generate(self.P, self._live_cfg(), self.towers, images[0].to(self.device), None if ctx is None else ctx.to(self.device),
                                input_ids[0].cpu(), max_new_tokens, self._eos(),
                                token_hook=self.token_hook, after_prefill=after_prefill, comm=self.comm,
                                trace=self.capture, stages=self.stages)
        if self.capture is not None:
            self.capture["emb"] = emb
        return out_ids.unsqueeze(0), emb

    def _text_and_hiera(self, images, context_images, sam, input_ids, max_new_tokens, all_frames=False, static_feats=False):
        """LLM side + Hiera features of this rank's frames, Hiera on the side stream (see _hiera_async)."""
        frames = self.comm.my_frames(sam.shape[0]) if self.comm is not None else None
        # single-GPU graph-replayed propagation: Hiera writes the clip's features into the replay's own input buffers (SAM2.video_static_feats)
        bufs = self.sam2.video_static_feats(sam.shape[0]) if (static_feats and frames is None and not all_frames) else None
        box = {}

        def start():
            box["feats"], box["join"] = self._hiera_async(sam, frames, all_frames, bufs)

        # Hiera goes to the side stream FIRST: it then shares the chip with the towers / LLM prefill (big-K, MFMA-bound
        # GEMMs that leave HBM idle, where Hiera's small-K GEMMs, norms and window shuffles are bandwidth-hungry) and is
        # mostly done when the HBM-saturated decode loop starts.  Measured r01: 195 ms/clip vs 200.5 when it is enqueued
        # after the prefill (VG_HIERA_START=prefill) and 211 with no overlap at all.
        mode = os.environ.get("VG_HIERA_START", "first")
        if mode == "serial":      # no overlap (per-kernel timing runs: bench.py's instrumented step)
            stage_mark(self.stages, "begin")
            feats = self.comm.hiera_all_frames(self.sam2, sam) if all_frames else self.sam2.hiera_frames(sam, frames, bufs=bufs)
            stage_mark(self.stages, "hiera_fpn")
            out_ids, emb = self._text_side(images, context_images, input_ids, max_new_tokens)
            return out_ids, emb, feats
        if mode == "first":
            start()
            out_ids, emb = self._text_side(images, context_images, input_ids, max_new_tokens)
        else:
            out_ids, emb = self._text_side(images, context_images, input_ids, max_new_tokens, after_prefill=start)
        box["join"]()
        return out_ids, emb, box["feats"]

    def _binarize(self, logits):
        """[T,N,H,W] fp32 logits -> uint8 masks (logit > 0), small blobs removed when min_blob_size is set."""
        masks = ops.threshold(logits)
        return ops.remove_small_blobs(masks, self.min_blob_size) if self.min_blob_size > 0 else masks

    @staticmethod
    def _segments(mask_u8, frame_ids=None, obj_ids=None):
        """[T,N,H,W] uint8 (0 / 1) on host -> {frame: {obj: bool ndarray [H,W]}} (VideoGLaMM.py:757-766, 869-875); frame_ids / obj_ids:
        the global indices of the rows / columns (multi-GPU shards), default 0..T-1 / 0..N-1."""
        m = mask_u8.numpy().view(bool)       # 0 / 1 bytes ARE numpy bools: no second pass over the clip's masks
        frame_ids = range(m.shape[0]) if frame_ids is None else frame_ids
        obj_ids = range(m.shape[1]) if obj_ids is None else obj_ids
        return {t: {k: m[i, j] for j, k in enumerate(obj_ids)} for i, t in enumerate(frame_ids)}

    def _fast_masks(self):
        """thresholded masks straight from the low-res logits (vg_bilinear_mask) unless something needs the fp32 logits"""
        return self.capture is None and self.min_blob_size == 0

    def _to_host(self, masks):
        """device uint8 masks -> fresh host tensor.  The destination is a new PINNED tensor from torch's caching host allocator
        (a freed block of an earlier clip after the first call): one DMA, no staging copy — pageable D2H of 32 x 1024^2 masks
        costs ~5 ms, a reused staging buffer plus the host-side copy out of it ~3 ms on top of the 1 ms transfer."""
        if masks.device.type != "cuda":
            return masks
        out = torch.empty(masks.shape, dtype=masks.dtype, pin_memory=True)
        out.copy_(masks, non_blocking=True)
        torch.cuda.current_stream(masks.device).synchronize()
        return out

    def _hiera_async(self, sam, frames=None, all_frames=False, bufs=None):
        """Hiera + FPN of the SAM frames on a side HIP stream.  It depends only on the pixels, not on the LLM, and it is
        MFMA/LDS-bound while the LLM decode loop is an HBM-bound GEMV chain: the two overlap on the chip.  Returns
        (features per frame, join) — call join() on the consuming stream before reading the features."""
        # all_frames (multi-GPU video branch, r04): the rank's frames go through Hiera in chunks and every finished chunk is all-gathered right away
        # (FrameSharder.hiera_all_frames) — the object ranks' features arrive while the later chunks and the LLM side still run
        run = (lambda: self.comm.hiera_all_frames(self.sam2, sam)) if all_frames else (lambda: self.sam2.hiera_frames(sam, frames, bufs=bufs))
        if self.device.type != "cuda":
            return run(), (lambda: None)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            feats = run()

        def join():
            torch.cuda.current_stream(self.device).wait_stream(self._side)
            for f in feats.values():
                for t in f:
                    t.record_stream(torch.cuda.current_stream(self.device))
        return feats, join

    def inference_framewise(self, images, context_images, images_for_sam, input_ids, resize_list, original_size_list,
                            max_new_tokens=32):
        """R/model/VideoGLaMM.py:598-768 -> (output_ids [1,L+G], [ {frame: {obj: mask}} ])."""
        sam = images_for_sam[0].to(self.device)
        out_ids, emb, feats = self._text_and_hiera(images, context_images, sam, input_ids, max_new_tokens)
        if emb.shape[0] == 0:
            # the reference dereferences `.shape` of a tuple here (VideoGLaMM.py:732): same exception type
            raise AttributeError("'tuple' object has no attribute 'shape'")
        hw = tuple(original_size_list[0])
        if self.comm is not None:
            # this rank's frames under their global indices (the whole clip when the sharder gathers the masks)
            masks, fids = self.comm.framewise(self.sam2, sam, emb, hw, frame_feats=feats, binarize=None if self._fast_masks() else self._binarize)
            host = self._to_host(masks)
            stage_mark(self.stages, "mask_decode")
            return out_ids, [self._segments(host, frame_ids=fids)]
        elif self._fast_masks():
            masks = self._to_host(self.sam2.framewise_branch(sam, emb, hw, frame_feats=feats, as_masks=True)[0])
            stage_mark(self.stages, "mask_decode")
        else:
            logits, _ = self.sam2.framewise_branch(sam, emb, hw, frame_feats=feats)
            if self.capture is not None:
                self.capture["logits"] = logits
            masks = self._to_host(self._binarize(logits))
        return out_ids, [self._segments(masks)]

    def inference_video_branch(self, images, context_images, images_for_sam, input_ids, resize_list, original_size_list,
                               max_new_tokens=32):
        """R/model/VideoGLaMM.py:770-879; empty dict when no [SEG] was emitted (:840-842)."""
        sam = images_for_sam[0].to(self.device)
        # multi-GPU: frames shard for Hiera only (the propagation is a recurrence over frames) and every rank needs every frame's features:
        # one exchange of the whole clip after the last frame (default), or — FrameSharder(stream_features=True) — streamed chunk by chunk on a
        # communicator of their own while Hiera still runs
        streamed = self.comm is not None and self.comm.stream_features
        graphed = self.comm is None and self.device.type == "cuda" and os.environ.get("VG_VIDEO_GRAPH", "1") == "1"
        out_ids, emb, feats = self._text_and_hiera(images, context_images, sam, input_ids, max_new_tokens, all_frames=streamed, static_feats=graphed)
        if emb.shape[0] == 0:
            return out_ids, [{}]
        hw = tuple(original_size_list[0])
        if self.comm is not None:
            # OBJECTS shard for the propagation
            emb = self.comm.sync_seg_embeddings(emb)
            if not streamed:
                feats = self.comm.gather_frame_feats(feats, sam.shape[0], self.sam2)
            stage_mark(self.stages, "feature_all_gather")
            masks, oids = self.comm.video_branch_objects(self.sam2, sam, emb, hw, feats, binarize=None if self._fast_masks() else self._binarize)
            host = self._to_host(masks)
            stage_mark(self.stages, "propagation")
            return out_ids, [self._segments(host, obj_ids=oids)]
        if self.device.type == "cuda" and os.environ.get("VG_VIDEO_GRAPH", "1") == "1":
            # the propagation replayed from a HIP graph (same launches, same results: tests/test_host_sam2.py): the eager loop leaves
            # the GPU idle 16 % of a C2 clip waiting for Python between ~8000 small launches (r02: 378 -> 361 ms); VG_VIDEO_GRAPH=0 = eager
            fast = self._fast_masks()
            out = self.sam2.video_branch_graphed(sam, emb, hw, feats, as_masks=fast)
            if fast:
                host = self._to_host(out)
                stage_mark(self.stages, "propagation")
                return out_ids, [self._segments(host)]
            logits = out
        elif self._fast_masks():
            host = self._to_host(self.sam2.video_branch(sam, emb, hw, frame_feats=feats, as_masks=True))
            stage_mark(self.stages, "propagation")
            return out_ids, [self._segments(host)]
        else:
            logits = self.sam2.video_branch(sam, emb, hw, frame_feats=feats)
        if self.capture is not None:
            self.capture["logits"] = logits
        return out_ids, [self._segments(self._to_host(self._binarize(logits)))]
