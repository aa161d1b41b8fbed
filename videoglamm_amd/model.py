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
from .vlm import VisionTowers, generate


class _Cfg:
    def __init__(self, d):
        self.__dict__.update(d)


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
        self.config = _Cfg(dict(seg_token_idx=config["seg_token_idx"], use_sam2=True))
        self.dtype = torch_dtype
        self.device = torch.device(device)
        if self.device.type == "cuda":
            # the C-ABI launches go to HIP's CURRENT device and torch's current stream of it (ops._stream): bind this process to the
            # model's device (one process per GPU) instead of launching on device 0 against device-k pointers
            self.device = torch.device("cuda", torch.cuda.current_device() if self.device.index is None else self.device.index)
            torch.cuda.set_device(self.device)
        self.use_sam2_video_branch = use_sam2_video_branch
        self.P = Params(state_dict, self.device, torch_dtype)
        self.towers = VisionTowers(self.P, self.cfg)
        self.sam2 = SAM2(self.P, "model.visual_model.", self.cfg["sam2"])
        self.comm = comm
        # > 0: clear 4-connected blobs smaller than this from the thresholded masks ON THE DEVICE, before they cross PCIe —
        # what the reference's evaluation does on the host afterwards (remove_small_blobs(min_size=20),
        # R/eval_gcg_infer.py:20-29,182).  0 = the reference's inference() behaviour.
        self.min_blob_size = int(kwargs.get("min_blob_size", 0))
        # diagnostics only: when set to a dict, inference() leaves the fp32 mask logits ("logits", device), the [SEG] embeddings
        # ("emb") and the model's own per-step argmaxes ("argmax") in it (bench.py's self-check, tests)
        self.capture = None

    @classmethod
    def from_pretrained(cls, path, config=None, vision_tower=None, image_vision_tower=None, sam2_checkpoint=None,
                        seg_token_idx=None, **kwargs):
        """The released artefact layout (R/chat.py:277-319): an HF directory (*.safetensors or pytorch_model*.bin shards +
        config.json) with the LLM, the projectors, text_hidden_fcs and SAM2; the InternVideo2 .pt and the CLIP directory
        the config names under mm_vision_tower / image_mm_vision_tower (or given here); optionally a stand-alone SAM2
        checkpoint.  The architecture config is derived from config.json + tensor shapes (videoglamm_amd/ingest.py) unless
        `config` — or a "videoglamm_amd" section in config.json — gives it explicitly."""
        from . import ingest

        sd, hf = ingest.load_state_dict(path, vision_tower, image_vision_tower, sam2_checkpoint)
        if config is None:
            config = (hf or {}).get("videoglamm_amd") or ingest.derive_config(sd, hf, seg_token_idx)
        return cls(sd, config, **kwargs)

    def eval(self):
        return self

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
        _, emb = generate(self.P, self.cfg, self.towers, images[0].to(self.device), context_images[0].to(self.device), ids, 0)
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
        out_ids, emb = generate(self.P, self.cfg, self.towers, images[0].to(self.device), None if ctx is None else ctx.to(self.device),
                                input_ids[0].cpu(), max_new_tokens, self.cfg.get("eos_token_id"),
                                forced_tokens=self.cfg.get("forced_tokens"), after_prefill=after_prefill, comm=self.comm,
                                trace=self.capture)
        if self.capture is not None:
            self.capture["emb"] = emb
        return out_ids.unsqueeze(0), emb

    def _text_and_hiera(self, images, context_images, sam, input_ids, max_new_tokens):
        """LLM side + Hiera features of this rank's frames, Hiera on the side stream (see _hiera_async)."""
        frames = self.comm.my_frames(sam.shape[0]) if self.comm is not None else None
        box = {}

        def start():
            box["feats"], box["join"] = self._hiera_async(sam, frames)

        # Hiera goes to the side stream FIRST: it then shares the chip with the towers / LLM prefill (big-K, MFMA-bound
        # GEMMs that leave HBM idle, where Hiera's small-K GEMMs, norms and window shuffles are bandwidth-hungry) and is
        # mostly done when the HBM-saturated decode loop starts.  Measured r01: 195 ms/clip vs 200.5 when it is enqueued
        # after the prefill (VG_HIERA_START=prefill) and 211 with no overlap at all.
        mode = os.environ.get("VG_HIERA_START", "first")
        if mode == "serial":      # no overlap (per-kernel timing runs: bench.py's instrumented step)
            feats = self.sam2.hiera_frames(sam, frames)
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
    def _segments(mask_u8):
        """[T,N,H,W] uint8 on host -> {frame: {obj: bool ndarray [H,W]}} (VideoGLaMM.py:757-766, 869-875)."""
        m = mask_u8.numpy().astype(bool)
        return {t: {k: m[t, k] for k in range(m.shape[1])} for t in range(m.shape[0])}

    def _hiera_async(self, sam, frames=None):
        """Hiera + FPN of the SAM frames on a side HIP stream.  It depends only on the pixels, not on the LLM, and it is
        MFMA/LDS-bound while the LLM decode loop is an HBM-bound GEMV chain: the two overlap on the chip.  Returns
        (features per frame, join) — call join() on the consuming stream before reading the features."""
        if self.device.type != "cuda":
            return self.sam2.hiera_frames(sam, frames), (lambda: None)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            feats = self.sam2.hiera_frames(sam, frames)

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
            masks = self.comm.framewise(self.sam2, sam, emb, hw, frame_feats=feats, binarize=self._binarize)
        else:
            logits, _ = self.sam2.framewise_branch(sam, emb, hw, frame_feats=feats)
            if self.capture is not None:
                self.capture["logits"] = logits
            masks = self._binarize(logits).cpu()
        return out_ids, [self._segments(masks)]

    def inference_video_branch(self, images, context_images, images_for_sam, input_ids, resize_list, original_size_list,
                               max_new_tokens=32):
        """R/model/VideoGLaMM.py:770-879; empty dict when no [SEG] was emitted (:840-842)."""
        sam = images_for_sam[0].to(self.device)
        out_ids, emb, feats = self._text_and_hiera(images, context_images, sam, input_ids, max_new_tokens)
        if emb.shape[0] == 0:
            return out_ids, [{}]
        hw = tuple(original_size_list[0])
        if self.comm is not None:
            emb = self.comm.sync_seg_embeddings(emb)
            feats = self.comm.gather_frame_feats(feats, sam.shape[0])
        if self.device.type == "cuda" and os.environ.get("VG_VIDEO_GRAPH", "0") == "1":
            # the propagation replayed from a HIP graph: same results, measured neutral (r01: 201.05 vs 200.51 ms per clip), off by default
            logits = self.sam2.video_branch_graphed(sam, emb, hw, feats)
        else:
            logits = self.sam2.video_branch(sam, emb, hw, frame_feats=feats)
        if self.capture is not None:
            self.capture["logits"] = logits
        return out_ids, [self._segments(self._binarize(logits).cpu())]
