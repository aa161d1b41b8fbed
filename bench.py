#!/usr/bin/env python
"""bench.py — frames/sec end-to-end (text + masks) of the VideoGLaMM hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--branch framewise|video] [--no-cpu-baseline]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
              --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full pass of VideoGLaMMForCausalLM.inference() over one synthetic clip, from
device-resident preprocessed tensors + input_ids to (token ids on host, thresholded masks on host):
dual vision encoders -> V-L adapters -> Llama-3-8B prefill + 32 greedy decode steps with one [SEG]
-> L-V adapter -> SAM2-L (Hiera + FPN + mask decoder) over every frame.
Workload at N=1 = BASELINE config C2, the configuration BASELINE.json's metric is quoted on (32-frame 1024^2 clip
-> 32 x 1024^2 SAM frames, masks returned at 1024^2, Te=16 encoder frames -> 3361-row prompt, Llama-3-8B bf16,
InternVideo2-1B, CLIP-L/336, SAM2-L, one [SEG] object).  N>1 (default --scaling strong) = BASELINE config C3: the SAME 32-frame clip,
frames sharded N-way for Hiera + the mask decode, the vision towers sharded by frame / chunk and the LLM prefill by rows
(--replicate-llm switches both off: then every rank emits the single-GPU ids by construction), the decode loop replicated, RCCL
all-gather of the [SEG] embedding; the masks are all-gathered on the devices and rank 0 — the rank a caller reads — returns the
whole clip (the reference's contract).  --scaling weak: 32 SAM frames PER RANK (clip of 32N frames), LLM side replicated, every
rank keeps the masks of its own frames (no data-path collective).
Weights are random-init of the exact architectures (no network / no public Llama VideoGLaMM checkpoint).
The line is self-checking ("quality"): after the timed region the same clip is re-run ONCE in fp32 parity mode on the
GPU (same bf16-rounded weights, teacher-forced to the bf16 run's ids) and the bf16 masks / argmaxes are compared with it.
"""
import zlib
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--branch", default="framewise", choices=["framewise", "video"],
                    help="framewise = the reference's default path (chat.py without --use_sam2_video_branch)")
    ap.add_argument("--frames", "--frames-per-gpu", dest="frames", type=int, default=32,
                    help="SAM frames of the clip (strong scaling: of the whole clip; --scaling weak: per rank)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = BASELINE config C3 (the same clip sharded N-way, default); weak = --frames per rank")
    ap.add_argument("--replicate-llm", action="store_true",
                    help="N > 1, strong scaling: keep the vision towers and the LLM prefill replicated on every rank (ids == single-GPU ids by construction)")
    ap.add_argument("--te", type=int, default=16, help="encoder frames (NUM_FRAMES; the reference's default 16)")
    ap.add_argument("--src", type=int, default=1024, help="source (output mask) resolution")
    ap.add_argument("--max-new-tokens", type=int, default=32)
    ap.add_argument("--objects", type=int, default=1, help="[SEG] objects (multi-object GCG: 8)")
    ap.add_argument("--prefill", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: the LLM prefill GEMMs on the fp8 MFMA path (per-token / per-channel scales); with --decode-weights fp8 = "
                         "BASELINE config C4's fp8 LLM path (NOT the bf16 headline configuration)")
    ap.add_argument("--decode-weights", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: e4m3 weights + row scales in the decode step's MLP GEMVs and the lm_head (decode side of BASELINE config C4's fp8 "
                         "LLM path; NOT the bf16 headline configuration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-quality", action="store_true", help="skip the fp32 parity-mode re-run that fills the \"quality\" object")
    ap.add_argument("--no-video-record", action="store_true", help="skip the video-branch sub-run that fills the \"video_branch\" object of a framewise line")
    ap.add_argument("--video-steps", type=int, default=5, help="timed steps of the video-branch sub-run")
    ap.add_argument("--no-config-records", action="store_true", help="skip the C1 / C4-clip sub-records of the default (C2) line")
    ap.add_argument("--llm", default="llama3-8b", choices=["llama3-8b", "phi3-mini"],
                    help="llama3-8b = BASELINE configs C1-C3 (default); phi3-mini = the released checkpoint's LLM")
    ap.add_argument("--tiny", action="store_true", help="plumbing check on a toy architecture (NOT a valid bench)")
    ap.add_argument("--plumbing", action="store_true",
                    help="allow --gpus N with fewer than N devices (ranks share GPUs; VG_DIST_BACKEND=gloo): a plumbing check, NOT a scaling record — the line says so")
    return ap.parse_args()


def make_inputs(cfg, args, world, device):
    g = torch.Generator().manual_seed(1234)
    te, T = args.te, args.frames * (world if args.scaling == "weak" else 1)
    S = cfg["sam2"]["image_size"]
    iv, cl = cfg["iv2"]["img_size"], cfg["clip"]["img_size"]
    images = torch.randn(te, 3, iv, iv, generator=g).to(device)
    context = torch.randn(te, 3, cl, cl, generator=g).to(device)
    sam = torch.randn(T, 3, S, S, generator=g).to(device)
    ids = torch.cat([torch.tensor([1, 5, 6]), torch.full((te,), -200), torch.randint(3, cfg["llm"]["vocab"] - 2, (30,), generator=g)])[None]
    return images, context, sam, ids


class GemmMeter:
    """Per-launch HIP-event timing of the dominant kernel (gemm_tile_glds_kernel, bf16: every M > 16 ops.linear) on
    torch's current stream — the stream every vg_* kernel is launched on (videoglamm_amd/ops.py:_stream; the Hiera
    launches are metered on the side stream they run on)."""

    def __init__(self, ops):
        self.ops, self.orig, self.rec = ops, ops.linear, []
        self.skinny = {}           # VG_BENCH_SKINNY=1: the M <= 16 shapes seen (diagnostic, printed to stderr by main)
        self.shapes = []           # (M, N, K) per record (VG_BENCH_GEMM_SHAPES=1: per-shape table on stderr)
        self.scope = None          # "mask_decoder" while SAM2.mask_decoder runs (the north star's mask-decoder GEMMs)

    @staticmethod
    def kernel_of(M, N, K, glu, windowed):
        """the launcher's own routing (vg_gemm_route, videoglamm_amd/csrc/vg_gemm.hip): which tile kernel runs this shape"""
        from videoglamm_amd import _lib
        return {1: "glds", 2: "k64b", 3: "w128", 4: "s128", 5: "small64", 6: "p8n", 7: "rr"}[_lib.load().vg_gemm_route(int(M), int(N), int(K), 1, 1 if glu else 0, 1 if windowed else 0)]

    def __enter__(self):
        def timed(x, w, *a, **k):
            M = x.numel() // x.shape[-1]
            if M <= 16:   # skinny path
                if os.environ.get("VG_BENCH_SKINNY"):
                    self.skinny[(M, w.shape[0], w.shape[1], str(k.get("out_dtype")))] = self.skinny.get((M, w.shape[0], w.shape[1], str(k.get("out_dtype"))), 0) + 1
                return self.orig(x, w, *a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = self.orig(x, w, *a, **k)
            e1.record()
            N, K = w.shape[0] // (2 if k.get("glu") else 1), w.shape[1]
            es = x.element_size()
            nbytes = (M * K + w.shape[0] * K) * es + M * N * y.element_size() * (2 if k.get("residual") is not None else 1)
            self.rec.append((2.0 * M * w.shape[0] * K, e0, e1, nbytes, self.kernel_of(M, N, K, k.get("glu"), False), self.scope))
            self.shapes.append((M, w.shape[0], K))
            return y
        def timed_window(x, w, bias, B, H, W, ws, scatter, **k):      # Hiera's window-folded projections: same kernel
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = self.orig_window(x, w, bias, B, H, W, ws, scatter, **k)
            e1.record()
            M = B * (-(-H // ws)) * (-(-W // ws)) * ws * ws
            N, K = w.shape
            nbytes = (x.numel() + w.numel()) * x.element_size() + y.numel() * y.element_size() * (2 if k.get("residual") is not None else 1)
            # (r04: power-of-two windows that tile the image ride on the 256x256 phase-split kernel — the launcher routes them as if they were not
            # windowed (vg_gemm.hip: launch_gemm's `wroute`); every other window shape stays on the 128x128 kernels)
            p2 = lambda v: v > 0 and (v & (v - 1)) == 0      # noqa: E731
            on_p8 = (p2(ws) and H % ws == 0 and W % ws == 0 and p2(H // ws) and p2(W // ws) and x.dtype == torch.bfloat16
                     and os.environ.get("VG_GEMM_P8", "1") != "0")
            kern = self.kernel_of(M, N, K, False, not on_p8)
            if x.dtype == torch.bfloat16 and self.kernel_of(M, N, K, False, True) == "rr":      # (the row-register route takes any window shape)
                kern = "rr"
            self.rec.append((2.0 * M * N * K, e0, e1, nbytes, kern, self.scope))
            self.shapes.append((M, N, K))
            return y
        def timed_ln(x, ln, w, bias=None, act=0, window=None):            # LayerNorm -> projection (ops.linear_ln): one row-register launch where it is built;
            N, K = w.shape                                                  # its two-launch form goes through the hooks above
            M = x.numel() // x.shape[-1]
            if window is not None:
                B, H, W, ws = window
                M = B * (-(-H // ws)) * (-(-W // ws)) * ws * ws
            if not (x.dtype == torch.bfloat16 and self.kernel_of(M, N, K, False, window is not None) == "rr"):
                return self.orig_ln(x, ln, w, bias, act, window)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = self.orig_ln(x, ln, w, bias, act, window)
            e1.record()
            self.rec.append((2.0 * M * N * K, e0, e1, (x.numel() + w.numel()) * x.element_size() + y.numel() * y.element_size(), "rr", self.scope))
            self.shapes.append((M, N, K))
            return y
        from videoglamm_amd import sam2 as _sam2
        meter = self
        self.orig_md = _sam2.SAM2.mask_decoder

        def scoped_md(obj, *a, **k):
            meter.scope = "mask_decoder"
            try:
                return meter.orig_md(obj, *a, **k)
            finally:
                meter.scope = None
        _sam2.SAM2.mask_decoder = scoped_md
        self.ops.linear = timed
        self.orig_window = self.ops.linear_window
        self.ops.linear_window = timed_window
        self.orig_ln = self.ops.linear_ln
        self.ops.linear_ln = timed_ln
        return self

    def __exit__(self, *exc):
        from videoglamm_amd import sam2 as _sam2
        _sam2.SAM2.mask_decoder = self.orig_md
        self.ops.linear = self.orig
        self.ops.linear_window = self.orig_window
        self.ops.linear_ln = self.orig_ln

    def summary(self, kernel=None, scope=None):
        torch.cuda.synchronize()
        rec = [r for r in self.rec if (kernel is None or r[4] == kernel) and (scope is None or r[5] == scope)]
        flops = sum(r[0] for r in rec)
        ms = sum(r[1].elapsed_time(r[2]) for r in rec)
        return flops, ms, len(rec), sum(r[3] for r in rec)


class AttnMeter:
    """Per-launch HIP events around ops.attention / ops.attention_windows (attn_kernel<bf16, DP, 64, 4>, one record class per
    head-dim instantiation DP).  Algorithmic flops = 4 * B * H * Sq * Skv * D (QK^T and PV), halved under the causal mask,
    Skv = the window length for Hiera's packed windows."""

    def __init__(self, ops):
        self.ops, self.rec = ops, []

    @staticmethod
    def dp(D):
        return 32 if D <= 32 else 64 if D <= 64 else 96 if D <= 96 else 128 if D <= 128 else 256

    def __enter__(self):
        self.o_attn, self.o_win = self.ops.attention, self.ops.attention_windows

        def timed(q, k, v, scale, causal=False, window=0):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = self.o_attn(q, k, v, scale, causal, window)
            e1.record()
            B, Sq, H, D = q.shape
            Skv = k.shape[1]
            fl = 4.0 * B * H * Sq * Skv * D
            if causal:        # visible (query, key) pairs / all pairs; a window caps the keys per query
                vis = sum(min(Skv - Sq + i + 1, window or Skv) for i in range(Sq)) if window else Sq * (Skv - Sq) + Sq * (Sq + 1) / 2.0
                fl *= vis / (Sq * Skv)
            self.rec.append((fl, e0, e1, self.dp(D), (B, H, Sq, Skv, D)))
            return y

        def timed_win(q, k, v, scale):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n0 = len(self.rec)
            e0.record()
            y = self.o_win(q, k, v, scale)
            e1.record()
            del self.rec[n0:]            # (the unpacked fallback goes through ops.attention: count it once, here)
            Bw, wtok, H, D = q.shape
            own = q.dtype == torch.bfloat16 and D == 72 and wtok in (16, 64, 256)        # vg_window_attention's kernels
            self.rec.append((4.0 * Bw * H * wtok * wtok * D, e0, e1, "window" if own else self.dp(D), (Bw, H, wtok, wtok, D)))
            return y

        def timed_pooled(q, k, v, scale):       # q-pooled windows (4 x 16, 16 x 64): None = not taken by the window kernels
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = self.o_pool(q, k, v, scale)
            e1.record()
            if y is not None:
                Bw, wq, H, D = q.shape
                self.rec.append((4.0 * Bw * H * wq * k.shape[1] * D, e0, e1, "window", (Bw, H, wq, k.shape[1], D)))
            return y
        self.o_pool = self.ops.window_attention
        self._in_win = False
        self.ops.attention, self.ops.attention_windows = timed, timed_win
        self.o_dv = self.ops.attention_dv

        def timed_dv(q, k, v, scale):      # SAM2 memory cross-attention with the v-projection behind it: QK^T on D dims, PV on the memory's own DV dims
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = self.o_dv(q, k, v, scale)
            e1.record()
            B, Sq, H, D = q.shape
            self.rec.append((2.0 * B * H * Sq * k.shape[1] * (D + v.shape[-1]), e0, e1, f"{self.dp(D)}_dv{v.shape[-1]}", (B, H, Sq, k.shape[1], D, v.shape[-1])))
            return y
        self.ops.attention_dv = timed_dv

        def pooled_or_inner(q, k, v, scale):     # attention_windows calls window_attention itself: meter only the direct (q-pooled) calls
            return self.o_pool(q, k, v, scale) if q.shape[1] == k.shape[1] else timed_pooled(q, k, v, scale)
        self.ops.window_attention = pooled_or_inner
        return self

    def __exit__(self, *exc):
        self.ops.attention, self.ops.attention_windows = self.o_attn, self.o_win
        self.ops.window_attention = self.o_pool
        self.ops.attention_dv = self.o_dv

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for fl, e0, e1, dp, _ in self.rec:
            a = out.setdefault(dp, [0.0, 0.0, 0])
            a[0] += fl
            a[1] += e0.elapsed_time(e1)
            a[2] += 1
        return out


class DecodeMeter:
    """HIP events around every graph-replayed decode step (LlamaDecoder.decode_step): the HBM-bound half of the
    workload.  Bytes per step = every LLM weight matrix once (lm_head included) — the algorithmic traffic."""

    def __init__(self):
        from videoglamm_amd import vlm
        self.cls, self.orig, self.rec = vlm.LlamaDecoder, vlm.LlamaDecoder.decode_step, []

    def __enter__(self):
        meter = self

        def timed(dec):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            meter.orig(dec)
            e1.record()
            meter.rec.append((e0, e1))
        self.cls.decode_step = timed
        return self

    def __exit__(self, *exc):
        self.cls.decode_step = self.orig

    def summary(self):
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.rec), len(self.rec)



def meter_decode_gemv(model, ops, reps=3):
    """Per-launch HIP events around the decode step's gate|up GEMV (decode_gemv_fast_kernel<GLU>: the kernel with the most
    GPU time in a C1 step).  Inside the timed region it runs in a HIP-graph replay, where single launches cannot be
    bracketed, so `reps` extra decode steps are run eagerly on the same decoder state (KV cache, token, position rewound
    afterwards).  Returns (bytes per launch, [us per launch])."""
    dec = getattr(model.P, "_decoder", None)
    if dec is None or not getattr(dec, "fused_decode", False):
        return None
    orig, rec = ops.decode_gemv, []

    def timed(x, w, *a, **k):
        if not k.get("glu"):
            return orig(x, w, *a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = orig(x, w, *a, **k)
        e1.record()
        rec.append((e0, e1, w.numel() * w.element_size()))
        return y
    snap_tok, snap_pos = dec.tok_dev.clone(), dec.pos_dev.clone()
    ops.decode_gemv = timed
    try:
        for _ in range(reps):
            dec.pos_dev.copy_(snap_pos - 1)      # stay inside the cache: re-run the last position
            dec._decode_step()
    finally:
        ops.decode_gemv = orig
        dec.tok_dev.copy_(snap_tok)
        dec.pos_dev.copy_(snap_pos)
    torch.cuda.synchronize()
    if not rec:
        return None
    return rec[0][2], [1e3 * a.elapsed_time(b) for a, b, _ in rec]

PMC_TAG = "r06_c2"      # profiles/<PMC_TAG>_pmc_<kernel>.json: HBM traffic per launch of the default workload


def quality(cfg, args, model, step, device):
    """Self-check of the timed configuration's OUTPUTS (outside the timed region).  The bf16 model's clip is re-run once in
    fp32 parity mode on the GPU — the same bf16-rounded weights upcast to fp32, the exact-fp32 MFMA path that the -m gpu
    parity tests pin to the reference within 1e-3 — teacher-forced to the bf16 run's token ids so that both runs see the
    same sequence.  mask_miou_vs_fp32 follows R/eval_gcg_metrics.py:26-35 (sum of intersections / sum of unions over the
    frames of an object, mean over objects); ids_top1_agree = fraction of decode steps where the bf16 argmax equals the
    fp32 argmax on the same prefix (random-init logits are near-flat over 128k tokens: near-ties flip)."""
    from videoglamm_amd import synth
    from videoglamm_amd.model import VideoGLaMMForCausalLM

    cap = model.capture = {}
    out_ids, _ = step()
    model.capture = None
    lb = cap["logits"]                                   # [T,N,H,W] fp32 on the device
    ids = out_ids[0].tolist()
    n_prompt = len(ids) - len(cap["argmax"])
    q = {"mode": "bf16 run vs the same clip in fp32 parity mode on the GPU (same bf16-rounded weights, teacher-forced to the bf16 ids)",
         "finite": bool(torch.isfinite(lb).all()) and bool(torch.isfinite(cap["emb"].float()).all()),
         "mask_fraction": round(float((lb > 0).float().mean()), 4),
         "ids_crc32": zlib.crc32(str(ids).encode()) & 0xffffffff,
         "seg_objects": int(cap["emb"].shape[0])}
    mb = lb > 0
    del lb
    sd = synth.device_state_dict(synth.manifest(cfg), device, torch.bfloat16)
    sd = {k: v.float() for k, v in sd.items()}
    cfg32 = dict(cfg, forced_tokens={i: t for i, t in enumerate(ids[n_prompt:])})
    cfg32["llm"] = {k: v for k, v in cfg["llm"].items() if k not in ("decode_weights", "prefill_gemm")}
    m32 = synth.install_forced_tokens(VideoGLaMMForCausalLM(sd, cfg32, torch_dtype=torch.float32, device=device))
    del sd
    images, context, sam, pids = make_inputs(cfg, args, 1, device)
    cap32 = m32.capture = {}
    t0 = time.time()
    m32.inference([images], [context], [sam], pids, [(1024, 1024)], [(args.src, args.src)], max_new_tokens=args.max_new_tokens,
                  use_sam2_video_branch=args.branch == "video")
    torch.cuda.synchronize()
    q["fp32_run_s"] = round(time.time() - t0, 1)
    l32 = cap32["logits"]
    m32b = l32 > 0
    inter = (mb & m32b).sum(dim=(0, 2, 3)).double()
    union = (mb | m32b).sum(dim=(0, 2, 3)).double()
    iou = torch.where(union > 0, inter / union.clamp_min(1), torch.ones_like(union))
    q["mask_miou_vs_fp32"] = round(float(iou.mean()), 5)
    bi, bu = (~mb & ~m32b).sum(dim=(0, 2, 3)).double(), (~mb | ~m32b).sum(dim=(0, 2, 3)).double()
    q["background_miou_vs_fp32"] = round(float(torch.where(bu > 0, bi / bu.clamp_min(1), torch.ones_like(bu)).mean()), 5)      # the complement's IoU: the sensitive one when most pixels are on
    per_frame = (mb & m32b).sum(dim=(2, 3)).double() / (mb | m32b).sum(dim=(2, 3)).double().clamp_min(1)
    q["min_frame_iou_vs_fp32"] = round(float(per_frame.min()), 5)
    q["mask_fraction_fp32"] = round(float(m32b.float().mean()), 4)
    forced = cfg.get("forced_tokens") or {}
    free = [i for i in range(len(cap["argmax"])) if i not in forced]
    agree = sum(1 for i in free if cap["argmax"][i] == cap32["argmax"][i])
    q["ids_top1_agree"] = round(agree / max(len(free), 1), 4)
    q["seg_emb_cosine"] = round(float(torch.nn.functional.cosine_similarity(cap["emb"].float(), cap32["emb"].float(), dim=-1).min()), 5)
    del m32
    torch.cuda.empty_cache()
    return q


def cpu_baseline(cfg, args):
    """Reference algorithm (oracle/, CPU fp32 restatement pinned to the reference) on the host cores, bounded
    sample of the same workload: ONE frame through every per-frame stage at full architecture size
    (Hiera-L+FPN+mask decoder, CLIP-L/336, 1/4 of an InternVideo2-1B 4-frame chunk) and ONE decoder layer of the LLM on
    the whole prompt (all 32 layers in fp32 would need 32 GB of host RAM and minutes).  The clip time is assembled from
    the samples under the oracle's own schedule — vision once, the LM re-forwarded over the whole sequence for every
    generated token (the reference runs generate(use_cache=False), R/model/VideoGLaMM.py:610-626)."""
    from oracle import sam2 as osam, seeded, vlm as ovlm
    from videoglamm_amd import synth

    torch.set_grad_enabled(False)
    cores = torch.get_num_threads()
    t_total = 0.0
    man = synth.sam2_manifest(cfg["sam2"])
    sd = seeded.seeded_state_dict(man, 0, seeded.sam2_overrides())
    S = cfg["sam2"]["image_size"]
    img = torch.randn(1, 3, S, S, generator=torch.Generator().manual_seed(1))
    text = torch.randn(1, 256, generator=torch.Generator().manual_seed(2)) * 0.5
    t0 = time.time()
    osam.framewise_branch(sd, "", cfg["sam2"], img, text, (args.src, args.src))
    t_sam = time.time() - t0
    t_total += t_sam
    del sd
    vman = synth.vlm_manifest(cfg)
    pc = "model.image_vision_tower.vision_tower."
    sdc = seeded.seeded_state_dict({k: v for k, v in vman.items() if k.startswith(pc)}, 0)
    t0 = time.time()
    ovlm.clip_forward(sdc, pc, dict(num_heads=cfg["clip"]["num_heads"], num_layers=cfg["clip"]["num_layers"], patch_size=cfg["clip"]["patch_size"]),
                      torch.randn(1, 3, cfg["clip"]["img_size"], cfg["clip"]["img_size"]))
    t_clip = time.time() - t0
    t_total += t_clip
    del sdc
    pi = "model.vision_tower.vision_encoder."
    sdi = seeded.seeded_state_dict({k: v for k, v in vman.items() if k.startswith(pi)}, 0)
    t0 = time.time()
    ovlm.iv2_forward(sdi, pi, dict(depth=cfg["iv2"]["depth"], num_heads=cfg["iv2"]["num_heads"], patch_size=cfg["iv2"]["patch_size"]),
                     torch.randn(1, 4, 3, cfg["iv2"]["img_size"], cfg["iv2"]["img_size"]))
    t_iv2 = (time.time() - t0) / 4.0
    t_total += t_iv2
    del sdi
    # one LLM decoder layer on the S prompt rows
    c = cfg["llm"]
    pl = "model."
    sdl = seeded.seeded_state_dict({k: v for k, v in vman.items() if k.startswith("model.layers.0.") or k == "model.norm.weight"}, 0)
    S_llm = 208 * args.te + 33
    x = torch.randn(S_llm, c["hidden"], generator=torch.Generator().manual_seed(3)) * 0.1
    t0 = time.time()
    ovlm.llama_forward(sdl, pl, dict(c, num_layers=1), x)
    t_layer = time.time() - t0
    del sdl
    T, G, L = args.frames, args.max_new_tokens, c["num_layers"]
    t_vision = T * t_sam + args.te * t_clip + args.te * t_iv2
    t_llm = (G + 1) * L * t_layer                     # G + 1 full re-forwards of the sequence, L layers each
    return dict(value=round(T / (t_vision + t_llm), 5), unit="frames/sec", cores=cores, kind="port", method="extrapolated from timed per-module samples (not a timed clip)",
                sample=f"fp32 oracle at full architecture size: 1 frame through Hiera-L+FPN+mask decoder ({t_sam:.1f}s), 1 frame through CLIP-L/336 "
                       f"({t_clip:.1f}s), InternVideo2-1B chunk/4 ({t_iv2:.1f}s), 1 of {L} LLM layers on the {S_llm}-row prompt ({t_layer:.2f}s); clip time "
                       f"= {T} x SAM + {args.te} x (CLIP + IV2) + {G + 1} re-forwards x {L} layers (the oracle restates generate(use_cache=False)) "
                       f"= {t_vision:.0f}s vision + {t_llm:.0f}s LLM; vision alone: {T / t_vision:.4f} frames/sec")


def attn_roofs(am, traffic_of=lambda key: None, peak=2500.0):
    """one roofline object per attention kernel class of an AttnMeter pass (QK^T and PV on the MFMA: the north star's "attention-GEMM roofline")."""
    roofs = {}
    for dp, (fl, ms, n) in sorted(am.summary().items(), key=lambda kv: str(kv[0])):
        if ms <= 0:
            continue
        ach = fl / (ms * 1e-3) / 1e12
        if dp == "window":
            roofs["attn_window"] = {"bound": "hbm", "kernel": "win256_attn_kernel<72> / tiny_win_attn_kernel<72, wq, wk> (Hiera's windows: one workgroup per 256-token window and head, "
                                                                "one wave per 16- / 64-token window and head; bounded by the fused q|k|v projection's bytes)",
                                    "achieved_tflops": round(ach, 1), "launches": n, "algorithmic_tflop_per_step": round(fl / 1e12, 3),
                                    "avg_launch_us": round(1e3 * ms / n, 1), "kernel_ms_per_step": round(ms, 2), "traffic": traffic_of("attn_window")}
            continue
        if "_dv" in str(dp):
            label = (f"attn_dma_d256v64_kernel (r06: LDS-DMA ring, 256-dim keys, DV = {str(dp).split('_dv')[1]}; attn_kernel<bf16, 256, 64, 8, 2, 64> below 256 query rows) — "
                     "vg_attention_dv: SAM2 memory cross-attention, v-projection behind the attention; flops = 2 Sq Skv (D + DV); + the split-KV merge")
        elif str(dp).startswith("256"):
            label = "attn_kernel<bf16, 256, 64, 8, 2> (head dim 256: two key-split waves per 32 query rows; flash-style, QK^T / PV on the 32x32x16 MFMA; + the split-KV merge where used)"
        else:
            label = (f"attn_dma_kernel<{dp}> (long sequences: LDS-DMA staging, r06) / attn_kernel<bf16, {dp}, 64, 4 | 8> (flash-style, QK^T / PV on the 32x32x16 MFMA; "
                     "+ the split-KV merge where used)")
        roofs[f"attn_d{dp}"] = {"bound": "mfma", "kernel": label,
                                "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic_of(f"attn_d{dp}"),
                                "launches": n, "algorithmic_tflop_per_step": round(fl / 1e12, 3), "algorithmic_tflop_per_launch": round(fl / 1e12 / n, 5),
                                "avg_launch_us": round(1e3 * ms / n, 1), "kernel_ms_per_step": round(ms, 2)}
    return roofs


def video_record(args, model, ops, inputs):
    """The SAM2 video branch (memory attention + memory encoder + the predictor's recurrence: S4 / S5 / S9 / S10 — the half of the path the
    framewise default never launches) on the SAME clip, as a sub-record of the framewise line: timed steps bracketed like the headline
    (inputs resident in HBM -> ids + masks on the host), then one instrumented eager pass for the attention rooflines and the stage split,
    and the launch count of the propagation read off its captured HIP graph."""
    images, context, sam, ids = inputs

    def vstep():
        return model.inference([images], [context], [sam], ids, [(1024, 1024)], [(args.src, args.src)],
                               max_new_tokens=args.max_new_tokens, use_sam2_video_branch=True)
    vstep()                     # eager pass + graph capture of the propagation
    vstep()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.video_steps):
        out = vstep()
    torch.cuda.synchronize()
    dt = time.time() - t0
    T = sam.shape[0]
    rec = {"what": "the same clip through use_sam2_video_branch=True (R/model/VideoGLaMM.py:770-879): timed exactly like the headline, after it",
           "steps": args.video_steps, "warmup": 2, "ms_per_step": round(1e3 * dt / args.video_steps, 2), "value": round(T * args.video_steps / dt, 3),
           "unit": "frames/sec", "objects": len(next(iter(out[1][0].values()))) if out[1][0] else 0}
    nodes = model.sam2.video_graph_nodes()
    if nodes:
        k, (kern, cpy, other) = next(iter(nodes.items()))
        rec["propagation_graph"] = {"frames": k[0], "objects": k[1], "kernel_nodes": kern, "memcpy_nodes": cpy, "other_nodes": other,
                                    "kernel_launches_per_tracked_frame": round(kern / max(k[0] - 1, 1), 1)}
    knobs = {"VG_HIERA_START": "serial", "VG_TOWERS_OVERLAP": "0", "VG_VIDEO_GRAPH": "0"}
    prev = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        vstep()
        with AttnMeter(ops) as am:
            vstep()
        model.stages = []
        vstep()
        marks, model.stages = model.stages, None
    finally:
        for k, v in prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for k, v in attn_roofs(am).items():
        if k in ("attn_d256", "attn_d256_dv64"):
            rec["roofline_" + ("attn_dv" if "_dv" in k else "attn_d256_self")] = v
    rec["stages_eager_serial_ms"] = {b[0]: round(1e3 * (b[1] - a[1]), 2) for a, b in zip(marks, marks[1:]) if b[0] not in ("start", "begin")}
    return rec


def config_record(args, cfg, model, device, frames, src, objects, te, fp8=False, steps=3):
    """Another BASELINE configuration on the SAME loaded model, timed like the headline (inputs resident in HBM -> ids + masks on the host; barrier-free:
    one GPU), as a sub-record of the default line — so that every single-GPU configuration has a driver-timed number: C1 (8 x 512^2, one [SEG]) and C4's
    clip on one GPU (64 x 1024^2, 8 [SEG]; bf16, and with the fp8 LLM path BASELINE config C4 names: fp8 MFMA prefill GEMMs + fp8 decode weights)."""
    import argparse
    from videoglamm_amd import synth
    seg = cfg["seg_token_idx"]
    forced = {8: seg} if objects == 1 else {4 + 3 * i: seg for i in range(objects)}
    saved_llm, saved_dec, saved_hook = model.cfg["llm"], getattr(model.P, "_decoder", None), model.token_hook
    if fp8:
        model.cfg["llm"] = dict(saved_llm, decode_weights="fp8", prefill_gemm="fp8")
        model.P._decoder = None
    try:
        synth.install_forced_tokens(model, forced)
        images, context, sam, ids = make_inputs(cfg, argparse.Namespace(te=te, frames=frames, scaling="strong"), 1, device)

        def cstep():
            return model.inference([images], [context], [sam], ids, [(1024, 1024)], [(src, src)], max_new_tokens=args.max_new_tokens, use_sam2_video_branch=False)
        cstep()
        cstep()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(steps):
            out = cstep()
        torch.cuda.synchronize()
        dt = time.time() - t0
        n_obj = len(next(iter(out[1][0].values()))) if out[1][0] else 0
        rec = {"frames": frames, "source": src, "encoder_frames": te, "objects": n_obj, "steps": steps, "warmup": 2, "ms_per_step": round(1e3 * dt / steps, 2),
               "value": round(frames * steps / dt, 3), "unit": "frames/sec", "generated_tokens": int(out[0].shape[1] - ids.shape[1]),
               "dtype": "bf16" if not fp8 else "bf16 model; LLM prefill GEMMs fp8, decode-step MLP / lm_head weights fp8 (e4m3, row scales)"}
        del images, context, sam, out
    finally:
        model.cfg["llm"], model.token_hook = saved_llm, saved_hook
        if fp8:
            model.P._decoder = saved_dec
        torch.cuda.empty_cache()
    return rec


def mask_decoder_record(model, device, frames=64, objects=8, reps=3):
    """The SAM2 mask decoder (S7 / S8: two-way transformer, upscaling, hypernetwork product, mask selection) at BASELINE config C4's size on ONE
    GPU — 64 frames x 8 [SEG] objects = 512 (frame, object) instances — as a sub-record of the C2 line, where the stage is 32 instances and too
    small to say anything.  HBM-bound by SURVEY.md section 8(d): per instance the reference streams the [4096, 256] keys through 2 blocks x 3
    passes, the two high-resolution feature maps once, and writes 4 x 256^2 logits = 2 (3.1 + 4.2) + 0.26 M elements = 29.7 MB in bf16.
    Timed with HIP events around SAM2.framewise_branch on precomputed Hiera features (mask upsampling to 1024^2 + threshold included, no D2H)."""
    sam2 = model.sam2
    g = torch.Generator().manual_seed(4321)
    sam = torch.randn(frames, 3, sam2.S, sam2.S, generator=g).to(device)
    emb = (torch.randn(objects, 256, generator=g) * 0.5).to(device=device, dtype=sam2.dtype)
    feats = sam2.hiera_frames(sam)
    del sam
    hw = (1024, 1024)
    sam2.framewise_branch(None, emb, hw, frame_feats=feats, frames=list(range(frames)), as_masks=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        masks, _ = sam2.framewise_branch(None, emb, hw, frame_feats=feats, frames=list(range(frames)), as_masks=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    inst = frames * objects
    per_inst = (2 * (3.1e6 + 4.2e6) + 0.26e6) * 2.0
    gbs = inst * per_inst / (ms * 1e-3) / 1e9
    frac = float(masks.sum(dtype=torch.int64)) / masks.numel()
    del feats, masks
    torch.cuda.empty_cache()
    return {"what": f"SAM2 mask decoder on {frames} frames x {objects} objects = {inst} instances (BASELINE config C4's clip on one GPU), framewise branch, "
                    "precomputed Hiera features -> thresholded 1024^2 masks on the device",
            "bound": "hbm", "instances": inst, "ms": round(ms, 2), "us_per_instance": round(1e3 * ms / inst, 2),
            "algorithmic_bytes_per_instance": round(per_inst), "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
            "mask_fraction": round(frac, 4)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if world > torch.cuda.device_count() and not args.plumbing:
        raise SystemExit(f"--gpus {world} on a node with {torch.cuda.device_count()} device(s): ranks would share GPUs and the line would not be a scaling "
                         "record; pass --plumbing (with VG_DIST_BACKEND=gloo) for a plumbing check")
    local %= torch.cuda.device_count()     # (--plumbing: several ranks on one GPU)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    comm = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("VG_DIST_BACKEND", "nccl")      # "nccl" == RCCL over xGMI on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        from videoglamm_amd.dist import FrameSharder
        # strong scaling (C3): rank 0 returns the whole clip like the reference's inference(); weak scaling: explicit opt-out of the mask exchange
        comm = FrameSharder(gather_masks="rank0" if args.scaling == "strong" else False)
        # who is in the job: one entry per rank (device index, UUID, name) — the record is only a scaling record when the UUIDs are distinct
        props = torch.cuda.get_device_properties(local)
        mine = {"rank": rank, "device": local, "uuid": str(getattr(props, "uuid", "")), "name": props.name}
        members = [None] * world
        dist.all_gather_object(members, mine)
        if args.scaling == "strong" and not args.replicate_llm:
            os.environ.setdefault("VG_TOWERS_SHARDED", "1")
            os.environ.setdefault("VG_PREFILL_SHARDED", "1")

    from videoglamm_amd import ops, synth
    from videoglamm_amd.model import VideoGLaMMForCausalLM

    torch.set_grad_enabled(False)
    cfg = synth.videoglamm_llama3_8b() if args.llm == "llama3-8b" else synth.videoglamm_phi3_mini()
    if args.tiny:
        cfg = dict(seg_token_idx=319, projector_depth=2,
                   iv2=dict(img_size=224, patch_size=14, embed_dim=128, depth=3, num_heads=4, mlp_hidden=256),
                   clip=dict(img_size=336, patch_size=14, hidden=128, mlp=256, num_layers=3, num_heads=4),
                   llm=dict(vocab=320, hidden=128, ffn=256, num_layers=2, num_heads=4, num_kv_heads=2, rms_eps=1e-5, rope_theta=10000.0),
                   sam2=dict(image_size=1024, trunk=dict(embed_dim=16, num_heads=1, stages=[1, 2, 3, 1], global_att_blocks=[4, 5],
                                                         window_spec=[8, 4, 8, 4], window_pos_embed_bkg_spatial_size=[7, 7])))
    # exactly --objects [SEG] tokens (C1: one, at decode step 8; multi-object GCG: every third step from 4)
    assert 1 <= args.objects and 4 + 3 * (args.objects - 1) < args.max_new_tokens
    if args.decode_weights == "fp8":
        cfg["llm"] = dict(cfg["llm"], decode_weights="fp8")
    if args.prefill == "fp8":
        cfg["llm"] = dict(cfg["llm"], prefill_gemm="fp8")
    cfg["forced_tokens"] = {8: cfg["seg_token_idx"]} if args.objects == 1 else {4 + 3 * i: cfg["seg_token_idx"] for i in range(args.objects)}
    t0 = time.time()
    sd = synth.device_state_dict(synth.manifest(cfg), device, torch.bfloat16)
    model = synth.install_forced_tokens(VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.bfloat16, device=device, comm=comm))
    images, context, sam, ids = make_inputs(cfg, args, world, device)
    T = sam.shape[0]
    use_video = args.branch == "video"

    def step():
        return model.inference([images], [context], [sam], ids, [(1024, 1024)], [(args.src, args.src)],
                               max_new_tokens=args.max_new_tokens, use_sam2_video_branch=use_video)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 0)):
        out = step()
    del sd
    t_load = time.time() - t0
    barrier()
    t1 = time.time()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.time() - t1
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt[0])
    out_ids, segs = out
    n_obj = len(next(iter(segs[0].values()))) if segs[0] else 0      # (a rank's dict holds ITS frames under their global indices)
    shape = (args.frames, args.src, args.objects, args.te)
    strong = args.scaling == "strong"
    name = "C1" if shape == (8, 512, 1, 8) else ("C2" if world == 1 or not strong else f"C3 (C2's clip sharded {world}-way)") if shape == (32, 1024, 1, 16) \
        else "C4 share of one GPU (64 frames / 8)" if (args.frames, args.src, args.objects, world) == (8, 1024, 8, 1) \
        else "C4 clip (64 frames, 8 [SEG])" if (args.frames, args.src, args.objects) == (64, 1024, 8) and strong else "custom"
    llm_sharded = world > 1 and os.environ.get("VG_TOWERS_SHARDED", "0") == "1" and os.environ.get("VG_PREFILL_SHARDED", "0") == "1"
    if world == 1:
        par = "1 GPU"
    elif strong:
        par = (f"one clip over {world} GPUs: SAM frames sharded (Hiera + FPN, mask decode"
               + ("; video branch: objects sharded for the propagation, FPN features all-gathered" if use_video else "") + "), "
               + ("vision towers sharded by frame / chunk + sequence-parallel LLM prefill (a rank's row chunks take other GEMM tile routes: "
                  "bf16 ids can differ from the single-GPU run on near-ties; --replicate-llm restores id equality by construction)" if llm_sharded
                  else "vision towers + LLM prefill replicated (ids == single-GPU ids by construction)")
               + ", decode loop replicated, [SEG] embedding all-gathered, masks all-gathered on the devices and returned whole by rank 0")
    else:
        par = f"weak scaling: {args.frames} frames per rank sharded x{world} (masks stay with the rank that made them: gather_masks=False), LLM side replicated"
    if args.llm != "llama3-8b":
        name += " with the Phi-3-mini LLM"
    if args.decode_weights == "fp8" or args.prefill == "fp8":
        name += f" [fp8 LLM path: prefill {args.prefill}, decode weights {args.decode_weights} — not the bf16 headline configuration]"
    res = {
        "metric": "frames/sec end-to-end (text+masks)", "value": round(T * args.steps / dt, 3), "unit": "frames/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * dt / args.steps, 2),
        "higher_is_better": True, "scaling": args.scaling if world > 1 else "strong", "vs_baseline": None,
        "dtype": "bf16" if (args.decode_weights, args.prefill) == ("bf16", "bf16") else
                 f"bf16 model; LLM prefill GEMMs {args.prefill}, decode-step MLP / lm_head weights {args.decode_weights} (e4m3, row scales)", "data": "synthetic",
        "config": {"workload": f"{name}: {T}-frame {args.src}^2-source clip ({T} x 1024^2 SAM frames over {world} GPU(s)), "
                               f"Te={args.te}, {'Llama-3-8B' if args.llm == 'llama3-8b' else 'Phi-3-mini'} bf16 + InternVideo2-1B + CLIP-L/336 + SAM2-L, {n_obj} [SEG] object(s), "
                               f"{args.max_new_tokens} greedy tokens, {args.branch} SAM2 branch" + (" [TINY plumbing config]" if args.tiny else ""),
                   "frames": T, "encoder_frames": args.te, "generated_tokens": int(out_ids.shape[1] - ids.shape[1]),
                   "seq_len": 208 * args.te + ids.shape[1] - args.te, "parallelism": par,
                   "masks_returned": "whole clip" if world == 1 else ("whole clip on rank 0 (gather_masks='rank0')" if strong else "each rank its own frames (gather_masks=False)"),
                   "weights": "random-init (synthetic)"},
        "load_s": round(t_load, 1),
    }
    if world > 1:
        # self-verifying multi-GPU record: the backend, who took part, what the collectives moved and how long the issuing stream spent in them
        comm.profile = True
        comm.collective_report()
        step()                      # one extra pass in the timed (overlapped) configuration, outside the timed region
        rep = comm.collective_report()
        comm.profile = False
        names = sorted(rep)
        tt = torch.tensor([rep[k]["ms"] for k in names], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        for k, v in zip(names, tt.tolist()):
            rep[k]["ms_max_over_ranks"] = round(v, 3)
        uuids = sorted({m["uuid"] or f"device{m['device']}" for m in members})
        res["distributed"] = {"backend": torch.distributed.get_backend() + (" (RCCL)" if torch.distributed.get_backend() == "nccl" else ""),
                              "ranks": members, "distinct_devices": len(uuids), "device_uuids": uuids,
                              "valid_scaling_record": len(uuids) == world and torch.distributed.get_backend() == "nccl",
                              "plumbing": bool(args.plumbing),
                              "collectives": {"note": "one extra pass, rank 0's counts; ms = device time between events bracketing the collective on the stream that issues "
                                                      "it (includes waiting for the slowest rank to arrive); async streamed feature gathers (FrameSharder(stream_features=True) / VG_FEATURES_STREAMED=1, off by default) run from issue to the consuming stream's wait",
                                              **rep}}
        if len(uuids) != world:
            res["distributed"]["warning"] = f"{world} ranks on {len(uuids)} device(s): ranks time-slice GPUs — NOT a scaling measurement"
    if not args.no_roofline:
        # (every rank runs these extra steps — a step contains the frame-sharding collectives — rank 0 reports)
        # instrumented extra step (not part of the timed region): per-launch HIP events on the launch stream.  It runs
        # with the Hiera/LLM stream overlap switched off: an event pair on one of two concurrently fed streams also
        # brackets the time the launch waits behind the other stream's kernels, which is not kernel time
        # (the rocprofv3 summary under profiles/ is taken on the timed, overlapped configuration)
        knobs = {"VG_HIERA_START": "serial", "VG_TOWERS_OVERLAP": "0", "VG_VIDEO_GRAPH": "0"}     # (a graph replay has no per-launch host calls to bracket)
        prev = {k: os.environ.get(k) for k in knobs}
        os.environ.update(knobs)
        try:
            step()      # untimed: in this stream configuration the caching allocator first has to grow the main stream's pool,
            #             and a hipMalloc between an event pair's records would be billed to the launch it brackets
            with GemmMeter(ops) as gm, DecodeMeter() as dm, AttnMeter(ops) as am:
                step()
            # per-stage split of one more serial pass (device sync + host clock between the stages): which stages shard with N and
            # which are replicated on every rank — the Amdahl term of the strong-scaling configuration
            model.stages = []
            step()
            marks, model.stages = model.stages, None
        finally:
            for k, v in prev.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        dur = {b[0]: 1e3 * (b[1] - a[1]) for a, b in zip(marks, marks[1:]) if b[0] not in ("start", "begin")}
        if world > 1:      # a stage takes as long as its slowest rank
            tt = torch.tensor([dur[k] for k in sorted(dur)], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dur = dict(zip(sorted(dur), tt.tolist()))
        shards = {"hiera_fpn": world > 1, "towers": llm_sharded, "prefill": llm_sharded, "decode": False, "mask_decode": world > 1,
                  "feature_all_gather": False, "propagation": world > 1 and args.objects > 1}
        res["stages"] = {"note": "one extra pass with the streams serialised and a device sync between the stages (outside the timed region; max over ranks): "
                                 "ms per stage on a rank and whether the stage's work divides by the number of GPUs (sharded) or is repeated on every rank",
                         **{k: {"ms": round(v, 2), "sharded": bool(shards.get(k, False))} for k, v in dur.items()},
                         "replicated_ms": round(sum(v for k, v in dur.items() if not shards.get(k, False)), 2),
                         "sharded_ms": round(sum(v for k, v in dur.items() if shards.get(k, False)), 2)}
        if world > 1:
            # the Amdahl bound of THIS run's split next to what was measured: a sharded stage measured on a rank already holds 1/world of the clip's work
            rep_ms, sh_ms = res["stages"]["replicated_ms"], res["stages"]["sharded_ms"]
            one_gpu = rep_ms + sh_ms * world
            res["stages"]["amdahl"] = {"serial_one_gpu_ms_estimate": round(one_gpu, 2), "serial_this_run_ms": round(rep_ms + sh_ms, 2),
                                       "speedup_bound_at_this_n": round(one_gpu / (rep_ms + sh_ms), 3), "speedup_bound_at_infinity": round(one_gpu / max(rep_ms, 1e-9), 3),
                                       "measured_ms_per_step": res["ms_per_step"],
                                       "note": "serialised-stream stage sums (no Hiera/LLM overlap): the bound is the ratio of the sums, the measured step overlaps streams"}
        dec_ms, dec_n = dm.summary()
        if os.environ.get("VG_BENCH_GEMM_SHAPES"):
            torch.cuda.synchronize()
            tab = {}
            for r, sh in zip(gm.rec, gm.shapes):
                e = tab.setdefault((r[4],) + sh, [0, 0.0])
                e[0] += 1
                e[1] += r[1].elapsed_time(r[2])
            for k, (n, ms) in sorted(tab.items(), key=lambda kv: -kv[1][1])[:40]:
                print(f"gemm {k[0]:5s} M={k[1]:8d} N={k[2]:6d} K={k[3]:6d}  x{n:5d}  {ms:8.2f} ms  {ms / n * 1e3:8.1f} us each  {2.0 * k[1] * k[2] * k[3] * n / ms / 1e9:8.1f} TF/s", file=sys.stderr)
        if gm.skinny:
            print("skinny GEMM shapes (M, N, K, out_dtype): count", sorted(gm.skinny.items()), file=sys.stderr)
        peak = 2500.0
        # HBM traffic per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, committed summaries made by
        # tools/collect_profiles.sh + tools/pmc_json.py): only quoted for the workload they were measured on (C2 framewise, 1 GPU)
        pmc_ok = (world == 1 and not args.tiny and args.branch == "framewise" and (args.frames, args.te, args.src, args.objects) == (32, 16, 1024, 1)
                  and args.llm == "llama3-8b" and (args.decode_weights, args.prefill) == ("bf16", "bf16"))

        def traffic_of(key):
            pmc = os.path.join(ROOT, "profiles", f"{PMC_TAG}_pmc_{key}.json")
            if pmc_ok and os.path.exists(pmc):
                with open(pmc) as fh:
                    return round(json.load(fh)["traffic_bytes_per_launch"])
            return None

        def trace_of(key, mode):
            """per-launch average of this kernel class in the committed rocprofv3 kernel trace of the same command (tools/collect_profiles.sh +
            tools/kt_json.py): mode "overlapped" = the timed configuration (Hiera on its side stream), "serial" = the configuration of the
            instrumented pass the live `frac` is measured in."""
            path = os.path.join(ROOT, "profiles", f"{PMC_TAG}_kt_{mode}.json")
            if pmc_ok and key and os.path.exists(path):
                with open(path) as fh:
                    doc = json.load(fh)
                # the committed trace only speaks for the library it was taken on (tools/kt_json.py stamps its sha256): after any kernel change the
                # trace-derived fields are dropped until the profiles are re-collected — `frac` (live events of THIS run) is the headline fraction
                if doc.get("lib_sha16") != LIB_SHA16:
                    stale_traces.add(os.path.basename(path))
                    return None
                return doc["kernels"].get(key)
            return None

        import hashlib
        so = os.path.join(ROOT, "videoglamm_amd", "csrc", "libvgkernels.so")
        LIB_SHA16 = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16]
        stale_traces = set()

        def roof(kernel, label, scope=None, key=None):
            flops, ms, n, nbytes = gm.summary(kernel, scope)
            ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            r = {"bound": "mfma", "kernel": label, "achieved": round(ach, 1), "peak": peak,
                 "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic_of(key) if key else None,
                 "launches": n, "algorithmic_tflop_per_step": round(flops / 1e12, 2),
                 "algorithmic_tflop_per_launch": round(flops / 1e12 / max(n, 1), 4),
                 "algorithmic_bytes_per_launch": round(nbytes / max(n, 1)), "avg_launch_us": round(1e3 * ms / max(n, 1), 1),
                 "kernel_ms_per_step": round(ms, 2), "hbm_frac": round(nbytes / (ms * 1e-3) / 8e12, 4) if ms > 0 else 0.0,
                 "frac_serial": round(ach / peak, 4)}       # = frac: HIP events per launch in the serial-stream instrumented pass
            for mode in ("serial", "overlapped"):
                t = trace_of(key, mode)
                if t and n:
                    # the same algorithmic flops per launch over the rocprofv3 average of the committed trace of that stream configuration — only when
                    # the trace's class holds the launches the live meter counted (ops.linear): a class that also takes window / split-K / bmm launches
                    # (r03: 136 glds launches in the trace against 104 metered) would divide one population's flops by another's time
                    r[f"avg_launch_us_trace_{mode}"] = t["avg_us"]
                    r[f"launches_trace_{mode}"] = t["launches_per_pass"]
                    if abs(t["launches_per_pass"] - n) > 0.02 * n:
                        r[f"frac_trace_{mode}"] = None
                        continue
                    r[f"frac_trace_{mode}"] = round(flops / n / (t["avg_us"] * 1e-6) / 1e12 / peak, 4)
                    if t.get("mfma_busy_frac") is not None:
                        r[f"mfma_busy_frac_{mode}"] = t["mfma_busy_frac"]
            if r.get("frac_trace_overlapped") is not None:
                r["frac_overlapped"] = r["frac_trace_overlapped"]
            return r
        # one object per tile kernel; "roofline" is the one with the most GPU time in the step
        labels = {"glds": "gemm_tile_glds_kernel<bf16> (128x128 tile, 128-byte K steps)",
                  "k64b": "gemm_tile_k64b_kernel<bf16> (128x128 tile, 64-byte K steps: K*2 < 1024 B)",
                  "w128": "gemm_tile_p8_kernel<bf16> (256x256 tile, 8 waves of 128x64, phase-split K steps; r04: replaces gemm_tile_w128x8_kernel on the 256x256 route)",
                  "p8n": "gemm_tile_p8n_kernel<bf16> (256x192 tile, the phase-split pipeline with the operand roles swapped; r05: shapes whose rounds x tile cost favour the narrow tile)",
                  "s128": "gemm_tile_s128_kernel<bf16> (128x128 tile, one 128-byte-row stage: 1024 <= K*2 <= 3072 B)",
                  "rr": "gemm_rr_kernel<K / 16, resident | streamed W> (r05: K = 144 / 288 over >= 65536 rows — Hiera stages 1-2, FPN laterals: the wave's A rows in registers, W chunks in LDS, no tiles; HBM-bound at K = 144: see hbm_frac)",
                  "small64": "gemm_small64_kernel<bf16> (64x64 tile, whole K <= 256 in one DMA burst: problems of < 256 128x128 tiles — memory attention, mask decoder)"}
        roofs = {k: roof(k, labels[k], key="gemm_" + k) for k in labels}
        roofs = {"gemm_" + k: v for k, v in roofs.items() if v["launches"]}
        gv = None if args.tiny else meter_decode_gemv(model, ops)
        if gv is not None:
            nbytes, us = gv
            avg = sum(us) / len(us)
            per_step = avg * 1e-3 * len(us) / 3 * dec_n          # launches per decode step x decode steps per clip
            roofs["decode_gemv_glu"] = {"bound": "hbm", "kernel": "decode_gemv_fast_kernel<bf16, GLU> (decode step: RMSNorm + gate|up GEMV + SwiGLU of one row)",
                                        "achieved": round(nbytes / avg / 1e3, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / avg / 1e3 / 8000.0, 4),
                                        "traffic": traffic_of("decode_gemv_glu"), "launches": round(len(us) / 3 * dec_n), "algorithmic_bytes_per_launch": nbytes,
                                        "avg_launch_us": round(avg, 1), "kernel_ms_per_step": round(per_step, 2),
                                        "note": "timed on eager replays of the decode step; the timed region runs it inside a HIP graph"}
        # attention kernels, one object per head-dim instantiation (the north star's "attention-GEMM roofline": QK^T and PV on the MFMA)
        roofs.update(attn_roofs(am, traffic_of))
        if os.environ.get("VG_BENCH_ATTN_SHAPES") and rank == 0:      # per-shape table of the attention launches (stderr)
            by = {}
            for fl, e0, e1, dp, shp in am.rec:
                a = by.setdefault(shp, [0.0, 0.0, 0])
                a[0] += fl
                a[1] += e0.elapsed_time(e1)
                a[2] += 1
            for shp, (fl, ms, n) in sorted(by.items(), key=lambda kv: -kv[1][1]):
                print(f"attn (B,H,Sq,Skv,D)={shp}: {n} launches, {ms:.2f} ms, {1e3 * ms / n:.1f} us each, {fl / ms / 1e9:.1f} TF/s", file=sys.stderr)
        # "roofline" = the kernel with the most GPU time in a step; the others ride along under their own keys
        main = max(roofs, key=lambda k: roofs[k]["kernel_ms_per_step"])
        res["roofline"] = roofs.pop(main)
        if stale_traces:
            res["roofline"]["committed_traces_stale"] = sorted(stale_traces)      # taken on another build of the library: trace-derived fields omitted
        for k, v in roofs.items():
            res["roofline_" + k] = v
        # the mask-decoder GEMMs (S7/S8: two-way transformer projections, ConvT-as-GEMM upscaling, hypernetwork MLPs and product), over whatever
        # tile kernels they route to — a cross-section of the objects above, not an additional kernel
        md = roof(None, "every ops.linear launch inside SAM2.mask_decoder (two-way transformer q/k/v/out projections on [frames x objects, 4096, 256], "
                        "ConvTranspose-as-GEMM upscaling, token MLPs)", scope="mask_decoder")
        if md["launches"]:
            # these GEMMs are [frames x objects x 4096, 256] x [256, 128 | 256] projections: 86-170 flop per algorithmic byte against the
            # 312 flop/B ridge of the part — bound by HBM, not by the MFMA pipe: both fractions are reported, and the
            # MFMA fraction these shapes could reach if they streamed at the full 8 TB/s
            gbs = md["algorithmic_bytes_per_launch"] / (md["avg_launch_us"] * 1e-6) / 1e9 if md["avg_launch_us"] else 0.0
            ai = md["algorithmic_tflop_per_launch"] * 1e12 / max(md["algorithmic_bytes_per_launch"], 1)
            md.update({"bound": "hbm", "achieved_gbs": round(gbs, 1), "peak_gbs": 8000.0, "frac_hbm": round(gbs / 8000.0, 4), "frac_mfma": md["frac"],
                       "flop_per_algorithmic_byte": round(ai, 1), "mfma_frac_ceiling_at_hbm_peak": round(min(1.0, ai * 8e12 / 2.5e15), 3)})
            res["roofline_mask_decoder_gemm"] = md
        if dec_n and not args.tiny:
            c = cfg["llm"]
            hd = c["hidden"] // c["num_heads"]
            wbytes = 2.0 * (c["num_layers"] * (c["hidden"] * (c["num_heads"] + 2 * c["num_kv_heads"]) * hd + c["hidden"] * c["hidden"]
                                               + 3 * c["hidden"] * c["ffn"]) + c["vocab"] * c["hidden"])
            if args.decode_weights == "fp8":     # MLP and lm_head in fp8, attention projections in bf16
                wbytes -= 0.5 * 2.0 * (c["num_layers"] * 3 * c["hidden"] * c["ffn"] + c["vocab"] * c["hidden"])
            tbs = wbytes * dec_n / (dec_ms * 1e-3) / 1e12
            res["roofline_decode"] = {"bound": "hbm", "kernel": "decode step (HIP graph: decode_gemv_fast_kernel x4 (q|k|v with RoPE + append) + decode_attn2_kernel per layer)",
                                      "achieved": round(tbs * 1e3, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(tbs / 8.0, 4),
                                      "algorithmic_bytes_per_step": round(wbytes), "steps": dec_n, "ms_per_token": round(dec_ms / dec_n, 3)}
    if world == 1 and not use_video and not args.tiny and not args.no_video_record:
        res["video_branch"] = video_record(args, model, ops, (images, context, sam, ids))
    if world == 1 and not use_video and not args.tiny and not args.no_video_record:
        res["roofline_mask_decoder_c4clip"] = mask_decoder_record(model, device)
    is_c2 = (args.frames, args.src, args.objects, args.te) == (32, 1024, 1, 16) and args.llm == "llama3-8b" and (args.decode_weights, args.prefill) == ("bf16", "bf16")
    if world == 1 and not use_video and not args.tiny and not args.no_config_records and not args.no_video_record and is_c2:
        # the other single-GPU configurations of BASELINE.json on the same model, 3 timed steps each (C3 / C4's 8-GPU split need the node)
        res["configs"] = {"note": "BASELINE configs beside the headline (C2), timed after it on the same loaded model: inputs in HBM -> ids + masks on the host",
                          "c1": config_record(args, cfg, model, device, 8, 512, 1, 8),
                          "c4_clip_one_gpu_bf16": config_record(args, cfg, model, device, 64, 1024, 8, 16),
                          "c4_clip_one_gpu_fp8_llm": config_record(args, cfg, model, device, 64, 1024, 8, 16, fp8=True)}
    if world == 1 and not args.no_quality and not args.tiny:
        res["quality"] = quality(cfg, args, model, step, device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.tiny:      # the CPU leg runs at N = 1 only
        res["cpu_baseline"] = cpu_baseline(cfg, args)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()          # rank 0 may still be in its CPU-baseline leg: leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
