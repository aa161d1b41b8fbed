"""Multi-GPU sharding of the path: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

What shards naturally and what does not:
  * Hiera + FPN (dominant FLOPs) and the framewise mask decode are independent per frame -> contiguous
    blocks of frames per rank, NO communication inside.
  * The LLM side is a single sequence with no tensor parallelism in the reference -> replicated on every
    rank (identical greedy ids by construction).  The only exchange the decode needs is the tiny
    [N,256] [SEG] embedding: one all-gather, rank 0's copy is used everywhere so all ranks decode with
    bit-identical prompts (<= 8 KB: latency-bound, one-shot, never a ring of buckets).
  * The two vision towers are independent per CLIP frame / per 4-frame InternVideo2 chunk: each rank encodes and
    projects its block of frames / chunks, the projected, pooled tokens (a few MB) are all-gathered, so the replicated LLM
    sees identical visual tokens everywhere and the towers cost 1/world of a clip instead of a whole one per rank.
  * Results: gather_masks=True (default = the reference's contract: inference() returns the whole clip's video_segments) all-gathers
    the uint8 masks on the devices and every rank copies the whole clip to its host; "rank0": the same device all-gather, but only
    rank 0 — the rank a caller reads — pays the host copy of the whole clip, the others keep their own frames / objects; False
    (explicit opt-in, bench.py --scaling weak): no data-path collective, every rank returns the masks of ITS frames (framewise) /
    ITS objects (video branch) under their global indices (33.5 MB per 32 frames at 1024^2: the host copy of an N-times larger
    result on every rank is what breaks weak scaling).
  * Video-branch propagation is a recurrence over frames (memory of t-1..t-6), so frames do not shard there — OBJECTS do
    (non_overlap_masks_for_mem_enc is unset: objects never interact, R/.../sam2_video_predictor.py:571-612): the per-frame Hiera
    features are all-gathered (8.4 MB bf16 per frame at SAM2-L), rank r propagates its block of the N [SEG] objects and the
    uint8 masks are all-gathered along the object axis.  N = 1 is "replicas only" for this stage (every rank runs it).
Frame / object counts need not divide by the world size: blocks differ by at most one unit and every collective is padded
to the largest block.
"""
import os

import torch
import torch.distributed as dist

from . import ops


class FrameSharder:
    def __init__(self, group=None, gather_masks=True, profile=False, stream_features=None, stream_steps=4):
        """stream_features (video branch): True = the FPN features are all-gathered chunk by chunk, on a communicator of their own, while Hiera
        still runs (hiera_all_frames); False = one exchange after the last frame (gather_frame_feats).  Default: VG_FEATURES_STREAMED, off —
        the streamed order has not been timed on an N-GPU node yet (no such node was reachable in r04 / r05)."""
        assert dist.is_initialized(), "init torch.distributed first (torchrun: one process per GPU)"
        assert gather_masks in (True, False, "rank0")
        self.group = group
        self.gather_masks = gather_masks
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.stream_features = (os.environ.get("VG_FEATURES_STREAMED", "0") == "1") if stream_features is None else bool(stream_features)
        self.stream_steps = max(1, int(stream_steps))
        # a process group runs its collectives in order on ONE internal stream: the streamed feature gathers get their own communicator, or the
        # text side's small collectives (tower tokens, K/V rows, [SEG]) would queue behind every Hiera chunk of the slowest rank.  new_group is
        # collective over the default group, and every rank constructs its FrameSharder with the same arguments.
        self.feat_group = group
        if self.world > 1:
            # new_group below is collective: ranks that disagree on the setting (an env var set on some of them) would deadlock — agree first, loudly
            flag = torch.tensor([int(self.stream_features)], dtype=torch.int64)
            lo, hi = flag.clone(), flag.clone()
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
            lo, hi = lo.to(dev), hi.to(dev)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
            if int(lo) != int(hi):
                raise RuntimeError("FrameSharder: the ranks disagree on stream_features / VG_FEATURES_STREAMED")
        if self.stream_features and self.world > 1:
            ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
            self.feat_group = dist.new_group(ranks=ranks, backend=dist.get_backend(group))
        # profile (bench.py --gpus N): every collective is bracketed by events on the stream that issues it and its payload is counted, so that a
        # scaling record shows what was exchanged, how often and for how long — collective_report()
        self.profile = profile
        self._events = {}

    # ---- per-collective accounting (r04)
    def _timed(self, name, nbytes, fn):
        if not self.profile or not torch.cuda.is_available():
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self._events.setdefault(name, []).append((e0, e1, int(nbytes)))
        return out

    def collective_report(self, reset=True):
        """{name: {calls, bytes_received_per_rank, ms}} of the collectives since the last report (device time between the bracketing events on
        the issuing stream: includes the wait for the slowest rank to arrive)."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        rep = {}
        for name, evs in self._events.items():
            rep[name] = dict(calls=len(evs), bytes_received_per_rank=sum(b for _, _, b in evs), ms=round(sum(a.elapsed_time(b) for a, b, _ in evs), 3))
        if reset:
            self._events = {}
        return rep

    def my_frames(self, T):
        """this rank's contiguous block of the T frames (sizes differ by at most one; a rank beyond T gets none)."""
        start, n = self.block(T)
        return list(range(start, start + n))

    def counts(self, n):
        base, extra = divmod(n, self.world)
        return [base + (1 if r < extra else 0) for r in range(self.world)]

    def gather_blocks(self, local, n, dim, shape, dtype, device, name="block_all_gather"):
        """all-gather of per-rank blocks of unequal length along `dim`: `local` is this rank's block (or None when it has no
        unit), `shape` the full result's shape with n units along dim.  Blocks are padded to the largest so the collective is regular."""
        counts = self.counts(n)
        most = max(counts)
        pshape = list(shape)
        pshape[dim] = most
        buf = torch.zeros(pshape, dtype=dtype, device=device)
        if local is not None and local.shape[dim]:
            buf.narrow(dim, 0, local.shape[dim]).copy_(local)
        parts = self._all_gather(buf, name)
        return torch.cat([parts[r].narrow(dim, 0, counts[r]) for r in range(self.world) if counts[r]], dim=dim)

    def block(self, n):
        """contiguous block of n units for this rank (sizes differ by at most one; ranks beyond n get none) -> (start, count)."""
        base, extra = divmod(n, self.world)
        start = self.rank * base + min(self.rank, extra)
        return start, base + (1 if self.rank < extra else 0)

    def gather_rows(self, local, n, rows_per_unit, tail_shape, dtype, device, name="tower_tokens_all_gather"):
        """all-gather of per-unit row blocks of unequal count: `local` holds this rank's units ([count*rows_per_unit, *tail] or
        None when it has none); every rank gets all n units in order.  Blocks are padded to the largest count so that the
        collective is regular."""
        base, extra = divmod(n, self.world)
        most = base + (1 if extra else 0)
        buf = torch.zeros((most * rows_per_unit,) + tuple(tail_shape), dtype=dtype, device=device)
        if local is not None and local.shape[0]:
            buf[:local.shape[0]].copy_(local)
        parts = self._all_gather(buf, name)
        counts = [base + (1 if r < extra else 0) for r in range(self.world)]
        return torch.cat([parts[r][:counts[r] * rows_per_unit] for r in range(self.world) if counts[r]], dim=0)

    def _all_gather(self, t, name="all_gather"):
        nbytes = t.numel() * t.element_size() * self.world
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            # plumbing runs with several ranks on ONE GPU (VG_DIST_BACKEND=gloo; tests/test_dist_hip.py): gloo moves host memory
            def via_host():
                host = [torch.empty(t.shape, dtype=t.dtype) for _ in range(self.world)]
                dist.all_gather(host, t.contiguous().cpu(), group=self.group)
                return [h.to(t.device) for h in host]
            return self._timed(name + " [gloo: host-staged]", nbytes, via_host)

        def on_device():       # RCCL: device buffers, the collective is ordered on the current stream
            bufs = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(bufs, t.contiguous(), group=self.group)
            return bufs
        return self._timed(name, nbytes, on_device)

    def all_gather_into(self, recv, send, name="kv_all_gather"):
        """recv [world * m, ...] <- every rank's send [m, ...], in rank order."""
        nbytes = recv.numel() * recv.element_size()
        if send.is_cuda and dist.get_backend(self.group) == "gloo":
            recv.copy_(torch.cat(self._all_gather(send, name), dim=0))
        else:
            self._timed(name, nbytes, lambda: dist.all_gather(list(recv.chunk(self.world)), send, group=self.group))

    def sync_seg_embeddings(self, emb):
        """all-gather of the [N,256] [SEG] embeddings; every rank adopts rank 0's copy."""
        return self._all_gather(emb, "seg_all_gather")[0]

    def framewise(self, sam2, images_for_sam, emb, hw, frame_feats=None, binarize=None):
        """frame-sharded Hiera + mask decode -> (device uint8 masks [frames, N, H, W], their global frame indices): this rank's frames,
        or the whole clip when gather_masks is set.
        frame_feats: optional precomputed Hiera features of THIS rank's frames ({frame: [3 levels]});
        binarize: logits -> uint8 masks of this rank's frames (per-frame work, so it shards with them); None = logit > 0, made
        in one pass from the low-res logits."""
        emb = self.sync_seg_embeddings(emb)
        T, N = images_for_sam.shape[0], emb.shape[0]
        frames = self.my_frames(T)
        local = None
        if frames:
            out, _ = sam2.framewise_branch(images_for_sam, emb, hw, frames=frames, frame_feats=frame_feats, as_masks=binarize is None)
            local = out if binarize is None else binarize(out)             # [frames of this rank, N, H, W] uint8, on device
        mine = (local if local is not None else torch.zeros((0, N) + tuple(hw), dtype=torch.uint8, device=emb.device)), frames
        if not self.gather_masks:
            return mine
        full = self.gather_blocks(local, T, 0, (T, N) + tuple(hw), torch.uint8, emb.device, name="mask_gather"), list(range(T))
        return full if (self.gather_masks is True or self.rank == 0) else mine

    def video_branch_objects(self, sam2, images_for_sam, emb, hw, frame_feats, binarize=None, **kw):
        """object-sharded SAM2 propagation: every rank holds the Hiera features of ALL frames (gather_frame_feats) and runs the
        recurrence for its block of the N objects (they never interact) -> (device uint8 masks [T, objects, H, W], their global object
        indices): this rank's objects, or all N (all-gathered along the object axis) when gather_masks is set.  With fewer objects
        than ranks the surplus ranks hold none; N = 1: replicas only — every rank propagates the single object (no exchange)."""
        T, N = images_for_sam.shape[0], emb.shape[0]

        def propagate(e):
            # the same launch sequence as the single-GPU path (model.inference_video_branch): replayed from a HIP graph unless
            # VG_VIDEO_GRAPH=0 (the graph key includes the number of objects, so per-rank object blocks get their own capture)
            if emb.device.type == "cuda" and os.environ.get("VG_VIDEO_GRAPH", "1") == "1" and not kw:
                return sam2.video_branch_graphed(images_for_sam, e, hw, frame_feats, as_masks=binarize is None)
            return sam2.video_branch(images_for_sam, e, hw, frame_feats=frame_feats, as_masks=binarize is None, **kw)

        if N == 1:
            out = propagate(emb)
            return (out if binarize is None else binarize(out)), [0]
        o0, on = self.block(N)
        local = None
        if on:
            out = propagate(emb[o0:o0 + on])
            local = out if binarize is None else binarize(out)             # [T, objects of this rank, H, W]
        mine = (local if local is not None else torch.zeros((T, 0) + tuple(hw), dtype=torch.uint8, device=emb.device)), list(range(o0, o0 + on))
        if not self.gather_masks:
            return mine
        full = self.gather_blocks(local, N, 1, (T, N) + tuple(hw), torch.uint8, emb.device, name="mask_gather"), list(range(N))
        return full if (self.gather_masks is True or self.rank == 0) else mine

    def gather_frame_feats(self, local_feats, T, sam2):
        """all-gather per-frame FPN features ({frame: [3 levels]} of this rank's frames) -> the same for all T frames.
        The level shapes come from the SAM2 configuration (forward_image: [S/4,S/4,32], [S/8,S/8,64], [S/16,S/16,256]), not from
        a local frame, so a rank without frames (T < world size) takes part in the collective with an empty block."""
        frames = self.my_frames(T)
        S = sam2.S
        level_shapes = sam2.level_shapes()
        levels = []
        for lv in range(3):
            stacked = torch.cat([local_feats[t][lv] for t in frames], dim=0) if frames else None      # [frames of this rank, h, w, c]
            assert stacked is None or tuple(stacked.shape[1:]) == level_shapes[lv], (tuple(stacked.shape), level_shapes[lv])
            levels.append(self.gather_blocks(stacked, T, 0, (T,) + level_shapes[lv], sam2.dtype, sam2.device, name="feature_all_gather"))
        return {t: [levels[lv][t:t + 1] for lv in range(3)] for t in range(T)}

    def stream_plan(self, T, frame_chunk):
        """(chunk, steps, counts) of the streamed exchange: the largest per-rank block is cut into about `stream_steps` chunks (never more
        frames than SAM2.frame_chunk per Hiera launch group, never fewer than one), so that several exchanges exist to overlap with the
        remaining chunks — r04 used frame_chunk (16) itself: one step and 4x zero padding at 8 ranks x 4 frames."""
        counts = self.counts(T)
        most = max(counts)
        ch = max(1, min(max(1, frame_chunk), -(-most // self.stream_steps)))
        return ch, -(-most // ch) if most else 0, counts

    def hiera_all_frames(self, sam2, images_for_sam):
        """frame-sharded Hiera + FPN with the features STREAMED to every rank (video branch: the object ranks need all frames): the rank's frames
        go through Hiera in chunks (stream_plan) and a chunk's three levels are all-gathered — asynchronously, on the feature communicator, under
        RCCL — while the next chunk is being computed, instead of one exchange of the whole clip's features after the last frame (8.4 MB per
        frame at SAM2-L).  Every rank runs the same number of steps; a step's blocks are padded to the largest block any rank has IN THAT STEP
        (blocks differ by at most one frame, so the padding is at most one frame per rank and step) -> {frame: fpn levels} for all T frames."""
        T = images_for_sam.shape[0]
        frames = self.my_frames(T)
        ch, steps, counts = self.stream_plan(T, sam2.frame_chunk)
        level_shapes = sam2.level_shapes()
        starts = [sum(counts[:r]) for r in range(self.world)]
        asyn = images_for_sam.is_cuda and dist.get_backend(self.feat_group) != "gloo"
        es = torch.empty((), dtype=sam2.dtype).element_size()
        pending, out = [], {}
        for st in range(steps):
            mine = frames[st * ch:(st + 1) * ch]
            n_step = [max(0, min(ch, counts[r] - st * ch)) for r in range(self.world)]
            pad = max(n_step)
            feats = sam2.hiera_frames(images_for_sam, mine) if mine else {}
            for lv in range(3):
                per_frame = es
                for d in level_shapes[lv]:
                    per_frame *= d
                payload = sum(n_step) * per_frame                      # what the ranks actually hold in this step, without the padding
                send = torch.zeros((pad,) + level_shapes[lv], dtype=sam2.dtype, device=sam2.device)
                if mine:
                    send[:len(mine)].copy_(torch.cat([feats[t][lv] for t in mine], dim=0))
                if asyn:
                    recv = torch.empty((self.world * pad,) + level_shapes[lv], dtype=sam2.dtype, device=sam2.device)
                    e0 = None
                    if self.profile:
                        e0 = torch.cuda.Event(enable_timing=True)
                        e0.record()
                    work = dist.all_gather_into_tensor(recv, send, group=self.feat_group, async_op=True)
                    pending.append((st, lv, recv, work, pad, n_step, e0, payload))
                else:
                    parts = self._all_gather(send, "feature_all_gather (streamed)")
                    pending.append((st, lv, torch.cat(parts, dim=0), None, pad, n_step, None, payload))
        for st, lv, recv, work, pad, n_step, e0, payload in pending:
            if work is not None:
                work.wait()             # the current stream waits for the collective; the host does not
                if e0 is not None:      # issue -> completion as seen by the consuming stream (the time the exchange had to hide in)
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    self._events.setdefault("feature_all_gather (streamed, issue to consumed)", []).append((e0, e1, int(payload)))
            for r in range(self.world):
                for j in range(n_step[r]):
                    t = starts[r] + st * ch + j
                    out.setdefault(t, [None, None, None])[lv] = recv[r * pad + j:r * pad + j + 1]
        return out
