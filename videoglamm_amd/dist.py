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
  * Results: every rank returns the masks of ITS frames (framewise) / ITS objects (video branch) under their global indices —
    no data-path collective; FrameSharder(gather_masks=True) all-gathers the uint8 masks so that every rank holds the whole clip
    (33.5 MB per 32 frames at 1024^2: the host copy of an N-times larger result on every rank is what breaks weak scaling).
  * Video-branch propagation is a recurrence over frames (memory of t-1..t-6), so frames do not shard there — OBJECTS do
    (non_overlap_masks_for_mem_enc is unset: objects never interact, R/.../sam2_video_predictor.py:571-612): the per-frame Hiera
    features are all-gathered (8.4 MB bf16 per frame at SAM2-L), rank r propagates its block of the N [SEG] objects and the
    uint8 masks are all-gathered along the object axis.  N = 1 is "replicas only" for this stage (every rank runs it).
Frame / object counts need not divide by the world size: blocks differ by at most one unit and every collective is padded
to the largest block.
"""
import torch
import torch.distributed as dist

from . import ops


class FrameSharder:
    def __init__(self, group=None, gather_masks=False):
        assert dist.is_initialized(), "init torch.distributed first (torchrun: one process per GPU)"
        self.group = group
        self.gather_masks = gather_masks
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def my_frames(self, T):
        """this rank's contiguous block of the T frames (sizes differ by at most one; a rank beyond T gets none)."""
        start, n = self.block(T)
        return list(range(start, start + n))

    def counts(self, n):
        base, extra = divmod(n, self.world)
        return [base + (1 if r < extra else 0) for r in range(self.world)]

    def gather_blocks(self, local, n, dim, shape, dtype, device):
        """all-gather of per-rank blocks of unequal length along `dim`: `local` is this rank's block (or None when it has no
        unit), `shape` the full result's shape with n units along dim.  Blocks are padded to the largest so the collective is regular."""
        counts = self.counts(n)
        most = max(counts)
        pshape = list(shape)
        pshape[dim] = most
        buf = torch.zeros(pshape, dtype=dtype, device=device)
        if local is not None and local.shape[dim]:
            buf.narrow(dim, 0, local.shape[dim]).copy_(local)
        parts = self._all_gather(buf)
        return torch.cat([parts[r].narrow(dim, 0, counts[r]) for r in range(self.world) if counts[r]], dim=dim)

    def block(self, n):
        """contiguous block of n units for this rank (sizes differ by at most one; ranks beyond n get none) -> (start, count)."""
        base, extra = divmod(n, self.world)
        start = self.rank * base + min(self.rank, extra)
        return start, base + (1 if self.rank < extra else 0)

    def gather_rows(self, local, n, rows_per_unit, tail_shape, dtype, device):
        """all-gather of per-unit row blocks of unequal count: `local` holds this rank's units ([count*rows_per_unit, *tail] or
        None when it has none); every rank gets all n units in order.  Blocks are padded to the largest count so that the
        collective is regular."""
        base, extra = divmod(n, self.world)
        most = base + (1 if extra else 0)
        buf = torch.zeros((most * rows_per_unit,) + tuple(tail_shape), dtype=dtype, device=device)
        if local is not None and local.shape[0]:
            buf[:local.shape[0]].copy_(local)
        parts = self._all_gather(buf)
        counts = [base + (1 if r < extra else 0) for r in range(self.world)]
        return torch.cat([parts[r][:counts[r] * rows_per_unit] for r in range(self.world) if counts[r]], dim=0)

    def _all_gather(self, t):
        bufs = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(bufs, t.contiguous(), group=self.group)
        return bufs

    def sync_seg_embeddings(self, emb):
        """all-gather of the [N,256] [SEG] embeddings; every rank adopts rank 0's copy."""
        return self._all_gather(emb)[0]

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
        if not self.gather_masks:
            return (local if local is not None else torch.zeros((0, N) + tuple(hw), dtype=torch.uint8, device=emb.device)), frames
        return self.gather_blocks(local, T, 0, (T, N) + tuple(hw), torch.uint8, emb.device), list(range(T))

    def video_branch_objects(self, sam2, images_for_sam, emb, hw, frame_feats, binarize=None, **kw):
        """object-sharded SAM2 propagation: every rank holds the Hiera features of ALL frames (gather_frame_feats) and runs the
        recurrence for its block of the N objects (they never interact) -> (device uint8 masks [T, objects, H, W], their global object
        indices): this rank's objects, or all N (all-gathered along the object axis) when gather_masks is set.  With fewer objects
        than ranks the surplus ranks hold none; N = 1: replicas only — every rank propagates the single object (no exchange)."""
        T, N = images_for_sam.shape[0], emb.shape[0]
        if N == 1:
            out = sam2.video_branch(images_for_sam, emb, hw, frame_feats=frame_feats, as_masks=binarize is None, **kw)
            return (out if binarize is None else binarize(out)), [0]
        o0, on = self.block(N)
        local = None
        if on:
            out = sam2.video_branch(images_for_sam, emb[o0:o0 + on], hw, frame_feats=frame_feats, as_masks=binarize is None, **kw)
            local = out if binarize is None else binarize(out)             # [T, objects of this rank, H, W]
        if not self.gather_masks:
            return (local if local is not None else torch.zeros((T, 0) + tuple(hw), dtype=torch.uint8, device=emb.device)), list(range(o0, o0 + on))
        return self.gather_blocks(local, N, 1, (T, N) + tuple(hw), torch.uint8, emb.device), list(range(N))

    def gather_frame_feats(self, local_feats, T):
        """all-gather per-frame FPN features ({frame: [3 levels]} of this rank's frames) -> the same for all T frames."""
        frames = self.my_frames(T)
        ref = next(iter(local_feats.values())) if local_feats else None
        levels = []
        for lv in range(3):
            stacked = torch.cat([local_feats[t][lv] for t in frames], dim=0) if frames else None      # [frames of this rank, h, w, c]
            if ref is None:
                raise ValueError("gather_frame_feats: a rank without frames cannot describe the feature shapes (T < world size)")
            shape = (T,) + tuple(ref[lv].shape[1:])
            levels.append(self.gather_blocks(stacked, T, 0, shape, ref[lv].dtype, ref[lv].device))
        return {t: [levels[lv][t:t + 1] for lv in range(3)] for t in range(T)}

    def hiera_all_frames(self, sam2, images_for_sam):
        """frame-sharded Hiera + FPN, features all-gathered level by level -> {frame: fpn levels}."""
        T = images_for_sam.shape[0]
        return self.gather_frame_feats(sam2.hiera_frames(images_for_sam, self.my_frames(T)), T)
