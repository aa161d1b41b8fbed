"""N>1 path on CPU: world_size-2 gloo processes running the frame-sharded SAM2 decode (videoglamm_amd/dist.py)
must reproduce the single-process result bit for bit (operators swapped for tests/_cpu_ops.py)."""
import os
import sys

import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist

    import _cpu_ops
    import _golden as G
    from oracle import seeded
    from videoglamm_amd import ops
    from videoglamm_amd.dist import FrameSharder
    from videoglamm_amd.params import Params
    from videoglamm_amd.sam2 import SAM2

    torch.set_grad_enabled(False)
    torch.set_num_threads(2)
    os.environ["VG_TOWERS_SHARDED"] = "1"          # the opt-in tower sharding is what this test exercises
    for name in _cpu_ops.ALL:
        if hasattr(ops, name):
            setattr(ops, name, getattr(_cpu_ops, name))
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sd = G.weights("sam2_micro_manifest.json", 1, seeded.sam2_overrides())
    m = SAM2(Params(sd, "cpu", torch.float32), "", G.sam2_cfg())
    T, N, hw = 4, 2, (40, 56)
    images = G.rnd((T, 3, 256, 256), 11)
    text = G.rnd((N, 256), 12, 0.5)
    comm = FrameSharder()                           # default = the reference's contract: every rank returns the whole clip
    assert comm.gather_masks is True
    assert comm.my_frames(T) == [2 * rank, 2 * rank + 1]
    # a rank-dependent perturbation must be overwritten by rank 0's copy
    emb = comm.sync_seg_embeddings(text + 0.01 * rank)
    assert torch.equal(emb, text)
    masks, fids = comm.framewise(m, images, text + 0.01 * rank, hw)
    assert fids == [0, 1, 2, 3]
    shard = FrameSharder(gather_masks=False)        # explicit opt-in: no mask collective, every rank keeps its own frames / objects
    r0 = FrameSharder(gather_masks="rank0")         # rank 0 (the rank a caller reads) gets the whole clip, the others their own frames
    m0, f0 = r0.framewise(m, images, text, hw)
    assert (f0 == [0, 1, 2, 3] and torch.equal(m0, masks)) if rank == 0 else (f0 == [2, 3] and torch.equal(m0, masks[2:4]))
    local, lf = shard.framewise(m, images, text, hw)
    assert lf == [2 * rank, 2 * rank + 1] and torch.equal(local, masks[2 * rank:2 * rank + 2])
    feats = comm.hiera_all_frames(m, images)
    # the streamed exchange on its own communicator: chunks sized from the per-rank frame count (2 frames -> 2 steps of 1 frame, not one step
    # padded to SAM2.frame_chunk), same features as the single exchange after the last frame
    streamer = FrameSharder(stream_features=True)
    assert streamer.stream_features and streamer.feat_group is not None and not comm.stream_features
    assert streamer.stream_plan(T, m.frame_chunk)[:2] == (1, 2) and streamer.stream_plan(32, 16)[:2] == (4, 4) and streamer.stream_plan(1, 16)[:2] == (1, 1)
    feats_s = streamer.hiera_all_frames(m, images)
    feats_u = comm.gather_frame_feats(m.hiera_frames(images, comm.my_frames(T)), T, m)
    assert sorted(feats_s) == sorted(feats_u) == sorted(feats) == list(range(T))
    assert all(torch.equal(a, b) and torch.equal(a, c) for t in range(T) for a, b, c in zip(feats_s[t], feats_u[t], feats[t]))
    vid = m.video_branch(images, emb, hw, frame_feats=feats)
    # object-sharded propagation: rank r runs the recurrence for its object, masks all-gathered along the object axis
    vid_obj, oids = comm.video_branch_objects(m, images, emb, hw, feats)
    lobj, lo = shard.video_branch_objects(m, images, emb, hw, feats)
    assert oids == [0, 1] and lo == [rank] and torch.equal(lobj, vid_obj[:, rank:rank + 1])
    # uneven splits: 3 frames / 3 objects over 2 ranks (blocks 2 + 1), 1 frame (rank 1 has none)
    assert FrameSharder.my_frames(comm, 3) == ([0, 1] if rank == 0 else [2]) and FrameSharder.my_frames(comm, 1) == ([0] if rank == 0 else [])
    text3 = G.rnd((3, 256), 13, 0.5)
    masks3, _ = comm.framewise(m, images[:3], text3, hw)
    feats3 = comm.hiera_all_frames(m, images[:3])
    vid3, _ = comm.video_branch_objects(m, images[:3], text3, hw, feats3)
    masks1, _ = comm.framewise(m, images[:1], text, hw)
    l1, lf1 = shard.framewise(m, images[:1], text, hw)
    assert lf1 == ([0] if rank == 0 else []) and l1.shape[0] == len(lf1)
    # T < world in the video branch: rank 1 owns no frame and joins the feature all-gather with an empty block (shapes from the config)
    feats1 = comm.hiera_all_frames(m, images[:1])
    assert sorted(feats1) == [0] and [tuple(f.shape) for f in feats1[0]] == [(1, 64, 64, 32), (1, 32, 32, 64), (1, 16, 16, 256)]
    assert all(torch.equal(a, b) for a, b in zip(feats1[0], feats[0]))
    # vision towers sharded by CLIP frame / InternVideo2 chunk (Te = 4: one chunk -> rank 1 has none; two frames each)
    from test_oracle_e2e import e2e_setup
    from videoglamm_amd.vlm import VisionTowers
    _, esd, ecfg, inp = e2e_setup()
    towers = VisionTowers(Params(esd, "cpu", torch.float32), ecfg)
    vis = towers.encode(inp["images"], inp["context_images"], comm)
    both = [torch.empty_like(vis) for _ in range(world)]
    dist.all_gather(both, vis)
    same_everywhere = all(torch.equal(b, vis) for b in both)
    vis_ref = towers.encode(inp["images"], inp["context_images"])
    ok_towers = same_everywhere and vis.shape == vis_ref.shape and bool(torch.allclose(vis, vis_ref, rtol=1e-5, atol=1e-5))
    assert comm.block(5) == ((0, 3) if rank == 0 else (3, 2)) and comm.block(1) == ((0, 1) if rank == 0 else (1, 0))
    # sequence-parallel prefill: rows split over the ranks, K/V all-gathered per layer; decode replicated afterwards
    from videoglamm_amd.vlm import LlamaDecoder
    lc = G.configs.LLAMA_TINY
    lsd = {"model." + k: v for k, v in G.weights("llama_tiny_manifest.json", 4).items()}
    xs = G.rnd((1, 45, lc["hidden"]), 33)[0]
    d1 = LlamaDecoder(Params(lsd, "cpu", torch.float32), lc, 64, use_graph=False)
    last = d1.forward_sharded(xs, comm)
    d0 = LlamaDecoder(Params(lsd, "cpu", torch.float32), lc, 64, use_graph=False)
    ref_h = d0.forward(xs)
    ok_sp = bool(torch.allclose(last, ref_h[-1:], rtol=1e-5, atol=1e-5)) and d1.pos == 45 and int(d1.pos_dev[0]) == 45
    ok_sp = ok_sp and bool(torch.allclose(d1.hid_all[:45], ref_h, rtol=1e-5, atol=1e-5))        # every prompt row on every rank
    ok_sp = ok_sp and all(bool(torch.allclose(d1.kc[i][:45], d0.kc[i][:45], rtol=1e-5, atol=1e-5)) and
                          bool(torch.allclose(d1.vc[i][:45], d0.vc[i][:45], rtol=1e-5, atol=1e-5)) for i in range(lc["num_layers"]))
    step1, step0 = d1.forward(xs[:1] * 0.5), d0.forward(xs[:1] * 0.5)          # one more row through both caches
    ok_sp = ok_sp and bool(torch.allclose(step1, step0, rtol=1e-5, atol=1e-5))
    if rank == 0:
        q.put(("towers", ok_towers and ok_sp, (tuple(vis.shape), ok_towers, ok_sp)))
        ref_logits, _ = m.framewise_branch(images, text, hw)
        ref_vid = m.video_branch(images, text, hw)
        same = lambda a, b: bool(a.shape == b.shape) and float((a != b).float().mean()) < 1e-4   # noqa: E731  (per-object batches: summation order)
        ok_obj = same(vid_obj, (ref_vid > 0).to(torch.uint8))
        ref3, _ = m.framewise_branch(images[:3], text3, hw)
        ok_uneven = bool(torch.equal(masks3, (ref3 > 0).to(torch.uint8))) and same(vid3, (m.video_branch(images[:3], text3, hw) > 0).to(torch.uint8))
        ok_uneven = ok_uneven and bool(torch.equal(masks1, (ref_logits[:1] > 0).to(torch.uint8)))
        q.put((bool(torch.equal(masks, (ref_logits > 0).to(torch.uint8))), bool(torch.equal(vid, ref_vid)) and ok_obj and ok_uneven,
               tuple(masks.shape)))
    dist.barrier()
    dist.destroy_process_group()


def _worker8(rank, world, port, q):
    """world 8 (BASELINE configs C3 / C4's split): the partition maths and every collective of dist.py with 8 participants, on the micro model"""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist

    import _cpu_ops
    import _golden as G
    from oracle import seeded
    from videoglamm_amd import ops
    from videoglamm_amd.dist import FrameSharder
    from videoglamm_amd.params import Params
    from videoglamm_amd.sam2 import SAM2

    torch.set_grad_enabled(False)
    torch.set_num_threads(1)
    for name in _cpu_ops.ALL:
        if hasattr(ops, name):
            setattr(ops, name, getattr(_cpu_ops, name))
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sd = G.weights("sam2_micro_manifest.json", 1, seeded.sam2_overrides())
    m = SAM2(Params(sd, "cpu", torch.float32), "", G.sam2_cfg())
    hw = (40, 56)
    comm = FrameSharder()
    shard = FrameSharder(gather_masks=False)
    # ---- the partition itself: C3 = 32 frames / 8, C4 = 64 frames / 8 and 8 objects / 8, T < world, ragged
    assert comm.my_frames(32) == list(range(4 * rank, 4 * rank + 4)) and comm.my_frames(64) == list(range(8 * rank, 8 * rank + 8))
    assert comm.block(8) == (rank, 1) and comm.block(3) == ((rank, 1) if rank < 3 else (3, 0))
    cover = [None] * world
    dist.all_gather_object(cover, comm.my_frames(37))
    assert sorted(sum(cover, [])) == list(range(37)) and max(len(c) for c in cover) - min(len(c) for c in cover) <= 5      # blocks of ceil(37 / 8)
    # ---- C3: a 32-frame clip, one [SEG] object, frames sharded 8-way; the [SEG] embedding broadcast from rank 0; masks all-gathered
    T = 32
    images = G.rnd((T, 3, 256, 256), 61)
    text1 = G.rnd((1, 256), 62, 0.5)
    emb = comm.sync_seg_embeddings(text1 + 0.01 * rank)
    assert torch.equal(emb, text1)
    masks, fids = comm.framewise(m, images, emb, hw)
    local, lf = shard.framewise(m, images, emb, hw)
    assert fids == list(range(T)) and lf == comm.my_frames(T) and torch.equal(local, masks[lf[0]:lf[-1] + 1])
    # ---- C4's split: 8 [SEG] objects, one per rank, over all-gathered Hiera features (a 12-frame clip keeps the CPU run short; the frame
    #      count enters only through the partition checked above); every rank ends with every object's masks
    T4, N4 = 12, 8
    text8 = G.rnd((N4, 256), 63, 0.5)
    feats = comm.hiera_all_frames(m, images[:T4])
    assert sorted(feats) == list(range(T4))
    vid_obj, oids = comm.video_branch_objects(m, images[:T4], text8, hw, feats)
    lobj, lo = shard.video_branch_objects(m, images[:T4], text8, hw, feats)
    assert oids == list(range(N4)) and lo == [rank] and torch.equal(lobj, vid_obj[:, rank:rank + 1])
    # ---- T < world: 3 frames over 8 ranks (ranks 3..7 own none and still join every collective)
    m3, f3 = comm.framewise(m, images[:3], emb, hw)
    feats3 = comm.hiera_all_frames(m, images[:3])
    assert f3 == [0, 1, 2] and sorted(feats3) == [0, 1, 2]
    l3, lf3 = shard.framewise(m, images[:3], emb, hw)
    assert lf3 == ([rank] if rank < 3 else []) and l3.shape[0] == len(lf3)
    if rank == 0:
        ref_logits, _ = m.framewise_branch(images, text1, hw)
        ok_c3 = bool(torch.equal(masks, (ref_logits > 0).to(torch.uint8))) and bool(torch.equal(m3, (ref_logits[:3] > 0).to(torch.uint8)))
        ref_vid = (m.video_branch(images[:T4], text8, hw) > 0).to(torch.uint8)
        ok_c4 = bool(vid_obj.shape == ref_vid.shape) and float((vid_obj != ref_vid).float().mean()) < 1e-4      # per-object batches: summation order
        ok_feats = all(torch.equal(a, b) for t in range(3) for a, b in zip(feats3[t], feats[t]))
        q.put((ok_c3, ok_c4 and ok_feats, tuple(masks.shape), tuple(vid_obj.shape)))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_and_object_sharding_world8():
    """BASELINE configs C3 (32 frames sharded 8-way, RCCL all-gather of the [SEG] state) and C4's split (8 [SEG] objects over 8 ranks) as
    world-size-8 gloo processes on the CPU twins: the sharded result IS the single-process result."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    ok_c3, ok_c4, shape, vshape = q.get(timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert shape == (32, 1, 40, 56) and vshape == (12, 8, 40, 56)
    assert ok_c3, "C3: frame-sharded (8-way) framewise masks differ from the single-process result"
    assert ok_c4, "C4 split: object-sharded (8-way) propagation / T < world feature exchange differ from the single-process result"


def test_frame_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    tag, ok_towers, vshape = q.get(timeout=300)
    assert tag == "towers" and ok_towers, f"sharded vision towers / sequence-parallel prefill differ from the single-process result {vshape}"
    ok_fw, ok_vid, shape = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert shape == (4, 2, 40, 56)
    assert ok_fw, "frame-sharded framewise masks differ from the single-process result"
    assert ok_vid, "video branch on all-gathered Hiera features / object-sharded propagation / uneven splits differ from the single-process result"
