"""BASELINE config C3's path ON THE HIP KERNELS: two ranks on the one MI355X of the GPU box (torch.distributed "gloo": the
collectives move host copies — RCCL needs one GPU per rank), the whole façade (VideoGLaMMForCausalLM with a FrameSharder) on
the reference-made end-to-end fixture.  Frame-sharded framewise decode must equal the single-process HIP result bit for bit and
the reference's masks; the object-sharded propagation (14 objects -> 7 + 7) and the opt-in tower / prefill sharding must
reproduce the reference's ids exactly and its masks (fp32 parity mode)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _stack(seg):
    return np.stack([np.stack([seg[t][k] for k in sorted(seg[t])]) for t in sorted(seg)])


def _worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist

    from test_oracle_e2e import e2e_setup
    from videoglamm_amd.dist import FrameSharder
    from videoglamm_amd.model import VideoGLaMMForCausalLM

    torch.set_grad_enabled(False)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    fx, sd, cfg, inp = e2e_setup()
    res = {}

    def run(model, ids_in, branch):
        out_ids, segs = model.inference([inp["images"]], [inp["context_images"]], [inp["images_for_sam"]], ids_in[None], [(1024, 1024)],
                                        [inp["original_size"]], max_new_tokens=inp["max_new_tokens"], use_sam2_video_branch=branch)
        return out_ids[0].tolist(), segs[0]

    def iou(a, b):
        return float((a & b).sum() / max((a | b).sum(), 1))

    single = VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.float32, device=dev)
    full = VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.float32, device=dev, comm=FrameSharder())                       # whole clip on every rank
    own = VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.float32, device=dev, comm=FrameSharder(gather_masks=False))     # every rank keeps its shard
    ids6, ids14 = inp["input_ids"].long(), fx["input_ids8"].long()
    # --- framewise, 6 objects, 4 frames -> 2 + 2
    ref_ids, ref_seg = run(single, ids6, False)
    got_ids, got_seg = run(full, ids6, False)
    res["fw_ids"] = got_ids == ref_ids == fx["framewise_output_ids"].long().tolist()
    res["fw_bitexact"] = sorted(got_seg) == [0, 1, 2, 3] and bool(np.array_equal(_stack(got_seg), _stack(ref_seg)))
    res["fw_ref_iou"] = iou(_stack(got_seg), fx["framewise_masks"].numpy() > 0.5)
    _, my_seg = run(own, ids6, False)
    res["fw_shard"] = sorted(my_seg) == [2 * rank, 2 * rank + 1] and all(np.array_equal(my_seg[t][k], ref_seg[t][k]) for t in my_seg for k in my_seg[t])
    # --- video branch, 14 objects -> 7 + 7 (Hiera frames 2 + 2, features all-gathered, graph-replayed propagation per rank)
    ref_ids, ref_seg = run(single, ids14, True)
    got_ids, got_seg = run(full, ids14, True)
    res["vid_ids"] = got_ids == ref_ids == fx["video8_output_ids"].long().tolist()
    a, b = _stack(got_seg), _stack(ref_seg)
    res["vid_objects"] = a.shape[1]
    res["vid_vs_single"] = float((a != b).mean())            # a 7-object batch is another summation order than a 14-object one
    res["vid_ref_iou"] = iou(a, fx["video8_masks"].numpy() > 0.5)
    _, my_seg = run(own, ids14, True)
    res["vid_shard"] = sorted(my_seg[0]) == list(range(7 * rank, 7 * rank + 7)) and sorted(my_seg) == [0, 1, 2, 3]
    # --- T = 1 < world in the video branch: rank 1 owns no frame and still joins every collective
    one = [inp["images"]], [inp["context_images"]], [inp["images_for_sam"][:1]], ids6[None], [(1024, 1024)], [inp["original_size"]]
    _, s1 = full.inference(*one, max_new_tokens=inp["max_new_tokens"], use_sam2_video_branch=True)
    _, r1 = single.inference(*one, max_new_tokens=inp["max_new_tokens"], use_sam2_video_branch=True)
    res["t1"] = sorted(s1[0]) == [0] and float((_stack(s1[0]) != _stack(r1[0])).mean()) < 1e-4
    # --- opt-in LLM-side sharding (what bench.py --gpus N switches on): towers by frame / chunk, sequence-parallel prefill
    os.environ["VG_TOWERS_SHARDED"] = os.environ["VG_PREFILL_SHARDED"] = "1"
    got_ids, got_seg = run(full, ids6, False)
    res["llm_sharded_ids"] = got_ids == fx["framewise_output_ids"].long().tolist()
    res["llm_sharded_iou"] = iou(_stack(got_seg), fx["framewise_masks"].numpy() > 0.5)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_c3_two_ranks_on_one_gpu(cuda):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=900) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank in (0, 1):
        r = got[rank]
        assert r["fw_ids"] and r["fw_bitexact"] and r["fw_shard"] and r["fw_ref_iou"] > 0.999, (rank, r)
        assert r["vid_ids"] and r["vid_objects"] == 14 and r["vid_shard"] and r["vid_vs_single"] < 1e-4 and r["vid_ref_iou"] > 0.999, (rank, r)
        assert r["t1"] and r["llm_sharded_ids"] and r["llm_sharded_iou"] > 0.999, (rank, r)


def _worker_c3_full(rank, world, port, q):
    """BASELINE config C3's REAL workload — 32 x 1024^2 SAM frames, Te = 16, Llama-3-8B + InternVideo2-1B + CLIP-L/336 + SAM2-L in bf16, one forced [SEG],
    32 greedy tokens — as two ranks on the one GPU of the box (gloo moves host copies): frames sharded 16 + 16 for Hiera + FPN and the mask decoder,
    the [SEG] embedding synchronised from rank 0, masks gathered; against the SAME model run unsharded in the same process."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist

    from videoglamm_amd import synth
    from videoglamm_amd.dist import FrameSharder
    from videoglamm_amd.model import VideoGLaMMForCausalLM

    torch.set_grad_enabled(False)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = synth.videoglamm_llama3_8b()
    cfg["forced_tokens"] = {8: cfg["seg_token_idx"]}
    sd = synth.device_state_dict(synth.manifest(cfg), dev, torch.bfloat16)
    model = synth.install_forced_tokens(VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.bfloat16, device=dev))
    del sd
    g = torch.Generator().manual_seed(1234)
    te, T = 16, 32
    images = torch.randn(te, 3, 224, 224, generator=g).to(dev)
    context = torch.randn(te, 3, 336, 336, generator=g).to(dev)
    sam = torch.randn(T, 3, 1024, 1024, generator=g).to(dev)
    ids = torch.cat([torch.tensor([1, 5, 6]), torch.full((te,), -200), torch.randint(3, cfg["llm"]["vocab"] - 2, (30,), generator=g)])[None]

    def run(frames=sam):
        out_ids, segs = model.inference([images], [context], [frames], ids, [(1024, 1024)], [(1024, 1024)], max_new_tokens=32, use_sam2_video_branch=False)
        return out_ids[0].tolist(), segs[0]

    ref_ids, ref_seg = run()                                   # the whole clip unsharded, this process
    _, half_seg = run(sam[16 * rank:16 * rank + 16])           # this rank's 16 frames unsharded: the batch composition of its shard
    model.comm = FrameSharder()                                # whole clip back on every rank
    got_ids, got_seg = run()
    a, b = _stack(got_seg), _stack(ref_seg)
    mine = a[16 * rank:16 * rank + 16]
    res = {"ids": got_ids == ref_ids, "n_ids": len(got_ids) - ids.shape[1], "frames": sorted(got_seg) == list(range(T)), "shape": tuple(a.shape),
           "whole_clip_bitexact": bool(np.array_equal(a, b)), "pixel_agreement_vs_unsharded_32": float((a == b).mean()), "mask_fraction": float(b.mean()),
           "own_frames_bitexact_vs_unsharded_16": bool(np.array_equal(mine, _stack(half_seg)))}
    inter, union = (a & b).sum(axis=(1, 2, 3)), (a | b).sum(axis=(1, 2, 3))
    res["min_frame_iou"] = float((inter / np.maximum(union, 1)).min())
    model.comm = FrameSharder(gather_masks=False)              # every rank keeps its 16 frames
    _, my_seg = run()
    res["shard"] = sorted(my_seg) == list(range(16 * rank, 16 * rank + 16)) and bool(np.array_equal(_stack(my_seg), mine))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_c3_full_size_two_ranks_on_one_gpu(cuda):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_c3_full, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=1500) for _ in range(2))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for rank in (0, 1):
        r = got[rank]
        print(f"  C3 full size, rank {rank}: {r}")
        assert r["ids"] and r["n_ids"] == 32 and r["frames"] and r["shape"] == (32, 1, 1024, 1024), (rank, r)
        # a rank's shard IS the unsharded run of its 16 frames, bit for bit (same launches on the same data); against the unsharded 32-frame run the
        # mask decoder's batch is 16 instead of 32 (frame, object) pairs — other GEMM tile routes, another bf16 rounding order: the random-weight
        # masks (58 % foreground, boundary everywhere) move in ~0.2 % of the pixels; ids are equal by construction (LLM side replicated)
        assert r["own_frames_bitexact_vs_unsharded_16"] and r["shard"], (rank, r)
        assert r["pixel_agreement_vs_unsharded_32"] > 0.99 and r["min_frame_iou"] > 0.97 and 0.0 < r["mask_fraction"] < 1.0, (rank, r)


def _worker_rccl(port, q):
    """ONE rank, backend "nccl" (= RCCL): the device-buffer branches of FrameSharder (dist.all_gather on device tensors, all_gather_into on
    recv.chunk() views, the async all_gather_into_tensor of the streamed features) — the code the driver's multi-GPU run takes, which the
    gloo runs above never enter."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist

    from test_oracle_e2e import e2e_setup
    from videoglamm_amd.dist import FrameSharder
    from videoglamm_amd.model import VideoGLaMMForCausalLM

    torch.set_grad_enabled(False)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    fx, sd, cfg, inp = e2e_setup()
    res = {"backend": dist.get_backend()}

    def run(model, ids_in, branch):
        out_ids, segs = model.inference([inp["images"]], [inp["context_images"]], [inp["images_for_sam"]], ids_in[None], [(1024, 1024)],
                                        [inp["original_size"]], max_new_tokens=inp["max_new_tokens"], use_sam2_video_branch=branch)
        return out_ids[0].tolist(), segs[0]

    single = VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.float32, device=dev)
    comm = FrameSharder(profile=True, stream_features=True)
    full = VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.float32, device=dev, comm=comm)
    ids6, ids14 = inp["input_ids"].long(), fx["input_ids8"].long()
    os.environ["VG_TOWERS_SHARDED"] = os.environ["VG_PREFILL_SHARDED"] = "1"
    for tag, ids_in, branch in (("fw", ids6, False), ("vid", ids14, True)):
        ref_ids, ref_seg = run(single, ids_in, branch)
        got_ids, got_seg = run(full, ids_in, branch)
        res[tag + "_ids"] = got_ids == ref_ids
        res[tag + "_bitexact"] = bool(np.array_equal(_stack(got_seg), _stack(ref_seg)))
        res[tag + "_pixel_agreement"] = float((_stack(got_seg) == _stack(ref_seg)).mean())
        res[tag + "_collectives"] = comm.collective_report()
    # the LLM-side sharding's collectives are only entered at world > 1: call their device-buffer branches directly
    send = torch.randn(5, 3, device=dev)
    recv = torch.empty(5, 3, device=dev)
    comm.all_gather_into(recv, send)
    rows = comm.gather_rows(send[:4], 2, 2, (3,), torch.float32, dev)
    blocks = comm.gather_blocks(send.view(1, 5, 3), 5, 1, (1, 5, 3), torch.float32, dev)
    torch.cuda.synchronize()
    res["direct"] = bool(torch.equal(recv, send) and torch.equal(rows, send[:4]) and torch.equal(blocks, send.view(1, 5, 3)))
    res["direct_collectives"] = comm.collective_report()
    comm.stream_features = False                      # the one-exchange-after-Hiera form of the feature gather (the default)
    got_ids, got_seg = run(full, ids14, True)
    res["vid_unstreamed_bitexact"] = bool(np.array_equal(_stack(got_seg), _stack(ref_seg)))
    res["vid_unstreamed_collectives"] = comm.collective_report()
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_device_buffer_branches_single_rank(cuda):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_rccl, args=(31500 + os.getpid() % 2000, q))
    p.start()
    r = q.get(timeout=900)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert r["backend"] == "nccl"
    # the streamed exchange runs Hiera in chunks of ceil(frames / stream_steps) frames, the single-process run in one batch: other GEMM tile routes,
    # another fp32 summation order (tests/test_fullsize_gpu.py::test_sam2_large_batched_equals_serial_fp32) — masks agree to a handful of pixels;
    # the single exchange after the last frame keeps the batch composition and is bit-exact
    assert r["fw_ids"] and r["fw_bitexact"] and r["vid_ids"] and r["vid_pixel_agreement"] > 0.9999 and r["vid_unstreamed_bitexact"], r
    fw, vid, un = r["fw_collectives"], r["vid_collectives"], r["vid_unstreamed_collectives"]
    assert not any("gloo" in k for k in list(fw) + list(vid) + list(un)), (fw, vid, un)          # the device-buffer branch ran
    assert {"seg_all_gather", "mask_gather"} <= set(fw) and fw["seg_all_gather"]["bytes_received_per_rank"] == 6 * 256 * 4, fw
    assert any(k.startswith("feature_all_gather (streamed") for k in vid) and "mask_gather" in vid, vid
    assert "feature_all_gather" in un and not any("streamed" in k for k in un), un
    assert r["direct"] and {"kv_all_gather", "tower_tokens_all_gather", "block_all_gather"} <= set(r["direct_collectives"]), r
