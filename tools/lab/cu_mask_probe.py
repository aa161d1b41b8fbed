"""CU-partition probe: does a CU-masked HIP stream (hipExtStreamCreateWithCUMask) confine (a) eager kernels, (b) a replayed HIP graph — and what do the
HBM-bound decode loop and an MFMA-bound GEMM loop cost each other when they run on disjoint halves of the chip instead of time-slicing all of it?
usage: python tools/lab/cu_mask_probe.py [S=3361] [G=24]"""
import ctypes
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoglamm_amd import ops, synth  # noqa: E402
from videoglamm_amd.params import Params  # noqa: E402
from videoglamm_amd.vlm import LlamaDecoder  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    """bits: iterable of CU indices (0..NCU-1) the stream may use"""
    words = [0] * ((NCU + 31) // 32)
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(s.value)


def timed(fn, stream, n):
    with torch.cuda.stream(stream):
        fn()
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        stream.synchronize()
        return (time.perf_counter() - t0) / n


S = int(sys.argv[1]) if len(sys.argv) > 1 else 3361
G = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda:0")
print(f"{NCU} CUs", flush=True)

M = 8192
a = torch.randn(M, M, device=dev, dtype=torch.bfloat16) * 0.05
w = torch.randn(M, M, device=dev, dtype=torch.bfloat16) * 0.05
gemm = lambda: ops.linear(a, w)  # noqa: E731
# Hiera-like small-K shape (stage 3 fc1)
a3 = torch.randn(65536, 576, device=dev, dtype=torch.bfloat16) * 0.05
w3 = torch.randn(2304, 576, device=dev, dtype=torch.bfloat16) * 0.05
gemm3 = lambda: ops.linear(a3, w3)  # noqa: E731

masks = {
    "all": range(NCU),
    "low128": range(NCU // 2),
    "high128": range(NCU // 2, NCU),
    "even128": range(0, NCU, 2),
    "odd128": range(1, NCU, 2),
    "low64": range(NCU // 4),
    "low192": range(3 * NCU // 4),
    "high64": range(3 * NCU // 4, NCU),
    "high192": range(NCU // 4, NCU),
}
streams = {k: masked_stream(v) for k, v in masks.items()}
print("== eager GEMM 8192^3 / Hiera s3 fc1 under CU masks (us)")
for k, st in streams.items():
    print(f"  {k:8s}: {timed(gemm, st, 5) * 1e6:9.1f}   {timed(gemm3, st, 10) * 1e6:8.1f}", flush=True)

cfg = synth.videoglamm_llama3_8b()
man = {k: v for k, v in synth.vlm_manifest(cfg).items() if k.startswith(("model.layers.", "model.embed_tokens", "model.norm", "lm_head"))}
sd = synth.device_state_dict(man, dev, torch.bfloat16)
P = Params(sd, dev, torch.bfloat16)
llm = cfg["llm"]
dec = LlamaDecoder(P, llm, -(-(S + 4 * G + 2) // 1024) * 1024)
x = (torch.randn(S, llm["hidden"], device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 0.02).to(torch.bfloat16)
h = dec.forward(x)
dec.next_token(h[-1:])
dec.decode_step()          # eager + capture
torch.cuda.synchronize()


def decode_ms(stream, n=G):
    with torch.cuda.stream(stream):
        dec.decode_step()
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            dec.decode_step()
        stream.synchronize()
        return (time.perf_counter() - t0) * 1e3 / n


print("== graph-replayed decode step under CU masks (ms per token)")
for k in ("all", "low192", "low128", "even128", "low64"):
    print(f"  {k:8s}: {decode_ms(streams[k]):.3f}", flush=True)


def concurrent(dk, gk, fn, reps):
    """decode on streams[dk] while a GEMM loop runs on streams[gk] (host thread); returns (ms per token, us per GEMM)"""
    res = {}
    stop = threading.Event()

    def loop():
        st = streams[gk]
        n = 0
        with torch.cuda.stream(st):
            fn()
            st.synchronize()
            t0 = time.perf_counter()
            while not stop.is_set():
                for _ in range(reps):
                    fn()
                st.synchronize()
                n += reps
            res["us"] = (time.perf_counter() - t0) / max(n, 1) * 1e6

    th = threading.Thread(target=loop)
    th.start()
    time.sleep(0.05)
    ms = decode_ms(streams[dk], 2 * G)
    stop.set()
    th.join()
    return ms, res["us"]


print("== decode (mask A) while a GEMM loop runs (mask B): ms per token | us per GEMM")
for dk, gk in (("all", "all"), ("low128", "high128"), ("even128", "odd128"), ("low64", "high192"), ("low128", "all"), ("all", "high128")):
    ms, us = concurrent(dk, gk, gemm, 4)
    ms3, us3 = concurrent(dk, gk, gemm3, 16)
    print(f"  decode {dk:8s} | gemm {gk:8s}: 8192^3 {ms:.3f} ms/token, {us:9.1f} us/GEMM   |  s3 fc1 {ms3:.3f} ms/token, {us3:8.1f} us/GEMM", flush=True)
