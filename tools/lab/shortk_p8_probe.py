"""Would the phase-split kernels (256x256 / 256x192 tiles) beat the 64-byte-step kernel on Hiera stage 1-2 shapes if K were padded to a multiple of 64
(weights zero-padded once, A over-read into the next row)?  Upper bound: A physically padded.  Run once per VG_GEMM_P8 setting (1 rule, 2 force 256^2, 3 force 256x192).
usage: VG_GEMM_P8=3 python tools/lab/shortk_p8_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videoglamm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [("s1 qkv", 1048576, 432, 144), ("s1 proj", 1048576, 144, 144), ("s2 qkv", 262144, 864, 288), ("s2 proj", 262144, 288, 288), ("s2 fc1", 262144, 1152, 288),
          ("s1->s2 proj", 1048576, 288, 144)]
knob = os.environ.get("VG_GEMM_P8", "1")
for name, M, N, K in shapes:
    Kp = (K + 63) // 64 * 64
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16) * 0.1
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.1
    b = torch.randn(N, device=dev, dtype=torch.float32)
    ap = torch.zeros(M, Kp, device=dev, dtype=torch.bfloat16); ap[:, :K] = a
    wp = torch.zeros(N, Kp, device=dev, dtype=torch.bfloat16); wp[:, :K] = w
    us = t(lambda: ops.linear(a, w, b))
    usp = t(lambda: ops.linear(ap, wp, b))
    err = (ops.linear(a, w, b).float() - ops.linear(ap, wp, b).float()).abs().max().item()
    gb = (M * K + M * N) * 2 / 1e9
    print(f"P8={knob} {name:12s} M={M:8d} N={N:5d} K={K:4d}: true K {us:7.1f} us (route {ops.gemm_route(M, N, K) if hasattr(ops, 'gemm_route') else '?'}) | K padded to {Kp} {usp:7.1f} us | "
          f"HBM roof {gb / 8e3 * 1e6:6.1f} us | max diff {err:.3g}", flush=True)
    del a, w, ap, wp
