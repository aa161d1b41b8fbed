"""CU-masked streams (hipExtStreamCreateWithCUMask): how many CUs does the HBM-bound decode loop need, what does Hiera lose on a
subset of the CUs, and do the two co-run when they own disjoint CU sets?  usage: python tools/bench_cumask.py"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from videoglamm_amd import synth  # noqa: E402
from videoglamm_amd.model import VideoGLaMMForCausalLM  # noqa: E402
from videoglamm_amd.vlm import generate  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
NCU = 256


def masked_stream(bits):
    """bits: iterable of CU indices that stay enabled -> torch ExternalStream"""
    words = (ctypes.c_uint32 * (NCU // 32))()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), NCU // 32, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(st.value)


sys.argv = sys.argv[:1]
args = bench.parse()
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
cfg = synth.videoglamm_llama3_8b()
sd = synth.device_state_dict(synth.manifest(cfg), dev, torch.bfloat16)
model = synth.install_forced_tokens(VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.bfloat16, device=dev))
images, context, sam, ids = bench.make_inputs(cfg, args, 1, dev)
visual = torch.zeros(208 * args.te, cfg["llm"]["hidden"], dtype=torch.bfloat16, device=dev)
generate(model.P, model.cfg, model.towers, images, context, ids[0].cpu(), 2, visual=visual)
dec = model.P._decoder
S0, tok0 = dec.pos - 1, dec.tok_dev.clone()


def decode(n=31):
    for _ in range(n):
        dec.pos_dev.fill_(S0)
        dec.tok_dev.copy_(tok0)
        dec.pos = S0
        dec.decode_step()


def hiera():
    return model.sam2.hiera_frames(sam, None)


def wall(fn, stream=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if stream is None:
        fn()
    else:
        with torch.cuda.stream(stream):
            fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


decode(2); hiera()
print(f"unmasked: decode {wall(decode):.1f} ms, hiera {wall(hiera):.1f} ms")
for name, bits in [("all 256", range(256)), ("first 128", range(128)), ("every 2nd (128)", range(0, 256, 2)), ("every 4th (64)", range(0, 256, 4)),
                   ("first 64", range(64)), ("every 8th (32)", range(0, 256, 8)), ("every 3rd of 4 off: 192", [i for i in range(256) if i % 4 != 0])]:
    st = masked_stream(bits)
    with torch.cuda.stream(st):
        decode(2); hiera()
    print(f"mask {name:28s}: decode {wall(decode, st):7.1f} ms   hiera {wall(hiera, st):7.1f} ms")
# co-run on disjoint sets: decode on every 4th CU, Hiera on the other three
sd_, sh_ = masked_stream(range(0, 256, 4)), masked_stream([i for i in range(256) if i % 4 != 0])
for label, s_dec, s_hi in [("decode 64 | hiera 192", sd_, sh_), ("decode unmasked | hiera 192", None, sh_), ("both unmasked", None, None)]:
    def both():
        main = torch.cuda.current_stream()
        a = s_dec or main
        b = s_hi or side
        a.wait_stream(main); b.wait_stream(main)
        with torch.cuda.stream(b):
            hiera()
        with torch.cuda.stream(a):
            decode()
        main.wait_stream(a); main.wait_stream(b)
    side = torch.cuda.Stream()
    both()
    print(f"co-run {label:28s}: {wall(both):7.1f} ms")
