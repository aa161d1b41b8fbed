"""Decode-loop microbenchmark: Llama-3-8B (synthetic weights), prefill of S random embeddings, then G graph-replayed
decode steps; the 16 GB of weights stream from HBM every token (no MALL flattery as in a single-matrix loop).
usage: python tools/bench_decode.py [S=1697] [G=32]      env: VG_DECODE_FUSED=0/1"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoglamm_amd import synth  # noqa: E402
from videoglamm_amd.params import Params  # noqa: E402
from videoglamm_amd.vlm import LlamaDecoder  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1697
G = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
cfg = synth.videoglamm_llama3_8b()
man = {k: v for k, v in synth.vlm_manifest(cfg).items() if k.startswith(("model.layers.", "model.embed_tokens", "model.norm", "lm_head"))}
sd = synth.device_state_dict(man, dev, torch.bfloat16)
P = Params(sd, dev, torch.bfloat16)
llm = cfg["llm"]
nbytes = sum(v.numel() * 2 for k, v in sd.items() if "embed_tokens" not in k)
for fused in os.environ.get("BENCH_DECODE_FUSED", "0,1").split(","):
    os.environ["VG_DECODE_FUSED"] = fused
    dec = LlamaDecoder(P, llm, -(-(S + G + 2) // 1024) * 1024)
    x = (torch.randn(S, llm["hidden"], device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 0.02).to(torch.bfloat16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h = dec.forward(x)
    dec.next_token(h[-1:])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dec.decode_step()          # eager + capture
    toks = []
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    sync = os.environ.get("BENCH_DECODE_SYNC", "1") == "1"      # 1: read every token back before the next step (the r05 loop); 0: steps back to back
    for _ in range(G):
        dec.decode_step()
        if sync:
            toks.append(int(dec.tok_dev[0]))
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    ms = (t3 - t2) * 1e3 / G
    print(f"fused={fused} rope_path={dec.rope_path} kpw2={dec.kpw2} sync={int(sync)}: prefill S={S} {1e3 * (t1 - t0):.1f} ms; decode {ms:.3f} ms/token "
          f"({nbytes / ms / 1e9:.2f} TB/s of weight bytes); tokens {toks[:8]}")
    del dec
