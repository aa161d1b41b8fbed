"""What one rank pays for the sequence-parallel prefill (LlamaDecoder.forward_sharded) without the collectives: the LAST
chunk of the C1 prompt (S = 1697) through all layers against an already filled KV cache, for world = 1, 2, 4, 8.
usage: python tools/bench_prefill_chunk.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoglamm_amd import synth  # noqa: E402
from videoglamm_amd.params import Params  # noqa: E402
from videoglamm_amd.vlm import LlamaDecoder  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
cfg = synth.videoglamm_llama3_8b()
man = {k: v for k, v in synth.vlm_manifest(cfg).items() if k.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))}
sd = synth.device_state_dict(man, dev, torch.bfloat16)
dec = LlamaDecoder(Params(sd, dev, torch.bfloat16), cfg["llm"], 2048, use_graph=False)
S = 1697
x = torch.randn(S, cfg["llm"]["hidden"], device=dev, dtype=torch.bfloat16) * 0.1
for world in (1, 2, 4, 8):
    m = -(-S // world)
    a = (world - 1) * m          # the last rank's rows [a, S): the longest KV range
    dec.reset()
    if a:
        dec.forward(x[:a])
    ts = []
    for _ in range(reps + 1):
        dec.pos = a
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dec.forward(x[a:])
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"world {world}: rows {S - a:5d} of {S}  {min(ts[1:]):7.2f} ms per rank")
