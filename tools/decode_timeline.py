"""Where a decode step's time sits: kernel durations and the idle gaps between consecutive kernels of the graph-replayed step, by role, from a rocprofv3
kernel trace of bench.py.  usage: python tools/decode_timeline.py <results.db>
Roles per layer (videoglamm_amd/vlm.py:_layers_decode): qkv GEMV (norm fused) -> attention -> o GEMV (+ residual) -> gate|up GEMV (norm + SwiGLU) -> down GEMV
(+ residual); per token additionally the lm_head (skinny GEMM), argmax, the row store and the position bump."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))


def role(name):
    if "decode_attn_kernel" in name or "decode_attn2_kernel" in name:
        return "attention"
    if "decode_gemv_fast_kernel" in name:
        if "short, true," in name or "float, true," in name:
            return "gate|up (norm + SwiGLU)"
        if name.rstrip(">(DecGemvArgs) ").endswith(", 64"):
            return "qkv (norm + RoPE + append)"          # r06: the RoPE form (last template argument = D/2)
        return "down (+ residual)" if ", 7, " in name else "qkv / o"
    return None


# decode steps = maximal runs of kernels between two lm_head-sized gaps: take every kernel between the first and the last decode kernel of the trace's LAST clip
idx = [i for i, r in enumerate(rows) if role(r[0])]
# split into clips by large gaps (> 20 ms between decode kernels)
clips, cur = [], [idx[0]]
for a, b in zip(idx, idx[1:]):
    if rows[b][1] - rows[a][2] > 20e6:
        clips.append(cur)
        cur = []
    cur.append(b)
clips.append(cur)
last = clips[-1]
lo, hi = last[0], last[-1]
seq = rows[lo:hi + 1]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end, qo, rope_form = None, 0, False
for name, s, e in seq:
    r = role(name)
    if r == "qkv (norm + RoPE + append)":
        rope_form = True
    if r == "qkv / o":
        r = "o (+ residual)" if rope_form else ("qkv (norm fused)" if qo % 2 == 0 else "o (+ residual)")
        qo += 1
    if r is None:
        r = "per-token tail: " + name.split("(")[0].replace("void ", "")[:60]
    dur[r] += (e - s) / 1e3
    cnt[r] += 1
    if prev_end is not None:
        gap[r] += max(0.0, (s - prev_end) / 1e3)
    prev_end = e
tokens = max(1, cnt["attention"] // 32)
span = (seq[-1][2] - seq[0][1]) / 1e3
print(f"decode loop of the trace's last clip: {len(seq)} launches, {tokens} tokens (32 layers), span {span / 1e3:.2f} ms = {span / tokens:.1f} us per token")
print(f"{'role':58s} {'launches':>8s} {'avg us':>8s} {'avg idle before (us)':>22s} {'us per token':>13s}")
tk = tg = 0.0
for r in sorted(dur, key=lambda k: -dur[k]):
    print(f"{r:58s} {cnt[r]:8d} {dur[r] / cnt[r]:8.2f} {gap[r] / cnt[r]:22.2f} {(dur[r] + gap[r]) / tokens:13.1f}")
    tk += dur[r]
    tg += gap[r]
print(f"kernel time {tk / tokens:.1f} us per token, idle between kernels {tg / tokens:.1f} us per token ({100 * tg / (tk + tg):.1f} % of the step)")
