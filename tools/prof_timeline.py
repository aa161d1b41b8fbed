"""Timeline view of a rocprofv3 rocpd database: split the run into busy segments (separated by idle gaps > GAP ms),
and for the last N segments print wall span, busy time (union over queues), per-queue busy time, the idle-gap total
and the per-kernel time inside the segment.
usage: python tools/prof_timeline.py <results.db> [n_segments=1] [gap_ms=3]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
nseg = int(sys.argv[2]) if len(sys.argv) > 2 else 1
gap_ns = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 3e6
rows = list(db.execute("select name, queue_id, start, end from kernels order by start"))
segs, cur, cur_end = [], [], None
for r in rows:
    if cur and r[2] - cur_end > gap_ns:
        segs.append(cur)
        cur = []
        cur_end = None
    cur.append(r)
    cur_end = r[3] if cur_end is None else max(cur_end, r[3])
if cur:
    segs.append(cur)
print(f"{len(rows)} kernels, {len(segs)} busy segments (gap > {gap_ns / 1e6:g} ms)")
for s in segs[-nseg:]:
    t0, t1 = s[0][2], max(r[3] for r in s)
    busy, last = 0, t0
    for r in s:                       # union of intervals (sorted by start)
        a, b = max(r[2], last), r[3]
        if b > a:
            busy += b - a
            last = b
    perq = defaultdict(float)
    perk = defaultdict(lambda: [0, 0.0])
    for r in s:
        perq[r[1]] += (r[3] - r[2]) / 1e6
        k = perk[r[0]]
        k[0] += 1
        k[1] += (r[3] - r[2]) / 1e6
    print(f"\nsegment: {len(s)} kernels, span {(t1 - t0) / 1e6:.2f} ms, busy(union) {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms")
    print("  per queue busy ms:", {q: round(v, 2) for q, v in perq.items()})
    # coarse phases: 10 equal slices of the span with their top kernel
    n = 12
    for i in range(n):
        a, b = t0 + (t1 - t0) * i / n, t0 + (t1 - t0) * (i + 1) / n
        acc = defaultdict(float)
        for r in s:
            o = min(r[3], b) - max(r[2], a)
            if o > 0:
                acc[(r[1], r[0][:60])] += o / 1e6
        top = sorted(acc.items(), key=lambda kv: -kv[1])[:3]
        print(f"  [{(a - t0) / 1e6:7.1f},{(b - t0) / 1e6:7.1f}] ms: " + "; ".join(f"q{q}:{k} {v:.1f}" for (q, k), v in top))
    print(f"  {'ms':>9} {'n':>6} {'avg us':>9}  kernel")
    for name, (c, ms) in sorted(perk.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"  {ms:9.2f} {c:6d} {ms / c * 1e3:9.1f}  {name[:100]}")
