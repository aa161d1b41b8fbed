"""Where the GPU sits idle inside a clip: union of the kernel intervals (all streams) of the LAST clip of a rocprofv3 kernel trace of bench.py, and the
largest gaps with the kernels on either side.  usage: python tools/gap_census.py <results.db> [min_gap_us=15] [rows=25]
The last clip = the kernels after the last gap > 5 ms that precedes an im2col / preprocessing kernel run (clips are separated by the host-side barrier of
bench.py's timed loop)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = list(db.execute("select name, start, end from kernels order by start"))
# clips: split where the GPU was idle for > 3 ms (host-side step boundary: synchronize + barrier + result handling)
cuts = [0]
busy_end = rows[0][2]
for i in range(1, len(rows)):
    if rows[i][1] - busy_end > 3e6:
        cuts.append(i)
    busy_end = max(busy_end, rows[i][2])
segs = [rows[a:b] for a, b in zip(cuts, cuts[1:] + [len(rows)])]
segs = [s for s in segs if len(s) > 2000]
print(f"{len(rows)} kernels, {len(segs)} clip-sized segments; the last one:")
seg = segs[-1]
t0, t1 = seg[0][1], max(r[2] for r in seg)
busy, gaps = 0.0, []
cur_s, cur_e, last_name = seg[0][1], seg[0][2], seg[0][0]
for name, s, e in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(((s - cur_e) / 1e3, (cur_e - t0) / 1e6, last_name, name))
        cur_s, cur_e = s, e
        last_name = name
    elif e > cur_e:
        cur_e = e
        last_name = name
busy += cur_e - cur_s
span = (t1 - t0) / 1e6
print(f"span {span:.2f} ms, some kernel running {busy / 1e6:.2f} ms, idle {span - busy / 1e6:.2f} ms in {len(gaps)} gaps "
      f"({sum(1 for g in gaps if g[0] >= min_gap)} of them >= {min_gap:.0f} us = {sum(g[0] for g in gaps if g[0] >= min_gap) / 1e3:.2f} ms)")
short = lambda n: n.replace("void ", "").split("(")[0][:48]  # noqa: E731
# idle by 10 % bins of the clip
bins = [0.0] * 10
for g in gaps:
    bins[min(9, int(g[1] / span * 10))] += g[0] / 1e3
print("idle ms per tenth of the clip:", " ".join(f"{b:.2f}" for b in bins))
print(f"{'gap us':>9s} {'at ms':>8s}  after -> before")
for g in sorted(gaps, key=lambda g: -g[0])[:nrows]:
    print(f"{g[0]:9.1f} {g[1]:8.2f}  {short(g[2])} -> {short(g[3])}")
