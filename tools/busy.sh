#!/bin/bash
# GPU busy fraction of the timed steps: union of kernel intervals vs wall span (rocprofv3 kernel trace of bench.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/busy
timeout 400 rocprofv3 --kernel-trace -d /tmp/busy -o busy -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-quality --no-roofline "$@" > /tmp/busy.log 2>&1 < /dev/null
grep '^{"metric' /tmp/busy.log | cut -c1-200
d=$(find /tmp/busy -name '*.db' | head -1)
if [ -z "$d" ]; then echo "no db"; tail -5 /tmp/busy.log; exit 1; fi
python - "$d" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = sorted(cur.execute("select start, end, name from kernels"))
t0, t1 = rows[0][0], rows[-1][1]
# find big gaps (> 20 ms) to split phases: load / warmup / steps
segs, cur_s, cur_e, busy = [], rows[0][0], rows[0][1], 0
gaps = []
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"span {(t1-t0)/1e6:.1f} ms, union busy {busy/1e6:.1f} ms, kernels {len(rows)}")
# last 3 steps: take the final 3*ms window
import json
gaps.sort(reverse=True)
print("largest gaps (ms, at ms-from-end, next kernel):")
for g, at, n in gaps[:25]:
    print(f"  {g/1e6:8.3f} at -{(t1-at)/1e6:8.1f}  {n[:80]}")
# busy fraction in the last 900 ms
for win in (300e6, 600e6, 900e6):
    lo = t1 - win
    b = 0; cs = ce = None
    for s, e, n in rows:
        if e < lo: continue
        s = max(s, lo)
        if cs is None: cs, ce = s, e
        elif s > ce: b += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    b += ce - cs
    print(f"last {win/1e6:.0f} ms: busy {b/1e6:.1f} ms = {b/win:.3f}")
import collections, re, os
step_ms = float(os.environ.get("STEP_MS", "308"))
lo = t1 - 3 * step_ms * 1e6
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    if s >= lo:
        k = re.sub(r"\(.*", "", n)[:90]
        agg[k][0] += e - s; agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"kernel time in the last 3 steps: {tot/3e6:.1f} ms/step (sum over streams)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{v[0]/3e6:8.2f} ms/step {v[1]/3:8.1f} launches  {v[0]/v[1]/1e3:8.1f} us  {k}")
PY
