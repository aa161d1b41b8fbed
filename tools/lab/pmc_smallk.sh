#!/bin/bash
# lab: instruction mix of the small-K GEMM (SQ counters) on the real C2 shapes
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export VG_BENCH_GEMM_ONLY=1 VG_BENCH_SHAPES="${SHAPES:-c2 s1 qkv,c2 s2 fc1,hiera s3 fc1}" VG_GEMM_RING64=0
rm -rf /tmp/pmcA /tmp/pmcB
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/pmcA -o p -- python $R/tools/bench_gemm.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --kernel-trace -d /tmp/pmcB -o p -- python $R/tools/bench_gemm.py > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
for d in ("/tmp/pmcA", "/tmp/pmcB"):
    db = sqlite3.connect(glob.glob(d + "/**/*.db", recursive=True)[0])
    q = ("select k.name, k.grid_x, k.grid_y, p.counter_name, count(*), avg(p.counter_value), avg(k.end-k.start)/1e3 from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
         "where k.name like '%gemm_tile%' group by k.name, k.grid_x, k.grid_y, p.counter_name")
    cur = {}
    for name, gx, gy, ctr, n, avg, us in db.execute(q):
        cur.setdefault((name[:60], gx, gy, round(us)), {})[ctr] = avg
    for k, v in cur.items():
        print(k)
        print("   ", {c: f"{x:.4g}" for c, x in v.items()})
PY
