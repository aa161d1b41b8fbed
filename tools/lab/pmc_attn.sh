#!/bin/bash
# lab: instruction mix and wait reasons of the flash attention kernel (SQ counters) on the pipeline's shapes (tools/attn_ab.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmcA /tmp/pmcB /tmp/pmcC
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/pmcA -o p -- python $R/tools/attn_ab.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM --kernel-trace -d /tmp/pmcB -o p -- python $R/tools/attn_ab.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS SQ_INSTS_WAVE32_LDS SQ_LDS_ADDR_CONFLICT --kernel-trace -d /tmp/pmcC -o p -- python $R/tools/attn_ab.py > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
for d in ("/tmp/pmcA", "/tmp/pmcB", "/tmp/pmcC"):
    g = glob.glob(d + "/**/*.db", recursive=True)
    if not g:
        print(d, "no db"); continue
    db = sqlite3.connect(g[0])
    q = ("select k.name, k.grid_x, k.grid_y, p.counter_name, count(*), avg(p.counter_value), avg(k.end-k.start)/1e3 from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
         "where (k.name like '%attn_kernel%' or k.name like '%attn64_kernel%' or k.name like '%attn_dma_kernel%') group by k.name, k.grid_x, k.grid_y, p.counter_name")
    cur = {}
    for name, gx, gy, ctr, n, avg, us in db.execute(q):
        cur.setdefault((name[:60], gx, gy, round(us)), {})[ctr] = avg
    for k, v in cur.items():
        print(k)
        print("   ", {c: f"{x:.4g}" for c, x in v.items()})
PY
