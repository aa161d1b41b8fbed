#!/bin/bash
# lab: SQ counters (instruction mix, LDS conflicts, waits) of the tile GEMMs on selected shapes (tools/bench_gemm.py, VG_BENCH_SHAPES)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export VG_BENCH_GEMM_ONLY=1 VG_BENCH_SHAPES=${SHAPES:-"c2 llm,hiera s3 fc1,hiera s3 qkv,hiera s3 fc2,square 8k"}
rm -rf /tmp/pgA /tmp/pgB /tmp/pgC
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/pgA -o p -- python $R/tools/bench_gemm.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM --kernel-trace -d /tmp/pgB -o p -- python $R/tools/bench_gemm.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS SQ_LDS_ADDR_CONFLICT --kernel-trace -d /tmp/pgC -o p -- python $R/tools/bench_gemm.py > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
acc = {}
for d in ("/tmp/pgA", "/tmp/pgB", "/tmp/pgC"):
    g = glob.glob(d + "/**/*.db", recursive=True)
    if not g:
        print(d, "no db"); continue
    db = sqlite3.connect(g[0])
    q = ("select k.name, k.grid_x, p.counter_name, count(*), avg(p.counter_value), avg(k.end-k.start)/1e3 from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
         "where k.name like '%gemm_tile%' group by k.name, k.grid_x, p.counter_name")
    for name, gx, ctr, n, avg, us in db.execute(q):
        acc.setdefault((name.split("(")[0][-40:], gx), {"us": round(us)})[ctr] = avg
for k, v in sorted(acc.items()):
    m = v.get("SQ_INSTS_MFMA", 0) or 1
    print(k, f"us {v['us']}")
    print(f"    per MFMA: VALU {v.get('SQ_INSTS_VALU', 0) / m:.2f} SALU {v.get('SQ_INSTS_SALU', 0) / m:.2f} LDS {v.get('SQ_INSTS_LDS', 0) / m:.2f};"
          f" MFMA busy / wave cycles {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 4 / max(v.get('SQ_WAVE_CYCLES', 1), 1):.3f};"
          f" LDS bank conflict / LDS active {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f};"
          f" wait_inst_any / wave cycles {v.get('SQ_WAIT_INST_ANY', 0) / max(v.get('SQ_WAVE_CYCLES', 1), 1):.3f}; wait LDS {v.get('SQ_WAIT_INST_LDS', 0) / max(v.get('SQ_WAVE_CYCLES', 1), 1):.3f};"
          f" LDS active / wave cycles {v.get('SQ_LDS_IDX_ACTIVE', 0) / max(v.get('SQ_WAVE_CYCLES', 1), 1):.3f}")
PY
