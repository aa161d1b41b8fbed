#!/bin/bash
# same-box A/B of the r05 GEMM route: the shape rule with the 256x192 tile (default) against the rule without it (VG_GEMM_P8=4), twice each, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2; do
  for v in 1 4; do
    VG_GEMM_P8=$v python bench.py --no-cpu-baseline --no-quality --steps 5 > gpurun_out/r05_ab_$v.json 2> gpurun_out/r05_ab_$v.err
    python - <<PY
import json
r = json.load(open("gpurun_out/r05_ab_$v.json"))
st = r["stages"]
print("VG_GEMM_P8=$v  C2 %.2f ms  video %.2f ms  stages hiera %.2f towers %.2f prefill %.2f decode %.2f" % (r["ms_per_step"], r["video_branch"]["ms_per_step"], st["hiera_fpn"]["ms"], st["towers"]["ms"], st["prefill"]["ms"], st["decode"]["ms"]))
PY
  done
done
