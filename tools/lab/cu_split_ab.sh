#!/bin/bash
# CU partition A/B on the bench line: VG_CU_SPLIT = CUs of the decode loop (0 = one full-chip side stream), VG_HIERA_START = first | prefill
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-quality --no-roofline --no-video-record $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
{
  for r in 1 2; do
    for cfg in "VG_CU_SPLIT=0" "VG_CU_SPLIT=128" "VG_CU_SPLIT=128 VG_HIERA_START=first" "VG_CU_SPLIT=96" "VG_CU_SPLIT=160" "VG_CU_SPLIT=64"; do
      echo -n "framewise $cfg: "; EXTRA="" run $cfg
    done
  done
  for cfg in "VG_CU_SPLIT=0" "VG_CU_SPLIT=128" "VG_CU_SPLIT=96"; do
    echo -n "video $cfg: "; EXTRA="--branch video" run $cfg
    echo -n "c4clip $cfg: "; EXTRA="--frames 64 --objects 8" run $cfg
    echo -n "c1 $cfg: "; EXTRA="--frames 8 --te 8 --src 512" run $cfg
  done
} > gpurun_out/cu_split_ab.log 2>&1
cat gpurun_out/cu_split_ab.log
