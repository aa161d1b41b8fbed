#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-quality --no-roofline --no-video-record $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
{
  for r in 1 2; do
    for k in 1 0; do
      echo -n "C2 framewise VG_GEMM_RR=$k: "; EXTRA="" run VG_GEMM_RR=$k
      echo -n "hiera alone VG_GEMM_RR=$k: "; VG_GEMM_RR=$k python tools/lab/hiera_kt.py 32 4 2>/dev/null | tail -1
    done
  done
  for k in 1 0; do echo -n "C4 clip VG_GEMM_RR=$k: "; EXTRA="--frames 64 --objects 8" run VG_GEMM_RR=$k; done
} > gpurun_out/rr_clip_ab.log 2>&1
cat gpurun_out/rr_clip_ab.log
