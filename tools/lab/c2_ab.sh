#!/bin/bash
# end-to-end C2 A/B of a routing knob, same box
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{
  timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
  for r in 1 2; do
    for kb in 2304 1152; do
      echo "== VG_W128_MINKB=$kb"
      VG_W128_MINKB=$kb python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-quality --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('stages'))"
    done
  done
  echo "== VG_GEMM_P8=0"
  VG_GEMM_P8=0 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-quality --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('stages'))"
} > gpurun_out/c2_ab.log 2>&1
cat gpurun_out/c2_ab.log
