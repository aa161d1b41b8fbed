#!/bin/bash
# window gather on the phase-split kernel: tests + Hiera / C2 A/B (VG_GEMM_P8_WINDOW)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{
  timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
  timeout 900 python -m pytest tests/test_host_sam2.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3
  for r in 1 2; do
    for f in 1 0; do
      echo "== C2 VG_GEMM_P8_WINDOW=$f"
      VG_GEMM_P8_WINDOW=$f python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-quality --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
    done
  done
} > gpurun_out/p8win_test.log 2>&1
tail -c 2500 gpurun_out/p8win_test.log
