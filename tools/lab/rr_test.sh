#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{
  timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm_rr or gemm_ln or gemm_window or mlp_rows" 2>&1 | tail -12
  timeout 900 python -m pytest tests/test_host_sam2.py tests/test_oracle_sam2.py -x -q -m gpu 2>&1 | tail -4
  for r in 1 2; do
    for k in 1 0; do
      echo -n "hiera alone VG_GEMM_RR=$k: "; VG_GEMM_RR=$k python tools/lab/hiera_kt.py 32 4 2>/dev/null | tail -1
    done
  done
} > gpurun_out/rr_test.log 2>&1
cat gpurun_out/rr_test.log
