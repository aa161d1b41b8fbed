#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{
  timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm_rr" 2>&1 | tail -3
  for k in 1 0; do VG_GEMM_RR=$k timeout 300 python tools/lab/rr_bench.py 2>&1 | grep "^RR"; done
} > gpurun_out/rr_test.log 2>&1
cat gpurun_out/rr_test.log
