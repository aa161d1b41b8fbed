#!/bin/bash
# r04: the phase-split 256x256 GEMM (VG_GEMM_P8=1, default) against the lock-step kernel (VG_GEMM_P8=0) and the vendor library, same box.
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
S="c2 llm,iv2,clip,hiera s4,square"
{
  timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -5
  for P in 1 0 1 0; do
    echo "== VG_GEMM_P8=$P"
    VG_GEMM_P8=$P VG_BENCH_SHAPES="$S" timeout 600 python tools/gemm_vs_lib.py
  done
} > gpurun_out/p8_ab.log 2>&1
tail -70 gpurun_out/p8_ab.log
