#!/bin/bash
# stream-K: correctness tests, ours-vs-library table with and without it, C2 end to end A/B
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{
  timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -8
  for sk in 1 0; do
    echo "== VG_GEMM_SK=$sk"
    VG_GEMM_SK=$sk VG_BENCH_ROUNDS=2 python tools/gemm_vs_lib.py 2>/dev/null | grep -v "hiera s[12]"
  done
  for r in 1 2; do
    for sk in 1 0; do
      echo "== C2 VG_GEMM_SK=$sk"
      VG_GEMM_SK=$sk python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-quality --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
    done
  done
} > gpurun_out/sk_test.log 2>&1
tail -c 7000 gpurun_out/sk_test.log
