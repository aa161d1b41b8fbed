#!/bin/bash
# per-shape GEMM table of the C2 step + ours-vs-library on every C2 shape + the new tests
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{
  timeout 1200 python -m pytest tests/test_dist_hip.py tests/test_workload_gpu.py -x -q -m gpu 2>&1 | tail -8
  echo "== shapes"
  VG_BENCH_GEMM_SHAPES=1 VG_BENCH_ATTN_SHAPES=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-quality 2>&1 >/dev/null | grep -E "^gemm|^attn"
  echo "== vs lib"
  python tools/gemm_vs_lib.py 2>/dev/null
} > gpurun_out/r04_shapes.log 2>&1
tail -c 8000 gpurun_out/r04_shapes.log
