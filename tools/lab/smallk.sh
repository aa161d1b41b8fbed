#!/bin/bash
# lab: ablations of the small-K GEMM kernels on the real C2 shapes (VG_KERNELS_SO = the -DVG_LAB build)
export VG_KERNELS_SO=$PWD/tools/lab/libvg_lab.so VG_BENCH_GEMM_ONLY=1 VG_BENCH_SHAPES="${SHAPES:-c2 s}"
for ring in 0 1; do for ab in ${ABL:-0 1 2 4 6}; do
  echo "== ring=$ring ablate=$ab (1 no stores, 2 no loads, 4 no epilogue)"
  VG_GEMM_RING64=$ring VG_GEMM_ABLATE=$ab python tools/bench_gemm.py 2>&1 | grep "^gemm"
done; done
