#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{
  H="gateup,down,o,8k,4k,s4fc1,iv2fc1"
  for r in 1 2 3; do
    TAG=new SHAPES=$H python tools/lab/gemm_time.py
    for v in build/variants/libvg_*.so; do TAG=old SHAPES=$H VG_KERNELS_SO=$PWD/$v python tools/lab/gemm_time.py; done
    TAG=w128x8 VG_GEMM_P8=0 SHAPES=$H python tools/lab/gemm_time.py
  done
} > gpurun_out/p8_variants.log 2>&1
cat gpurun_out/p8_variants.log
