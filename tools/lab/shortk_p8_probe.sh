#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{ for k in 1 3 2; do VG_GEMM_P8=$k timeout 300 python tools/lab/shortk_p8_probe.py 2>&1 | grep "^P8"; done; } > gpurun_out/shortk_p8_probe.log 2>&1
cat gpurun_out/shortk_p8_probe.log
