#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{
  for v in "" rr_nostore; do
    echo "== variant '$v'"
    if [ -z "$v" ]; then timeout 300 python tools/lab/rr_bench.py 2>&1 | grep "^RR"; else VG_KERNELS_SO=build/variants/libvg_$v.so timeout 300 python tools/lab/rr_bench.py 2>&1 | grep "^RR"; fi
  done
} > gpurun_out/rr_abl.log 2>&1
cat gpurun_out/rr_abl.log
