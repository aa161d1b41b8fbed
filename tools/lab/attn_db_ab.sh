#!/bin/bash
# double-buffered K / V tiles in the key-split dv kernel (VG_ATTN_DB): kernel parity, then dv_bench on the default build and on the -DVG_ATTN_DB=0 variant
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{
  timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -5
  for r in 1 2; do
    echo "== default (DB)"; timeout 300 python tools/lab/dv_bench.py 2>/dev/null
    echo "== noskew"; VG_KERNELS_SO=build/variants/libvg_noskew.so timeout 300 python tools/lab/dv_bench.py 2>/dev/null
    echo "== nodb"; VG_KERNELS_SO=build/variants/libvg_nodb.so timeout 300 python tools/lab/dv_bench.py 2>/dev/null
  done
} > gpurun_out/attn_db_ab.log 2>&1
tail -c 4000 gpurun_out/attn_db_ab.log
