#!/bin/bash
# memory-encoder kernels (strip dwconv, fused conv + LN + GELU): tests, then the video branch A/B on C2 and C4's clip
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{
  timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "memory_encoder or rope or attention_dv" 2>&1 | tail -4
  timeout 900 python -m pytest tests/test_host_sam2.py tests/test_host_vlm.py -x -q -m gpu 2>&1 | tail -3
  for f in 1 0; do
    echo "== C2 video VG_SELFATTN_FUSED=$f"
    VG_SELFATTN_FUSED=$f python bench.py --steps 4 --warmup 2 --branch video --no-cpu-baseline --no-quality --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  done
  for f in 1 0; do
    echo "== C4 clip video VG_SELFATTN_FUSED=$f"
    VG_SELFATTN_FUSED=$f python bench.py --steps 3 --warmup 2 --branch video --frames 64 --objects 8 --no-cpu-baseline --no-quality --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  done
} > gpurun_out/memenc_test.log 2>&1
tail -c 3000 gpurun_out/memenc_test.log
