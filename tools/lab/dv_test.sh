#!/bin/bash
# low-rank memory cross-attention (vg_attention_dv): kernel + fixture parity, then the video branch A/B on C2 and on C4's clip
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{
  timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -5
  timeout 900 python -m pytest tests/test_host_sam2.py tests/test_oracle_e2e.py tests/test_host_vlm.py -x -q -m gpu 2>&1 | tail -5
  for r in 1 2; do
    for lr in 1 0; do
      echo "== C2 video VG_MEMATTN_LOWRANK=$lr"
      VG_MEMATTN_LOWRANK=$lr python bench.py --steps 4 --warmup 2 --branch video --no-cpu-baseline --no-quality --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
    done
  done
  for lr in 1 0; do
    echo "== C4 clip video VG_MEMATTN_LOWRANK=$lr"
    VG_MEMATTN_LOWRANK=$lr python bench.py --steps 3 --warmup 2 --branch video --frames 64 --objects 8 --no-cpu-baseline --no-quality --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  done
} > gpurun_out/dv_test.log 2>&1
tail -c 5000 gpurun_out/dv_test.log
