#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{
  timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "twoway or mask_upscale" 2>&1 | tail -8
  timeout 1500 python -m pytest tests/test_host_sam2.py tests/test_host_vlm.py tests/test_e2e_shapes.py tests/test_workload_gpu.py tests/test_dist_hip.py -x -q -m gpu 2>&1 | tail -8
  python tools/lab/tw_stage.py 32 1 2>&1 | tail -4; python tools/lab/tw_stage.py 64 8 2>&1 | tail -4
} > gpurun_out/tw_test.log 2>&1
cat gpurun_out/tw_test.log
