#!/bin/bash
# the 256x192-tile kernel against the routes it replaces (VG_GEMM_P8=1: shape rule incl. the narrow tile; VG_GEMM_P8=2 forces the 256x256 kernel on
# every eligible shape; the r04 routes are what VG_GEMM_P8N=0 ... there is no such knob: compare with the r04 log) + the vendor library, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
echo "== shape rule (r05)"; python tools/gemm_vs_lib.py
echo "== 256x256 forced (VG_GEMM_P8=2)"; VG_GEMM_P8=2 VG_BENCH_SHAPES="qkv,proj,fc2,fc1" python tools/gemm_vs_lib.py
} > gpurun_out/${1:-p8n_ab}.log 2>&1
cat gpurun_out/${1:-p8n_ab}.log
