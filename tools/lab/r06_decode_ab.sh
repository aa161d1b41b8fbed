#!/bin/bash
# r06: the decode step with RoPE in the q|k|v GEMV + the wave-private attention (VG_DECODE_ROPE=1) against the r05 launches, same box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export BENCH_DECODE_FUSED=1
for rep in 1 2; do
  VG_DECODE_ROPE=0 python tools/bench_decode.py 3361 32
  VG_DECODE_ROPE=1 VG_DEC2_KPW=256 python tools/bench_decode.py 3361 32
  VG_DECODE_ROPE=1 VG_DEC2_KPW=128 python tools/bench_decode.py 3361 32
  VG_DECODE_ROPE=1 VG_DEC2_KPW=256 BENCH_DECODE_SYNC=0 python tools/bench_decode.py 3361 32
done
