#!/bin/bash
# C2 clip with / without the r06 decode launches and the run-ahead loop, same box: ms per step + the serial decode stage
cd "$(dirname "$0")/../.."
run() { env "$@" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-quality --no-video-record 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'decode stage', d['stages']['decode']['ms'], 'ms/token', d['roofline_decode']['ms_per_token'])"; }
for rep in 1 2; do
  echo -n "rope=0 ahead=0: "; run VG_DECODE_ROPE=0 VG_DECODE_AHEAD=0
  echo -n "rope=1 ahead=0: "; run VG_DECODE_ROPE=1 VG_DECODE_AHEAD=0
  echo -n "rope=1 ahead=1: "; run VG_DECODE_ROPE=1 VG_DECODE_AHEAD=1
  echo -n "rope=0 ahead=1: "; run VG_DECODE_ROPE=0 VG_DECODE_AHEAD=1
done
