#!/bin/bash
# the decode loop on n CUs (n / 8 per XCD) and the Hiera pass on the other 256 - n, started after the prefill (vg_stream_create_cu_range) against the default
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-quality --no-roofline --no-video-record --no-config-records"
run() { echo "== $*"; env "$@" timeout 300 $B 2>/tmp/err.log | python -c 'import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])' || tail -5 /tmp/err.log; }
run VG_HIERA_START=first
for n in 32 64 96 128; do run VG_HIERA_START=prefill VG_CU_SPLIT=$n; done
