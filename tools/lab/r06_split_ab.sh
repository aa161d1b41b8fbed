#!/bin/bash
# the captured decode step as ONE graph or as TWO (step head + n layers | the rest): the chip idles while the host writes a replay's ~170 packets
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-quality --no-roofline --no-video-record --no-config-records"
run() { echo "== $*"; env "$@" BENCH_DECODE_FUSED=1 python tools/bench_decode.py 3361 32 2>/dev/null | tail -1; env "$@" $B 2>/dev/null | python -c 'import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print("C2", d["ms_per_step"], d["value"])'; }
run VG_DECODE_SPLIT=0
run VG_DECODE_SPLIT=2
run VG_DECODE_SPLIT=4
run VG_DECODE_SPLIT=8
run VG_DECODE_SPLIT=0
