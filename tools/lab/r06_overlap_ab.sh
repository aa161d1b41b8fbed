#!/bin/bash
# where the Hiera pass (MFMA-bound) best overlaps the text side: with the towers + prefill (MFMA-bound too; 'first') or with the HBM-bound decode loop ('prefill'),
# and whether stream priorities help the decode's short kernels get CUs between the GEMM workgroups
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-quality --no-roofline --no-video-record --no-config-records"
run() { echo "== $*"; env "$@" $B 2>/dev/null | python -c 'import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])'; }
python -c "import torch; print(torch.cuda.Stream.priority_range())"
run VG_HIERA_START=first
run VG_HIERA_START=prefill
run VG_HIERA_START=first VG_TEXT_PRIO=-1
run VG_HIERA_START=prefill VG_TEXT_PRIO=-1
run VG_HIERA_START=prefill VG_HIERA_PRIO=1
run VG_HIERA_START=first
