#!/bin/bash
# same-box A/B of the Hiera start point: first (default: beside towers + prefill) vs prefill (enqueued when the decode loop starts)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2; do
  for v in first prefill; do
    VG_HIERA_START=$v python bench.py --no-cpu-baseline --no-quality --no-roofline --steps 5 > gpurun_out/sched_ab_$v.json 2> gpurun_out/sched_ab_$v.err
    python -c "
import json; r=json.load(open('gpurun_out/sched_ab_$v.json')); print('VG_HIERA_START=$v  C2 %.2f ms  video %.2f ms  graph %s' % (r['ms_per_step'], r['video_branch']['ms_per_step'], r['video_branch'].get('propagation_graph')))"
  done
done
