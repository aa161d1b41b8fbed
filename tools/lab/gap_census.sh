#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/gapkt
rocprofv3 --kernel-trace -d $O/gapkt -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-quality --no-roofline --no-video-record $BENCH_ARGS > $O/gapkt.log 2>&1
{ grep '^{"metric' $O/gapkt.log | cut -c1-200; python $R/tools/gap_census.py "$(find $O/gapkt -name '*.db' | head -1)" 15 40; } > $O/${1:-gap_census}.txt 2>&1
rm -rf $O/gapkt
cat $O/${1:-gap_census}.txt | cut -c1-200
