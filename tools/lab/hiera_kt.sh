#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/hkt
rocprofv3 --kernel-trace --stats -d $O/hkt -o kt -- python $R/tools/lab/hiera_kt.py 32 4 > $O/hkt.log 2>&1
{ tail -2 $O/hkt.log; python $R/tools/prof_summary.py "$(find $O/hkt -name '*.db' | head -1)" 4 60; } > $O/${1:-hiera_kt}.txt
rm -rf $O/hkt
head -70 $O/${1:-hiera_kt}.txt | cut -c1-230
