#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/mdkt
rocprofv3 --kernel-trace --stats -d $O/mdkt -o kt -- python $R/tools/lab/maskdec_kt.py > $O/mdkt.log 2>&1
{ tail -2 $O/mdkt.log; python $R/tools/prof_summary.py "$(find $O/mdkt -name '*.db' | head -1)" 6; } > $O/${1:-maskdec_kt}.txt
rm -rf $O/mdkt
head -40 $O/${1:-maskdec_kt}.txt | cut -c1-200
