#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/tw_kt
rocprofv3 --kernel-trace --stats -d $O/tw_kt -o kt -- python $R/tools/lab/tw_stage.py 64 8 > $O/tw_kt.log 2>&1
python $R/tools/prof_summary.py "$(find $O/tw_kt -name '*.db' | head -1)" 16 | head -60 > $O/tw_kt.txt
rm -rf $O/tw_kt
cat $O/tw_kt.txt
