#!/bin/bash
# kernel trace of the isolated decode loop (tools/bench_decode.py at C2's cache length) -> per-role durations (tools/decode_timeline.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/deckt
BENCH_DECODE_FUSED=1 BENCH_DECODE_SYNC=${SYNC:-0} rocprofv3 --kernel-trace -d $O/deckt -o kt -- python $R/tools/bench_decode.py 3361 32 > $O/deckt.log 2>&1
{ grep fused $O/deckt.log; python $R/tools/decode_timeline.py "$(find $O/deckt -name '*.db' | head -1)"; } > $O/${1:-decode_timeline}.txt 2>&1
rm -rf $O/deckt
cat $O/${1:-decode_timeline}.txt | cut -c1-220
