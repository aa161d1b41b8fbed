#!/bin/bash
# kernel trace of the C2 clip on the video branch (serial stream configuration): where the propagation's time goes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export VG_HIERA_START=serial VG_TOWERS_OVERLAP=0
rm -rf $O/video_kt
rocprofv3 --kernel-trace --stats -d $O/video_kt -o kt -- python $R/bench.py --branch video --steps 3 --warmup 1 --no-cpu-baseline --no-quality --no-roofline ${VIDEO_ARGS:-} > $O/video_kt.log 2>&1
{ grep '^{"metric' $O/video_kt.log | cut -c1-200; python $R/tools/prof_summary.py "$(find $O/video_kt -name '*.db' | head -1)" ${PASSES:-4} 80; } > $O/${1:-video_kt}.txt
rm -rf $O/video_kt
head -70 $O/${1:-video_kt}.txt | cut -c1-190
