#!/bin/bash
# kernel-by-kernel trace of one tracked frame of the SAM2-L propagation (tools/lab/video_frame_seq.py) -> gpurun_out/$1.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/vfs
rocprofv3 --kernel-trace -d $O/vfs -o kt -- python $R/tools/lab/video_frame_seq.py run ${2:-1} > $O/vfs.log 2>&1
python $R/tools/lab/video_frame_seq.py show "$(find $O/vfs -name '*.db' | head -1)" > $O/${1:-video_frame_seq}.txt 2>&1
rm -rf $O/vfs
tail -40 $O/${1:-video_frame_seq}.txt
