#!/bin/bash
# kernel table of the C4 clip (64 frames x 8 objects = 512 mask-decoder instances) with the fused / unfused two-way image side
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  rm -rf $O/tw_kt_$f
  VG_TWOWAY_FUSED=$f rocprofv3 --kernel-trace --stats -d $O/tw_kt_$f -o kt -- python $R/bench.py --steps 1 --warmup 1 --frames-per-gpu 64 --objects 8 --no-cpu-baseline --no-quality --no-roofline > $O/tw_kt_$f.log 2>&1
  python $R/tools/prof_summary.py "$(find $O/tw_kt_$f -name '*.db' | head -1)" 2 | head -45 > $O/tw_kt_$f.txt
  rm -rf $O/tw_kt_$f
done
paste -d'\n' /dev/null; echo "=== fused"; cat $O/tw_kt_1.txt; echo "=== unfused"; cat $O/tw_kt_0.txt
