#!/bin/bash
# Runs ON the GPU box (gpurun -- 'bash tools/collect_profiles.sh TAG'): the bench lines and rocprofv3 passes that
# profiles/ is made from.  Outputs under gpurun_out/TAG_*; tools/prof_summary.py turns the rocpd databases into tables.
#   1. bench.py defaults (C1 framewise, cpu_baseline included)           -> TAG_bench.json
#   2. rocprofv3 --kernel-trace --stats of bench.py --steps 3 --warmup 1 -> TAG_kt.txt (+ bench line of the traced run)
#   3. two PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only)        -> TAG_pmc_fetch.txt / TAG_pmc_write.txt
#   4. bench lines of the video branch and of the Phi-3-mini composition -> TAG_bench_video.json / TAG_bench_phi3.json
set -u
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name '*.db' | head -1; }

ONLY=${2:-all}
if [ $ONLY = all ]; then
python $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
tail -c 600 $O/${TAG}_bench.json

rm -rf $O/${TAG}_kt
rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${TAG}_kt.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline"; echo "# bench line of the traced run:"; grep '^{"metric' $O/${TAG}_kt.log; echo;
  python $R/tools/prof_summary.py "$(db $O/${TAG}_kt)" 5; } > $O/${TAG}_kt.txt
head -12 $O/${TAG}_kt.txt | cut -c1-200
rm -rf $O/${TAG}_kt
fi

for C in FETCH_SIZE WRITE_SIZE; do
  c=$(echo $C | tr A-Z a-z | cut -d_ -f1)
  rm -rf $O/${TAG}_pmc_$c
  rocprofv3 --pmc $C --kernel-trace -d $O/${TAG}_pmc_$c -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/${TAG}_pmc_$c.log 2>&1
  python $R/tools/prof_summary.py "$(db $O/${TAG}_pmc_$c)" 2 > $O/${TAG}_pmc_$c.txt
  grep -A8 '^PMC' $O/${TAG}_pmc_$c.txt | cut -c1-160
  rm -rf $O/${TAG}_pmc_$c          # the databases are large; the table is what is kept
done
[ $ONLY = all ] || exit 0

python $R/bench.py --branch video --no-cpu-baseline > $O/${TAG}_bench_video.json 2>> $O/${TAG}_bench.err
python $R/bench.py --llm phi3-mini --no-cpu-baseline > $O/${TAG}_bench_phi3.json 2>> $O/${TAG}_bench.err
cut -c1-140 $O/${TAG}_bench_video.json $O/${TAG}_bench_phi3.json
