#!/bin/bash
# Runs ON the GPU box (gpurun -- 'bash tools/collect_profiles.sh TAG [all|kt|pmc]'): the bench lines and rocprofv3 passes that
# profiles/ is made from.  Outputs under gpurun_out/TAG_*; tools/prof_summary.py turns the rocpd databases into tables,
# tools/pmc_json.py the two PMC databases into per-kernel traffic files.
#   1. bench.py defaults (C2 framewise, quality + cpu_baseline included)  -> TAG_bench.json
#   2. rocprofv3 --kernel-trace --stats of bench.py --steps 3 --warmup 1  -> TAG_kt.txt (+ bench line of the traced run)
#   3. two PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only)         -> TAG_pmc_fetch.txt / TAG_pmc_write.txt / TAG_pmc_<kernel>.json
#   4. bench lines of the video branch, C1 and the Phi-3-mini composition -> TAG_bench_video.json / TAG_bench_c1.json / TAG_bench_phi3.json
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
tail -c 900 $O/${TAG}_bench.json
fi
if [ $ONLY = all ] || [ $ONLY = kt ]; then
rm -rf $O/${TAG}_kt
rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-quality > $O/${TAG}_kt.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-quality   (C2 framewise, 1 x MI355X; passes of the hot path: 1 warm-up + 3 timed + 2 instrumented (serial streams) + 3 eager decode steps; ms/step columns are totals / 5)"; echo "# bench line of the traced run:"; grep '^{"metric' $O/${TAG}_kt.log; echo;
  python $R/tools/prof_summary.py "$(db $O/${TAG}_kt)" 5; } > $O/${TAG}_kt.txt
head -14 $O/${TAG}_kt.txt | cut -c1-200
rm -rf $O/${TAG}_kt
fi
if [ $ONLY = all ] || [ $ONLY = pmc ]; then
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-quality"
for C in FETCH_SIZE WRITE_SIZE; do
  c=$(echo $C | tr A-Z a-z | cut -d_ -f1)
  rm -rf $O/${TAG}_pmc_$c
  rocprofv3 --pmc $C --kernel-trace -d $O/${TAG}_pmc_$c -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-quality > $O/${TAG}_pmc_$c.log 2>&1
  python $R/tools/prof_summary.py "$(db $O/${TAG}_pmc_$c)" 2 > $O/${TAG}_pmc_$c.txt
  grep -A8 '^PMC' $O/${TAG}_pmc_$c.txt | cut -c1-160
done
python $R/tools/pmc_json.py "$(db $O/${TAG}_pmc_fetch)" "$(db $O/${TAG}_pmc_write)" $O $TAG "C2 framewise (bench.py defaults), 1 GPU, 2 steps (warm-up + 1)" \
  "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- $CMD  (second pass: --pmc WRITE_SIZE); tools/collect_profiles.sh"
rm -rf $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write          # the databases are large; the tables are what is kept
fi
[ $ONLY = all ] || exit 0

python $R/bench.py --branch video --no-cpu-baseline > $O/${TAG}_bench_video.json 2>> $O/${TAG}_bench.err
python $R/bench.py --frames-per-gpu 8 --te 8 --src 512 --no-cpu-baseline > $O/${TAG}_bench_c1.json 2>> $O/${TAG}_bench.err
python $R/bench.py --llm phi3-mini --no-cpu-baseline > $O/${TAG}_bench_phi3.json 2>> $O/${TAG}_bench.err
cut -c1-140 $O/${TAG}_bench_video.json $O/${TAG}_bench_c1.json $O/${TAG}_bench_phi3.json
