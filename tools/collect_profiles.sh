#!/bin/bash
# Runs ON the GPU box (gpurun -- 'bash tools/collect_profiles.sh TAG [all|kt|pmc|mfma|bench]'): the bench lines and rocprofv3 passes that
# profiles/ is made from.  Outputs under gpurun_out/TAG_*.
#   bench: bench.py defaults (C2 framewise, quality + cpu_baseline included), run AFTER the traces -> TAG_bench.json
#   kt:    TWO kernel traces of `bench.py --steps 3 --warmup 1 --no-roofline ...` (4 passes each, nothing else in the process):
#          the timed stream configuration (Hiera on its side stream)                     -> TAG_kt_overlapped.txt / .json
#          VG_HIERA_START=serial VG_TOWERS_OVERLAP=0 (the instrumented pass's config)    -> TAG_kt_serial.txt / .json
#   mfma:  --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass in both configurations    -> mfma_busy_frac inside TAG_kt_*.json
#   pmc:   two PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only)                    -> TAG_pmc_fetch.txt / TAG_pmc_write.txt / TAG_pmc_<kernel>.json
#   all:   everything above + bench lines of the video branch, C1, Phi-3-mini and C4's clip on one GPU -> TAG_bench_{video,c1,phi3,c4clip,c4clip_fp8,c4clip_video}.json
set -u
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name '*.db' | head -1; }
has() { [ $ONLY = all ] || [ $ONLY = $1 ]; }

ONLY=${2:-all}
BENCH="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-quality --no-roofline --no-video-record"
PASSES=4
if has kt || has mfma; then
for MODE in overlapped serial; do
  if [ $MODE = serial ]; then export VG_HIERA_START=serial VG_TOWERS_OVERLAP=0; ENVS="VG_HIERA_START=serial VG_TOWERS_OVERLAP=0 "; else unset VG_HIERA_START VG_TOWERS_OVERLAP; ENVS=""; fi
  rm -rf $O/${TAG}_kt_$MODE $O/${TAG}_mfma_$MODE
  rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt_$MODE -o kt -- python $R/$BENCH > $O/${TAG}_kt_$MODE.log 2>&1
  MF=""
  if has mfma; then
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/${TAG}_mfma_$MODE -o pmc -- python $R/$BENCH > $O/${TAG}_mfma_$MODE.log 2>&1
    MF="$(db $O/${TAG}_mfma_$MODE)"
  fi
  CMD="${ENVS}rocprofv3 --kernel-trace --stats -- python $BENCH"
  { echo "# $CMD   (C2 framewise, 1 x MI355X; $PASSES passes of the hot path, all in the '$MODE' stream configuration: 1 warm-up + 3 timed; per-pass columns = totals / $PASSES)"
    echo "# bench line of the traced run:"; grep '^{"metric' $O/${TAG}_kt_$MODE.log; echo
    python $R/tools/prof_summary.py "$(db $O/${TAG}_kt_$MODE)" $PASSES; } > $O/${TAG}_kt_$MODE.txt
  [ $MODE = serial ] && python $R/tools/decode_timeline.py "$(db $O/${TAG}_kt_$MODE)" > $O/${TAG}_decode_timeline.txt 2>&1
  (cd $R/tools && python kt_json.py "$(db $O/${TAG}_kt_$MODE)" $O/${TAG}_kt_$MODE.json $PASSES $MODE "$CMD  (+ a --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace pass of the same command for mfma_busy_frac); tools/collect_profiles.sh" $MF)
  rm -rf $O/${TAG}_kt_$MODE $O/${TAG}_mfma_$MODE
done
unset VG_HIERA_START VG_TOWERS_OVERLAP
fi
if has bench; then
# after the traces: the line quotes profiles/${TAG}_kt_*.json (frac_trace_*, mfma_busy_frac_*), so the traces of THIS call (same box, same build) go
# into the box's copy of profiles/ first
for MODE in overlapped serial; do [ -f $O/${TAG}_kt_$MODE.json ] && cp $O/${TAG}_kt_$MODE.json $R/profiles/; done
python $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
tail -c 900 $O/${TAG}_bench.json
fi
if has pmc; then
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-quality --no-video-record"
for C in FETCH_SIZE WRITE_SIZE; do
  c=$(echo $C | tr A-Z a-z | cut -d_ -f1)
  rm -rf $O/${TAG}_pmc_$c
  rocprofv3 --pmc $C --kernel-trace -d $O/${TAG}_pmc_$c -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-quality --no-video-record > $O/${TAG}_pmc_$c.log 2>&1
  python $R/tools/prof_summary.py "$(db $O/${TAG}_pmc_$c)" 2 > $O/${TAG}_pmc_$c.txt
  grep -A8 '^PMC' $O/${TAG}_pmc_$c.txt | cut -c1-160
done
python $R/tools/pmc_json.py "$(db $O/${TAG}_pmc_fetch)" "$(db $O/${TAG}_pmc_write)" $O $TAG "C2 framewise (bench.py defaults), 1 GPU, 2 steps (warm-up + 1)" \
  "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- $CMD  (second pass: --pmc WRITE_SIZE); tools/collect_profiles.sh"
rm -rf $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write          # the databases are large; the tables are what is kept
fi
[ $ONLY = all ] || exit 0

python $R/bench.py --branch video --no-cpu-baseline > $O/${TAG}_bench_video.json 2>> $O/${TAG}_bench.err
python $R/bench.py --frames 8 --te 8 --src 512 --no-cpu-baseline --no-video-record > $O/${TAG}_bench_c1.json 2>> $O/${TAG}_bench.err
python $R/bench.py --llm phi3-mini --no-cpu-baseline --no-video-record > $O/${TAG}_bench_phi3.json 2>> $O/${TAG}_bench.err
# BASELINE config C4's clip on ONE GPU (64 frames, 8 [SEG] objects = 512 mask-decoder instances), bf16 and the fp8 LLM path, framewise and video branch
python $R/bench.py --frames 64 --objects 8 --no-cpu-baseline --no-quality --no-video-record > $O/${TAG}_bench_c4clip.json 2>> $O/${TAG}_bench.err
python $R/bench.py --frames 64 --objects 8 --prefill fp8 --decode-weights fp8 --no-cpu-baseline --no-quality --no-video-record > $O/${TAG}_bench_c4clip_fp8.json 2>> $O/${TAG}_bench.err
python $R/bench.py --frames 64 --objects 8 --branch video --no-cpu-baseline --no-quality --steps 2 > $O/${TAG}_bench_c4clip_video.json 2>> $O/${TAG}_bench.err
cut -c1-140 $O/${TAG}_bench_video.json $O/${TAG}_bench_c1.json $O/${TAG}_bench_phi3.json $O/${TAG}_bench_c4clip.json $O/${TAG}_bench_c4clip_fp8.json $O/${TAG}_bench_c4clip_video.json
# r05: the video branch (kernel trace of the C2 clip and of C4's clip on it, one tracked frame kernel by kernel), the mask decoder at C4's clip size, GEMMs against the vendor library
P=${TAG%_c2}
VIDEO_ARGS="" bash $R/tools/lab/video_kt.sh ${TAG}_video_kt > /dev/null 2>&1
PASSES=3 VIDEO_ARGS="--frames 64 --objects 8 --steps 2" bash $R/tools/lab/video_kt.sh ${P}_c4clip_video_kt > /dev/null 2>&1
bash $R/tools/lab/video_frame_seq.sh ${P}_video_frame_seq 1 > /dev/null 2>&1
bash $R/tools/lab/maskdec_kt.sh ${P}_maskdec_kt > /dev/null 2>&1
(cd $R && python tools/gemm_vs_lib.py > $O/${P}_gemm_vs_lib.log 2>&1)
ls $O | grep "^${P}" | tr '\n' ' '
