#!/bin/bash
# the bench lines of tools/collect_profiles.sh without the traces (after a change that leaves libvgkernels.so alone: the committed kernel traces / PMC tables stay valid)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r05_c2}; mkdir -p $O; cd $R
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python bench.py --branch video --no-cpu-baseline > $O/${TAG}_bench_video.json 2>> $O/${TAG}_bench.err
python bench.py --frames 8 --te 8 --src 512 --no-cpu-baseline --no-video-record > $O/${TAG}_bench_c1.json 2>> $O/${TAG}_bench.err
python bench.py --llm phi3-mini --no-cpu-baseline --no-video-record > $O/${TAG}_bench_phi3.json 2>> $O/${TAG}_bench.err
python bench.py --frames 64 --objects 8 --no-cpu-baseline --no-quality --no-video-record > $O/${TAG}_bench_c4clip.json 2>> $O/${TAG}_bench.err
python bench.py --frames 64 --objects 8 --prefill fp8 --decode-weights fp8 --no-cpu-baseline --no-quality --no-video-record > $O/${TAG}_bench_c4clip_fp8.json 2>> $O/${TAG}_bench.err
python bench.py --frames 64 --objects 8 --branch video --no-cpu-baseline --no-quality --steps 2 > $O/${TAG}_bench_c4clip_video.json 2>> $O/${TAG}_bench.err
cut -c1-150 $O/${TAG}_bench.json $O/${TAG}_bench_video.json $O/${TAG}_bench_c1.json $O/${TAG}_bench_phi3.json $O/${TAG}_bench_c4clip.json $O/${TAG}_bench_c4clip_fp8.json $O/${TAG}_bench_c4clip_video.json
