#!/bin/bash
# r04 state of the world: gpu tests, C2 bench line with stages, C2 video branch
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{
  timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
  echo "== C2 bench"
  python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r04_c2.json; cat gpurun_out/r04_c2.json
  echo "== C2 video"
  python bench.py --steps 4 --warmup 2 --branch video --no-cpu-baseline --no-quality --no-roofline 2>/dev/null | tail -1
} > gpurun_out/r04_base.log 2>&1
tail -c 6000 gpurun_out/r04_base.log
