#!/bin/bash
# call p: the executor keeps its own tile schedule / depth-bound culling after the first statistics epoch -- test + A/B on a 3000-iteration run
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_stats.py tests/test_gpu_cull.py -x -q -m gpu > gpurun_out/pytest_p.log 2>&1; tail -4 gpurun_out/pytest_p.log
for mode in stat always; do
  LITEGS_STAT_TILE_SCHEDULE=$mode LITEGS_CONV_SKIP_OPERATOR=profiles/r03_convergence_3m.json timeout -s KILL 150 python tests/convergence_3m.py --iterations 4500 --runs 1 --eval-every 5 --out gpurun_out/conv_short_$mode.md > gpurun_out/conv_short_$mode.log 2>&1
  echo "mode=$mode"; grep "^executor" gpurun_out/conv_short_$mode.log; grep "frames repeated\|steps replayed\|ms per iteration" gpurun_out/conv_short_$mode.md
done
