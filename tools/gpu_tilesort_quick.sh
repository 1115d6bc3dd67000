#!/bin/bash
# usage: tools/gpu_tilesort_quick.sh -> tile-sort unit tests + a kernel trace of the tile mode (duration of tile_depth_sort_kernel and the emission)
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 200 python -m pytest tests/test_gpu_tilesort.py -x -q > gpurun_out/pytest_tsq.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_tsq.log | tail -1; grep -E "^E  " gpurun_out/pytest_tsq.log | head -4
cd /tmp && export TMPDIR=/tmp
LITEGS_DEPTH_ORDER=tile timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tile -o r02 -- python $R/bench.py --steps 40 --warmup 16 --no-cpu-baseline --no-operator-path > $R/gpurun_out/rocprof_tile.log 2>&1
grep -E "tile_depth_sort|dup_small|dup_big" $R/gpurun_out/prof_tile/r02_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,200-
grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/rocprof_tile.log | head -1
