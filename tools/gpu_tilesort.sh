#!/bin/bash
# usage: tools/gpu_tilesort.sh -> unit + end-to-end tests of the per-tile depth sort, A/B bench and kernel stats of both modes
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 300 python -m pytest tests/test_gpu_tilesort.py tests/test_gpu_fused.py tests/test_gpu_cull.py tests/test_gpu_fullsize.py tests/test_gpu_training.py -x -q > gpurun_out/pytest_tilesort.log 2>&1; tail -15 gpurun_out/pytest_tilesort.log
for MODE in tile global; do
  LITEGS_DEPTH_ORDER=$MODE timeout -s KILL 200 python bench.py --no-cpu-baseline --no-operator-path > gpurun_out/bench_$MODE.log 2>&1; tail -1 gpurun_out/bench_$MODE.log | cut -c1-330
done
cd /tmp && export TMPDIR=/tmp
for MODE in tile global; do
  LITEGS_DEPTH_ORDER=$MODE timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$MODE -o r02 -- python $R/bench.py --steps 40 --warmup 16 --no-cpu-baseline --no-operator-path > $R/gpurun_out/rocprof_$MODE.log 2>&1
done
