#!/bin/bash
# usage: tools/gpu_final.sh TAG -> what the driver runs at round end (GPU tests, smoke, default bench) plus the round's kernel profiles
TAG=$1
R=$GRAFT_REPO_ROOT
timeout -s KILL 240 python -m pytest tests -m gpu -x -q > gpurun_out/pytest$TAG.log 2>&1; grep -E "passed|failed" gpurun_out/pytest$TAG.log | tail -1
timeout -s KILL 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke$TAG.log 2>&1; tail -1 gpurun_out/smoke$TAG.log
timeout -s KILL 200 python bench.py > gpurun_out/bench$TAG.log 2>&1; tail -1 gpurun_out/bench$TAG.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof$TAG -o r01 -- python $R/bench.py --steps 40 --warmup 16 --no-cpu-baseline > $R/gpurun_out/rocprof$TAG.log 2>&1
timeout -s KILL 90 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof${TAG}_500k -o r01 -- python $R/bench.py --config 500k_1080p --steps 40 --warmup 16 --no-cpu-baseline > $R/gpurun_out/rocprof${TAG}_500k.log 2>&1
