#!/bin/bash
# usage: tools/gpu_prof.sh TAG -> kernel profiles at 3M and 500k only (no tests)
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof$TAG -o r01 -- python $R/bench.py --steps 40 --warmup 16 --no-cpu-baseline > $R/gpurun_out/rocprof$TAG.log 2>&1
timeout -s KILL 90 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof${TAG}_500k -o r01 -- python $R/bench.py --config 500k_1080p --steps 40 --warmup 16 --no-cpu-baseline > $R/gpurun_out/rocprof${TAG}_500k.log 2>&1
