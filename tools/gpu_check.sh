#!/bin/bash
# usage (on the GPU box, from repo root): tools/gpu_check.sh TAG  -> tests, bench, kernel profile under gpurun_out/
# every step is bounded with SIGKILL: a kernel that spins forever must not eat the GPU budget
TAG=$1
timeout -s KILL 240 python -m pytest tests -m gpu -x -q > gpurun_out/pytest$TAG.log 2>&1; grep -E "passed|failed" gpurun_out/pytest$TAG.log | tail -1
timeout -s KILL 150 python bench.py > gpurun_out/bench$TAG.log 2>&1; tail -1 gpurun_out/bench$TAG.log | cut -c1-330
cd /tmp && export TMPDIR=/tmp && timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof$TAG -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 16 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof$TAG.log 2>&1
