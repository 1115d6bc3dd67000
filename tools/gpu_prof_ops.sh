#!/bin/bash
# usage: tools/gpu_prof_ops.sh -> kernel-trace stats of the operator path (bench.py --operator-path) under gpurun_out/prof_ops
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ops -o r02 -- python $R/bench.py --operator-path --no-cpu-baseline --steps 32 --warmup 8 > $R/gpurun_out/rocprof_ops.log 2>&1
tail -1 $R/gpurun_out/rocprof_ops.log | cut -c1-200
