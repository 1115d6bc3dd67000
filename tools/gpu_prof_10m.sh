#!/bin/bash
# usage: tools/gpu_prof_10m.sh -> kernel-trace stats of the 10 M @1600x1200 configuration (BASELINE configs[4]) under gpurun_out/prof_10m
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_10m -o r02 -- python $R/bench.py --config 10m_1600x1200 --frames 4 --steps 40 --warmup 16 --no-cpu-baseline --no-operator-path > $R/gpurun_out/rocprof_10m.log 2>&1
tail -1 $R/gpurun_out/rocprof_10m.log | cut -c1-200
