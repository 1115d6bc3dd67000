#!/bin/bash
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o r02 -- python $R/tools/cull_ab.py > $R/gpurun_out/rocprof_$TAG.log 2>&1
grep step $R/gpurun_out/rocprof_$TAG.log
