#!/bin/bash
# usage: tools/gpu_prof2.sh TAG [pmc] -> kernel-trace stats of the default bench (3M) under gpurun_out/prof_TAG; with "pmc": SQ counter passes too
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o r02 -- python $R/bench.py --steps 40 --warmup 16 --no-cpu-baseline > $R/gpurun_out/rocprof_$TAG.log 2>&1
tail -1 $R/gpurun_out/rocprof_$TAG.log | cut -c1-600
if [ "$2" == "pmc" ]; then
  rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/sq_counters_available.txt
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_sqA_$TAG -o r02 -- python $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline > $R/gpurun_out/pmc_sqA_$TAG.log 2>&1
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_LEVEL_WAVES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pmc_sqB_$TAG -o r02 -- python $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline > $R/gpurun_out/pmc_sqB_$TAG.log 2>&1
fi
