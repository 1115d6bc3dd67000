#!/bin/bash
# round 3, second GPU call: blend-backward experiments, both-state kernel timeline + SQ counters, fp16 distance report, DP control flow
TAG=${1:-b}
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 300 python tools/bwd_ab.py > gpurun_out/bwd_ab_$TAG.log 2>&1; tail -14 gpurun_out/bwd_ab_$TAG.log
timeout -s KILL 400 python tools/fp16_distance.py > gpurun_out/fp16_distance_$TAG.md 2> gpurun_out/fp16_distance_$TAG.err; tail -22 gpurun_out/fp16_distance_$TAG.md; tail -3 gpurun_out/fp16_distance_$TAG.err
bash tools/gpu_dp_sanity.sh
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o r03 -- python $R/bench.py --no-cpu-baseline --no-operator-path --no-pmc > $R/gpurun_out/rocprof_$TAG.log 2>&1
tail -1 $R/gpurun_out/rocprof_$TAG.log | cut -c1-400
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_sq_$TAG -o r03 -- python $R/bench.py --no-cpu-baseline --no-operator-path --no-pmc > $R/gpurun_out/pmc_sq_$TAG.log 2>&1
cd $R
T=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1); C=$(find gpurun_out/pmc_sq_$TAG -name "*counter_collection.csv" | head -1)
python tools/profile_r03.py $T $C > gpurun_out/step_timeline_$TAG.md 2> gpurun_out/step_timeline_$TAG.err; head -50 gpurun_out/step_timeline_$TAG.md; tail -3 gpurun_out/step_timeline_$TAG.err
S=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/kernel_stats_$TAG.csv
# the raw traces are large: keep only the summaries
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_sq_$TAG
