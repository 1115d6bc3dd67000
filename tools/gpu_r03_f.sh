#!/bin/bash
# round 3, call f: kernel timeline of both states with the new binning / blend kernels; SQ counters fresh + trained state
TAG=${1:-f}
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o r03 -- python $R/bench.py --no-cpu-baseline --no-operator-path --no-pmc > $R/gpurun_out/rocprof_$TAG.log 2>&1
tail -1 $R/gpurun_out/rocprof_$TAG.log | cut -c1-300
cd $R
T=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python tools/profile_r03.py $T > gpurun_out/step_timeline_$TAG.md 2> gpurun_out/step_timeline_$TAG.err; head -48 gpurun_out/step_timeline_$TAG.md; tail -3 gpurun_out/step_timeline_$TAG.err
S=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/kernel_stats_$TAG.csv
rm -rf gpurun_out/prof_$TAG
timeout -s KILL 200 python tools/soaked_probe.py save /tmp/soaked.npz 1000 > gpurun_out/soaked_save_$TAG.log 2>&1; tail -1 gpurun_out/soaked_save_$TAG.log
cd /tmp
for STATE in fresh trained; do
  if [ $STATE = fresh ]; then CMD="python $R/bench.py --pmc-child --steps 8 --warmup 0"; else CMD="python $R/tools/soaked_probe.py run /tmp/soaked.npz 8"; fi
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_sq_${STATE}_$TAG -o r03 -- $CMD > $R/gpurun_out/pmc_sq_${STATE}_$TAG.log 2>&1
  C=$(find $R/gpurun_out/pmc_sq_${STATE}_$TAG -name "*counter_collection.csv" | head -1)
  python $R/tools/sq_summary.py $C "$STATE cloud" > $R/gpurun_out/sq_counters_${STATE}_$TAG.md 2>> $R/gpurun_out/step_timeline_$TAG.err; head -14 $R/gpurun_out/sq_counters_${STATE}_$TAG.md
  rm -rf $R/gpurun_out/pmc_sq_${STATE}_$TAG
done
