#!/bin/bash
# usage: tools/gpu_r03_final.sh TAG -> what the driver runs at round end (GPU tests, smoke, default bench) + the round's profiles
TAG=${1:-final}
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/flip_counts.jsonl
LITEGS_COLLECT_FLIPS=1 timeout -s KILL 900 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/pytest_$TAG.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_$TAG.log | tail -1; grep -E "FAILED" gpurun_out/pytest_$TAG.log | head
timeout -s KILL 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; tail -1 gpurun_out/smoke_$TAG.log
timeout -s KILL 400 python bench.py > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-600
timeout -s KILL 300 python bench.py --config 10m_1600x1200 --frames 4 --no-cpu-baseline --no-operator-path --no-pmc > gpurun_out/bench_10m_$TAG.log 2>&1; tail -1 gpurun_out/bench_10m_$TAG.log | cut -c1-300
timeout -s KILL 200 python bench.py --config 500k_1080p --no-cpu-baseline --no-operator-path --no-pmc --soak-steps 0 > gpurun_out/bench_500k_$TAG.log 2>&1; tail -1 gpurun_out/bench_500k_$TAG.log | cut -c1-300
LITEGS_BENCH_ONE_GPU=1 timeout -s KILL 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 4 --config 500k_1080p > gpurun_out/bench_dp2_$TAG.log 2>&1; grep '^{' gpurun_out/bench_dp2_$TAG.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o r03 -- python $R/bench.py --no-cpu-baseline --no-operator-path --no-pmc > $R/gpurun_out/rocprof_$TAG.log 2>&1
cd $R
T=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python tools/profile_r03.py $T > gpurun_out/step_timeline_$TAG.md 2> gpurun_out/step_timeline_$TAG.err; sed -n 3,30p gpurun_out/step_timeline_$TAG.md; tail -3 gpurun_out/step_timeline_$TAG.err
S=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/kernel_stats_$TAG.csv
rm -rf gpurun_out/prof_$TAG
timeout -s KILL 200 python tools/soaked_probe.py save /tmp/soaked.npz 1000 > gpurun_out/soaked_save_$TAG.log 2>&1; tail -1 gpurun_out/soaked_save_$TAG.log
cd /tmp
for STATE in fresh trained; do
  if [ $STATE = fresh ]; then CMD="python $R/bench.py --pmc-child --steps 8 --warmup 0"; else CMD="python $R/tools/soaked_probe.py run /tmp/soaked.npz 8"; fi
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_sq_${STATE}_$TAG -o r03 -- $CMD > $R/gpurun_out/pmc_sq_${STATE}_$TAG.log 2>&1
  C=$(find $R/gpurun_out/pmc_sq_${STATE}_$TAG -name "*counter_collection.csv" | head -1)
  python $R/tools/sq_summary.py $C "$STATE cloud" > $R/gpurun_out/sq_counters_${STATE}_$TAG.md 2>> $R/gpurun_out/step_timeline_$TAG.err; sed -n 7,12p $R/gpurun_out/sq_counters_${STATE}_$TAG.md
  rm -rf $R/gpurun_out/pmc_sq_${STATE}_$TAG
done
