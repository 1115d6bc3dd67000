#!/bin/bash
# usage: tools/gpu_r02_final.sh TAG -> what the driver runs at round end (GPU tests, smoke, default bench) + the round's profiles:
# kernel-trace stats (3M, 500k, 10M), FETCH_SIZE / WRITE_SIZE PMC passes, SQ counters of the step
TAG=$1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/flip_counts.jsonl
LITEGS_COLLECT_FLIPS=1 timeout -s KILL 900 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/pytest_$TAG.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_$TAG.log | tail -1; grep FAILED gpurun_out/pytest_$TAG.log | head
timeout -s KILL 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; tail -1 gpurun_out/smoke_$TAG.log
timeout -s KILL 240 python bench.py > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log
timeout -s KILL 200 python bench.py --config 10m_1600x1200 --no-cpu-baseline --no-operator-path --frames 4 > gpurun_out/bench_10m_$TAG.log 2>&1; tail -1 gpurun_out/bench_10m_$TAG.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o r02 -- python $R/bench.py --steps 40 --warmup 16 --no-cpu-baseline --no-operator-path > $R/gpurun_out/rocprof_$TAG.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_500k -o r02 -- python $R/bench.py --config 500k_1080p --steps 40 --warmup 16 --no-cpu-baseline --no-operator-path > $R/gpurun_out/rocprof_${TAG}_500k.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -o r02 -- python $R/bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-operator-path > $R/gpurun_out/pmc_$C.log 2>&1
done
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_sqA -o r02 -- python $R/bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-operator-path > $R/gpurun_out/pmc_sqA.log 2>&1
ls $R/gpurun_out | head -50
