#!/bin/bash
# usage: tools/gpu_r02_status.sh TAG -> GPU tests, smoke, default bench, kernel-trace stats (3M, 500k) and FETCH/WRITE PMC passes
TAG=$1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/flip_counts.jsonl
LITEGS_COLLECT_FLIPS=1 timeout -s KILL 600 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/pytest_$TAG.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_$TAG.log | tail -2
timeout -s KILL 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; tail -1 gpurun_out/smoke_$TAG.log
timeout -s KILL 240 python bench.py > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o r02 -- python $R/bench.py --steps 40 --warmup 16 --no-cpu-baseline > $R/gpurun_out/rocprof_$TAG.log 2>&1
tail -1 $R/gpurun_out/rocprof_$TAG.log | cut -c1-400
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -o r02 -- python $R/bench.py --steps 8 --warmup 8 --no-cpu-baseline > $R/gpurun_out/pmc_$C.log 2>&1
done
ls $R/gpurun_out
