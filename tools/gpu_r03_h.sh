#!/bin/bash
# round 3, call h: parity suite (speculative culling, loss value in the backward, spread scatter atomics), bench, timeline
TAG=${1:-h}
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/flip_counts.jsonl
LITEGS_COLLECT_FLIPS=1 timeout -s KILL 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_$TAG.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_$TAG.log | tail -1; grep -E "FAILED|Error|assert" gpurun_out/pytest_$TAG.log | head
timeout -s KILL 400 python bench.py > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-1000
LITEGS_SPECULATIVE=0 timeout -s KILL 400 python bench.py --no-cpu-baseline --no-operator-path --no-pmc > gpurun_out/bench_nospec_$TAG.log 2>&1; tail -1 gpurun_out/bench_nospec_$TAG.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o r03 -- python $R/bench.py --no-cpu-baseline --no-operator-path --no-pmc > $R/gpurun_out/rocprof_$TAG.log 2>&1
cd $R
T=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python tools/profile_r03.py $T > gpurun_out/step_timeline_$TAG.md 2> gpurun_out/step_timeline_$TAG.err; sed -n 3,40p gpurun_out/step_timeline_$TAG.md; tail -3 gpurun_out/step_timeline_$TAG.err
S=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/kernel_stats_$TAG.csv
rm -rf gpurun_out/prof_$TAG
