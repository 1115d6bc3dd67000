#!/bin/bash
# round 3, call g: parity suite with the vector-path blend forward and the LDS-staged offsets kernel, A/Bs, both-state timeline
TAG=${1:-g}
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/flip_counts.jsonl
LITEGS_COLLECT_FLIPS=1 timeout -s KILL 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_$TAG.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_$TAG.log | tail -1; grep -E "FAILED|Error" gpurun_out/pytest_$TAG.log | head
timeout -s KILL 300 python tools/bwd_ab.py > gpurun_out/bwd_ab_$TAG.log 2>&1; grep "forward only" gpurun_out/bwd_ab_$TAG.log
timeout -s KILL 300 python tools/scatter_ab.py > gpurun_out/scatter_ab_$TAG.log 2>&1; tail -8 gpurun_out/scatter_ab_$TAG.log
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o r03 -- python $R/bench.py --no-cpu-baseline --no-operator-path --no-pmc > $R/gpurun_out/rocprof_$TAG.log 2>&1
cd $R
T=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python tools/profile_r03.py $T > gpurun_out/step_timeline_$TAG.md 2> gpurun_out/step_timeline_$TAG.err; sed -n 8,34p gpurun_out/step_timeline_$TAG.md; tail -3 gpurun_out/step_timeline_$TAG.err
S=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/kernel_stats_$TAG.csv
rm -rf gpurun_out/prof_$TAG
