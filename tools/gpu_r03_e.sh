#!/bin/bash
# round 3, call e: parity suite (LDS-aggregated tile scatter, packed blend forward), A/Bs, long density-control timing, bench
TAG=${1:-e}
mkdir -p gpurun_out
rm -f gpurun_out/flip_counts.jsonl
LITEGS_COLLECT_FLIPS=1 timeout -s KILL 900 python -m pytest tests -m gpu -q -x --durations=6 > gpurun_out/pytest_$TAG.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_$TAG.log | tail -1; grep -E "FAILED|Error" gpurun_out/pytest_$TAG.log | head
timeout -s KILL 300 python tools/scatter_ab.py > gpurun_out/scatter_ab_$TAG.log 2>&1; tail -9 gpurun_out/scatter_ab_$TAG.log
timeout -s KILL 300 python tools/bwd_ab.py > gpurun_out/bwd_ab_$TAG.log 2>&1; tail -26 gpurun_out/bwd_ab_$TAG.log
timeout -s KILL 300 python tools/densify_profile.py executor 150 42 > gpurun_out/densify_executor_$TAG.log 2>&1; tail -42 gpurun_out/densify_executor_$TAG.log
timeout -s KILL 400 python bench.py > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-1200
