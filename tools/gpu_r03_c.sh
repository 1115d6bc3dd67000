#!/bin/bash
# round 3, third GPU call: parity suite with the tile scatter (collecting flip counts), scatter A/B, bench, 30k-iteration convergence at 3 M
TAG=${1:-c}
mkdir -p gpurun_out
rm -f gpurun_out/flip_counts.jsonl
LITEGS_COLLECT_FLIPS=1 timeout -s KILL 900 python -m pytest tests -m gpu -q -x --durations=6 > gpurun_out/pytest_$TAG.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_$TAG.log | tail -1; grep -E "FAILED|Error" gpurun_out/pytest_$TAG.log | head
timeout -s KILL 300 python tools/scatter_ab.py > gpurun_out/scatter_ab_$TAG.log 2>&1; tail -9 gpurun_out/scatter_ab_$TAG.log
timeout -s KILL 400 python bench.py > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-1500
timeout -s KILL 900 python tests/convergence_3m.py --out gpurun_out/convergence_3m_$TAG.md > gpurun_out/convergence_3m_$TAG.log 2>&1; tail -45 gpurun_out/convergence_3m_$TAG.log
