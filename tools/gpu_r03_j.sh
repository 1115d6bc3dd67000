#!/bin/bash
# round 3, call j: gradient replicas -- parity suite, blend-backward A/B (replicas on / off), bench
TAG=${1:-j}
mkdir -p gpurun_out
rm -f gpurun_out/flip_counts.jsonl
LITEGS_COLLECT_FLIPS=1 timeout -s KILL 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_$TAG.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_$TAG.log | tail -1; grep -E "FAILED|Error|assert" gpurun_out/pytest_$TAG.log | head
LITEGS_GRAD_REPLICAS=1 timeout -s KILL 300 python tools/bwd_ab.py 3m_1080p 1000 > gpurun_out/bwd_ab_rep1_$TAG.log 2>&1; grep "fast kernel  \|fast kernel, normal\|fast kernel, atomics off" gpurun_out/bwd_ab_rep1_$TAG.log
LITEGS_GRAD_REPLICAS=0 timeout -s KILL 300 python tools/bwd_ab.py 3m_1080p 1000 > gpurun_out/bwd_ab_rep0_$TAG.log 2>&1; grep "fast kernel  \|fast kernel, normal\|fast kernel, atomics off" gpurun_out/bwd_ab_rep0_$TAG.log
timeout -s KILL 400 python bench.py --no-cpu-baseline --no-operator-path --no-pmc > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-330
LITEGS_GRAD_REPLICAS=0 timeout -s KILL 400 python bench.py --no-cpu-baseline --no-operator-path --no-pmc > gpurun_out/bench_rep0_$TAG.log 2>&1; tail -1 gpurun_out/bench_rep0_$TAG.log | cut -c1-330
