#!/bin/bash
# usage: tools/gpu_tests_all.sh TAG -> the whole GPU suite without -x (every failure listed), flip counts collected
TAG=$1
mkdir -p gpurun_out
rm -f gpurun_out/flip_counts.jsonl
LITEGS_COLLECT_FLIPS=1 timeout -s KILL 900 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/pytest_$TAG.log 2>&1
grep -E "passed|failed|error" gpurun_out/pytest_$TAG.log | tail -3
