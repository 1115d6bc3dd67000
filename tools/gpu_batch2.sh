#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/flip_counts.jsonl
LITEGS_COLLECT_FLIPS=1 timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest2.log 2>&1
tail -8 gpurun_out/pytest2.log
timeout 600 python tools/bwd_variants.py > gpurun_out/variants.log 2>&1
cat gpurun_out/variants.log
timeout 600 python bench.py > gpurun_out/bench_r2a.log 2>&1
tail -2 gpurun_out/bench_r2a.log
