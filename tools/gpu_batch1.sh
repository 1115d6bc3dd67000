#!/bin/bash
# round 2, GPU batch 1: instruction-cost micro-benchmark, per-tile work distribution, flip-count collection run
mkdir -p gpurun_out
rm -f gpurun_out/flip_counts.jsonl
./tools/ubench/valu_ubench > gpurun_out/ubench.log 2>&1
timeout 300 python tools/tile_stats.py 3m_1080p > gpurun_out/tile_stats.log 2>&1
LITEGS_COLLECT_FLIPS=1 timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_collect.log 2>&1
tail -5 gpurun_out/pytest_collect.log
cat gpurun_out/ubench.log
tail -3 gpurun_out/tile_stats.log
