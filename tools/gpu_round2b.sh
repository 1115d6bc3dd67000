#!/bin/bash
# full GPU suite + margin sweep + bench with the new defaults
mkdir -p gpurun_out
timeout -s KILL 500 python tools/margin_ab.py 3m_1080p 40 global 0,75,100,125,150 2>&1 | grep -v amdgpu.ids > gpurun_out/margin_ab2.log; cat gpurun_out/margin_ab2.log
timeout -s KILL 200 python bench.py --no-cpu-baseline > gpurun_out/bench_m100.log 2>&1; tail -1 gpurun_out/bench_m100.log | cut -c1-420
LITEGS_COLLECT_FLIPS=1 timeout -s KILL 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_all2.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_all2.log | tail -2; grep FAILED gpurun_out/pytest_all2.log | head
