#!/bin/bash
# usage: tools/gpu_quick.sh "<pytest args>" -> targeted tests + default bench (no CPU baseline)
mkdir -p gpurun_out
timeout -s KILL 400 python -m pytest $1 -x -q > gpurun_out/pytest_quick.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_quick.log | tail -2; grep -E "^E  " gpurun_out/pytest_quick.log | head -8
timeout -s KILL 200 python bench.py --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1; tail -1 gpurun_out/bench_quick.log | python -c "
import sys, json
b = json.loads(sys.stdin.read())
print({k: b[k] for k in ('value', 'ms_per_step', 'fwd_msplats_per_s', 'operator_path_ms')})"
