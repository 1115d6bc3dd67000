#!/bin/bash
mkdir -p gpurun_out
timeout -s KILL 300 python -m pytest tests/test_gpu_adam_skip.py tests/test_gpu_cull.py tests/test_gpu_tilesort.py tests/test_gpu_fused.py tests/test_gpu_pipeline.py tests/test_gpu_convergence.py tests/test_gpu_refine.py -x -q > gpurun_out/pytest_adamskip.log 2>&1; tail -12 gpurun_out/pytest_adamskip.log
timeout -s KILL 200 python bench.py --no-cpu-baseline > gpurun_out/bench_skip.log 2>&1; tail -1 gpurun_out/bench_skip.log | cut -c1-300
LITEGS_ADAM_SKIP_UNTOUCHED=0 timeout -s KILL 200 python bench.py --no-cpu-baseline --no-operator-path > gpurun_out/bench_noskip.log 2>&1; tail -1 gpurun_out/bench_noskip.log | cut -c1-300
