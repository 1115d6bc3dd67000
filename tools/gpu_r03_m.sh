#!/bin/bash
# call m: slice descriptors (tests + A/B) and the hunt for the rare memory access fault of the long run
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_tilesort.py tests/test_gpu_fused.py tests/test_gpu_cull.py -x -q -m gpu > gpurun_out/pytest_m.log 2>&1; tail -5 gpurun_out/pytest_m.log
for d in 1 0; do
  LITEGS_SLICE_DESC=$d timeout 120 python bench.py --steps 40 --warmup 16 --no-pmc --no-cpu-baseline --no-operator-path > gpurun_out/bench_m_desc$d.log 2>&1
  grep '^{' gpurun_out/bench_m_desc$d.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('desc=$d', d['ms_per_step'], d['ms_p50'], d['fwd_ms'], d['steady_state']['ms_per_step'], d['steady_state']['fwd_ms'])"
done
PYTORCH_NO_CUDA_MEMORY_CACHING=1 LITEGS_GUARD_ALLOC=1 HIP_LAUNCH_BLOCKING=1 timeout -s KILL 180 python -X faulthandler tools/fault_hunt.py --seconds 130 > gpurun_out/fault_hunt_m.log 2>&1; tail -60 gpurun_out/fault_hunt_m.log
