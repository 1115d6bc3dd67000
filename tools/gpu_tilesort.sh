#!/bin/bash
# usage: tools/gpu_tilesort.sh -> unit + end-to-end tests of the per-tile depth sort and of the emission, A/B bench and kernel traces of both depth-order modes
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 400 python -m pytest tests/test_gpu_tilesort.py tests/test_gpu_ops.py tests/test_gpu_edge.py tests/test_gpu_fused.py tests/test_gpu_cull.py tests/test_gpu_fullsize.py tests/test_gpu_tilesizes.py -x -q > gpurun_out/pytest_tilesort.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_tilesort.log | tail -1; grep -E "^E  " gpurun_out/pytest_tilesort.log | head -6
timeout -s KILL 300 python tools/margin_ab.py 3m_1080p 40 global,tile 100 2>&1 | grep -v amdgpu.ids > gpurun_out/margin_ab3.log; cat gpurun_out/margin_ab3.log
cd /tmp && export TMPDIR=/tmp
for MODE in tile global; do
  LITEGS_DEPTH_ORDER=$MODE timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$MODE -o r02 -- python $R/bench.py --steps 40 --warmup 16 --no-cpu-baseline --no-operator-path > $R/gpurun_out/rocprof_$MODE.log 2>&1
done
