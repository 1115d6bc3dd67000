#!/bin/bash
# usage: tools/gpu_dp_sanity.sh -> DP tests + the N>1 control flow of bench.py on a one-GPU box (ranks share cuda:0 over gloo: not a measurement)
mkdir -p gpurun_out
timeout -s KILL 300 python -m pytest tests/test_gpu_dp.py tests/test_gpu_training.py -x -q > gpurun_out/pytest_dp.log 2>&1; tail -3 gpurun_out/pytest_dp.log
LITEGS_BENCH_ONE_GPU=1 timeout -s KILL 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 4 --config 500k_1080p > gpurun_out/bench_dp2.log 2>&1; tail -2 gpurun_out/bench_dp2.log | cut -c1-900
