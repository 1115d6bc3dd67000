#!/bin/bash
# the round's long-horizon run on the final build: 3 M Gaussians, 150 cameras @1080p, 30 000 iterations, executor x3 (speculative culling) + operator path
mkdir -p gpurun_out
timeout -s KILL 1100 python tests/convergence_3m.py --out gpurun_out/convergence_3m_final.md > gpurun_out/convergence_3m_final.log 2>&1; tail -50 gpurun_out/convergence_3m_final.log
