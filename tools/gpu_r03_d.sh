#!/bin/bash
mkdir -p gpurun_out
timeout -s KILL 200 python tools/densify_profile.py executor 30 12 > gpurun_out/densify_executor.log 2>&1; tail -13 gpurun_out/densify_executor.log
timeout -s KILL 200 python tools/densify_profile.py operator 30 12 > gpurun_out/densify_operator.log 2>&1; tail -13 gpurun_out/densify_operator.log
