#!/bin/bash
# gradient-path timing + the GPU suite.  Usage: gpurun -- bash scripts/gpu_grad_check.sh
set -o pipefail
python3 scripts/grad_timing.py > gpurun_out/grad_timing.log 2>&1 && cat gpurun_out/grad_timing.log && \
python3 -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
