#!/bin/bash
# Round-2 GPU session: mixed-precision tests first (new code), then the whole GPU suite.
# Usage: gpurun --timeout 1100 -- bash scripts/gpu_r2_check.sh
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_mixed_gpu.py -m gpu -x -q -s 2>&1 | tee gpurun_out/pytest_mixed.log | tail -40 &&
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_mixed_gpu.py 2>&1 | tee gpurun_out/pytest_gpu.log | tail -15
