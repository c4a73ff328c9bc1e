#!/bin/bash
# Round-2 GPU session: whole GPU suite (stops at the first failure), then the C5 timing table.
# Usage: gpurun --timeout 1100 -- bash scripts/gpu_r2_check.sh
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -12 gpurun_out/pytest_gpu.log | cut -c1-250
if grep -q "Memory access fault" gpurun_out/pytest_gpu.log; then exit 3; fi
[ $rc -eq 0 ] || exit $rc
python scripts/mixed_timing.py > gpurun_out/mixed_timing.log 2>&1 && cat gpurun_out/mixed_timing.log
