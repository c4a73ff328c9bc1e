#!/bin/bash
# One GPU-box session: microbenchmarks, parity tests, smoke.  Usage: gpurun -- bash scripts/gpu_check.sh
set -o pipefail
mkdir -p gpurun_out
python - <<'PY' 2>&1 | tee gpurun_out/microbench.log
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib
c = _lib.Context(0)
print("version", _lib.load().ngp_version().decode())
for it in (4000, 40000):
    print(f"mfma_f64 iters={it}: {c.microbench_mfma_f64(it):.2f} TFLOP/s")
w, cp = c.microbench_hbm(1 << 31)
print(f"hbm write {w:.0f} GB/s  copy {cp:.0f} GB/s")
PY
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu.log | tail -25
