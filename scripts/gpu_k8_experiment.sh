#!/bin/bash
# The 8-deep four-buffer k-loop experiment: working tree against build/libngp_base.so (HEAD).
# Usage: gpurun -- bash scripts/gpu_k8_experiment.sh
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
set -e
NGP_LIB=$R/build/libngp_base.so python3 $R/scripts/k8_dump.py $R/gpurun_out/k8_base.npz
python3 $R/scripts/k8_dump.py $R/gpurun_out/k8_new.npz
python3 - <<PY
import numpy as np
a, b = np.load("$R/gpurun_out/k8_base.npz"), np.load("$R/gpurun_out/k8_new.npz")
bad = 0
for k in a.files:
    same = np.array_equal(a[k], b[k], equal_nan=True)
    if not same:
        bad += 1
        d = np.abs(a[k].astype(float) - b[k].astype(float))
        print("DIFF", k, "max abs", np.nanmax(d), "rel", np.nanmax(d / (np.abs(a[k]) + 1e-300)))
print("bit-identical arrays:", len(a.files) - bad, "of", len(a.files), " info all zero:", all(not a[k].any() for k in a.files if k.endswith("info")))
PY
bash $R/scripts/gpu_ab_lib.sh k8_c3 --steps 6 --warmup 2 --headline-only
bash $R/scripts/gpu_ab_lib.sh k8_gradfit --mode grad --ensemble fitted --steps 2 --warmup 1 --headline-only
