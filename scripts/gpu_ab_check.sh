#!/bin/bash
# Working tree against build/libngp_base.so: scripts/k8_dump.py's result arrays compared bit for bit,
# then the C3 headline and the fitted gradient line, two rounds each.
# Usage: gpurun -- bash scripts/gpu_ab_check.sh TAG
TAG=${1:-ab}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
set -e
NGP_LIB=$R/build/libngp_base.so python3 $R/scripts/k8_dump.py $R/gpurun_out/${TAG}_base.npz
python3 $R/scripts/k8_dump.py $R/gpurun_out/${TAG}_new.npz
python3 - <<PY
import numpy as np
a, b = np.load("$R/gpurun_out/${TAG}_base.npz"), np.load("$R/gpurun_out/${TAG}_new.npz")
bad = 0
for k in a.files:
    if not np.array_equal(a[k], b[k], equal_nan=True):
        bad += 1
        d = np.abs(a[k].astype(float) - b[k].astype(float))
        print("DIFF", k, "max abs", np.nanmax(d), "rel", np.nanmax(d / (np.abs(a[k]) + 1e-300)))
print("bit-identical arrays:", len(a.files) - bad, "of", len(a.files), " info all zero:", all(not a[k].any() for k in a.files if k.endswith("info")))
PY
bash $R/scripts/gpu_ab_lib.sh ${TAG}_c3 --steps 6 --warmup 2 --headline-only
bash $R/scripts/gpu_ab_lib.sh ${TAG}_gradfit --mode grad --ensemble fitted --steps 2 --warmup 1 --headline-only
