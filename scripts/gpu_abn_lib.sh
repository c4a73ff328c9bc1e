#!/bin/bash
# Same-box comparison of several builds of libngp on one bench.py command line.
# Usage: gpurun -- bash scripts/gpu_abn_lib.sh TAG "lib1 lib2 ..." <bench.py arguments>
#        (a lib is a path under the repo root, or "tree" for nowcastautogp_amd/libngp.so)
TAG=$1; LIBS=$2; shift 2
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for round in 1 2; do
  for lib in $LIBS; do
    if [ $lib = tree ]; then unset NGP_LIB; else export NGP_LIB=$R/$lib; fi
    name=$(basename $lib .so)
    python3 $R/bench.py "$@" > $R/gpurun_out/abn_${TAG}_${name}_${round}.json 2> $R/gpurun_out/abn_${TAG}_${name}_${round}.err || { tail -5 $R/gpurun_out/abn_${TAG}_${name}_${round}.err; exit 1; }
    python3 - <<PY
import json
d = json.loads(open("$R/gpurun_out/abn_${TAG}_${name}_${round}.json").read().strip().splitlines()[-1])
print("%-14s round $round: %.1f ms/step  " % ("$name", d["ms_per_step"]) + "  ".join("%s %.1f" % (k, v) for k, v in d["kernels_ms_per_step"].items()))
PY
  done
done
