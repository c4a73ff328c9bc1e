#!/bin/bash
# Same-box A/B of two builds of libngp on one bench.py command line: build/libngp_base.so (built from
# a git revision by scripts/build_base_lib.sh) against the working tree's nowcastautogp_amd/libngp.so.
# Usage: gpurun -- bash scripts/gpu_ab_lib.sh TAG <bench.py arguments>
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for round in 1 2; do
  for which in base new; do
    if [ $which = base ]; then export NGP_LIB=$R/build/libngp_base.so; else unset NGP_LIB; fi
    python3 $R/bench.py "$@" > $R/gpurun_out/ab_${TAG}_${which}_${round}.json 2> $R/gpurun_out/ab_${TAG}_${which}_${round}.err || { tail -5 $R/gpurun_out/ab_${TAG}_${which}_${round}.err; exit 1; }
    python3 - <<PY
import json
d = json.loads(open("$R/gpurun_out/ab_${TAG}_${which}_${round}.json").read().strip().splitlines()[-1])
print("$which round $round: %.1f ms/step  " % d["ms_per_step"] + "  ".join("%s %.1f" % (k, v) for k, v in d["kernels_ms_per_step"].items()))
PY
  done
done
