#!/bin/bash
# A/B two builds of libngp on the SAME box, interleaved: build/libngp_a.so (a baseline built from an older revision, kept out of the package directory) against the shipped nowcastautogp_amd/libngp.so.  Usage: gpurun -- bash scripts/gpu_ab.sh [extra bench args]
for r in 1 2; do
for v in a b; do
  lib=$GRAFT_REPO_ROOT/build/libngp_$v.so; [ $v = b ] && lib=$GRAFT_REPO_ROOT/nowcastautogp_amd/libngp.so
  echo "== $v round $r"; NGP_LIB=$lib python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fit "$@" | grep -o '"ms_per_step.\{22\}\|kernels_ms_per_step.\{200\}'
done; done
