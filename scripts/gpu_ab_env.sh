#!/bin/bash
# A/B an environment switch on the SAME box, interleaved.  Usage: gpurun -- bash scripts/gpu_ab_env.sh VAR [bench args]
VAR=$1; shift
for r in 1 2; do for v in 0 1; do
  echo "== $VAR=$v round $r"; env $VAR=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fit "$@" | grep -o '"ms_per_step.\{22\}\|kernels_ms_per_step.\{90\}\|failed_items.\{5\}'
done; done
