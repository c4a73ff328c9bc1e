#!/bin/bash
# Timing ablations of the fat kernel (results are wrong by construction).  Usage: gpurun -- bash scripts/gpu_ablate.sh
for v in 0 1 2 3 4 7 0; do
  echo "== NGP_ABLATE=$v"; NGP_ABLATE=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fit --particles 16 | grep -o 'kernels_ms_per_step.\{60\}'
done
