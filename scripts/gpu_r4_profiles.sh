#!/bin/bash
# Round-4 evidence, one call: rocprofv3 kernel stats of the headline-only commands (C3 predict, C3
# gradient mode on the prior and on the fitted ensemble, C5), PMC passes (C3 predict, both gradient
# ensembles, C5) and the clock pass of C3.
# Usage: gpurun --timeout 1150 -- bash scripts/gpu_r4_profiles.sh [stats|pmc|all]   (then copy gpurun_out/r04_* into profiles/r04/)
WHAT=${1:-all}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
prof() {   # tag, bench.py arguments
  TAG=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_prof_$TAG -- python3 $R/bench.py "$@" --headline-only > $R/gpurun_out/r04_prof_$TAG.log 2>&1 || return 1
  find $R/gpurun_out/r04_prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/r04_kernel_stats_${TAG}_headline_only.csv
  grep '^{' $R/gpurun_out/r04_prof_$TAG.log > $R/gpurun_out/r04_bench_${TAG}_profiled_command.json
  rm -rf $R/gpurun_out/r04_prof_$TAG
  echo "== $TAG"; head -6 $R/gpurun_out/r04_kernel_stats_${TAG}_headline_only.csv | cut -c1-150
}
if [ "$WHAT" = "stats" ] || [ "$WHAT" = "all" ]; then
prof C3 --steps 3 --warmup 0 &&
prof C3_grad --mode grad --steps 2 --warmup 0 &&
prof C3_grad_fitted --mode grad --ensemble fitted --steps 2 --warmup 0 &&
prof C5 --config C5 --steps 3 --warmup 0 || exit 1
fi
if [ "$WHAT" = "pmc" ] || [ "$WHAT" = "all" ]; then
cd $R &&
bash scripts/gpu_pmc.sh C3 64 predict 50 &&
bash scripts/gpu_pmc.sh C3 64 grad 25 &&
bash scripts/gpu_pmc.sh C3 64 grad 25 fitted &&
bash scripts/gpu_pmc.sh C5 - predict - &&
bash scripts/gpu_clock.sh C3 64 50
fi
