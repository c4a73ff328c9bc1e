#!/bin/bash
# Round-3 evidence, one call: rocprofv3 kernel stats of the headline-only commands (C3 predict, C3
# gradient mode, C5), PMC passes (C3 predict, C3 gradient mode) and the clock pass of C3.
# Usage: gpurun --timeout 1150 -- bash scripts/gpu_r3_profiles.sh      (then copy gpurun_out/r03b_* into profiles/r03/)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
prof() {   # tag, bench.py arguments
  TAG=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03b_prof_$TAG -- python3 $R/bench.py "$@" --headline-only > $R/gpurun_out/r03b_prof_$TAG.log 2>&1 || return 1
  find $R/gpurun_out/r03b_prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/r03b_kernel_stats_${TAG}_headline_only.csv
  grep '^{' $R/gpurun_out/r03b_prof_$TAG.log > $R/gpurun_out/r03b_bench_${TAG}_profiled_command.json
  rm -rf $R/gpurun_out/r03b_prof_$TAG
  echo "== $TAG"; head -8 $R/gpurun_out/r03b_kernel_stats_${TAG}_headline_only.csv | cut -c1-150
}
prof C3 --steps 3 --warmup 0 &&
prof C3_grad --mode grad --steps 2 --warmup 0 &&
prof C5 --config C5 --steps 3 --warmup 0 &&
cd $R &&
bash scripts/gpu_pmc.sh C3 64 predict 50 &&
bash scripts/gpu_pmc.sh C3 64 grad 25 &&
bash scripts/gpu_clock.sh C3 64 50
