#!/bin/bash
# Round-2 evidence: bench lines (C3 full, C5 full), rocprofv3 kernel stats of the headline-only
# commands, PMC + clock passes of C5.  Usage: gpurun --timeout 1150 -- bash scripts/gpu_r2_profiles.sh
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python bench.py --steps 5 --warmup 2 > gpurun_out/r02_bench_c3.json 2> gpurun_out/r02_bench_c3.err
python bench.py --config C5 --steps 5 --warmup 2 > gpurun_out/r02_bench_c5.json 2> gpurun_out/r02_bench_c5.err
cd /tmp && export TMPDIR=/tmp
for CFG in C3 C5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof_$CFG -- python3 $R/bench.py --config $CFG --steps 3 --warmup 0 --headline-only > $R/gpurun_out/r02_prof_$CFG.log 2>&1
  find $R/gpurun_out/r02_prof_$CFG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/r02_kernel_stats_${CFG}_headline_only.csv
  grep '^{' $R/gpurun_out/r02_prof_$CFG.log > $R/gpurun_out/r02_bench_${CFG}_profiled_command.json
done
cd $R
bash scripts/gpu_pmc.sh C5
bash scripts/gpu_clock.sh C5 64
head -12 gpurun_out/r02_kernel_stats_C3_headline_only.csv | cut -c1-160
head -12 gpurun_out/r02_kernel_stats_C5_headline_only.csv | cut -c1-160
