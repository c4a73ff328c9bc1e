#!/bin/bash
# Everything profiles/r04 holds about the short-series path, in one gpurun call:
#   gpurun -- bash scripts/gpu_short_series_profiles.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/short
mkdir -p $O
cd $R
timeout -k 10 300 python3 scripts/short_series_probe.py > $O/short_series_probe.txt 2>&1 || exit 1
timeout -k 10 200 python3 scripts/resident_grad_overhead.py > $O/resident_grad_runs.txt 2>&1 || exit 1
( cd scripts/ubench && timeout -k 5 60 ./small_bench 208 0 24 && timeout -k 5 60 ./small_bench 208 1 24 && timeout -k 5 60 ./small_bench 256 1 24 ) > $O/short_series_stamps_final.txt 2>&1 || exit 1
timeout -k 10 400 python3 scripts/vignette_fit_probe.py > $O/vignette_fit.txt 2>&1 || exit 1
bash scripts/gpu_small_trace.sh 208 24 > $O/small_trace.log 2>&1 || exit 1
cp $R/gpurun_out/sbt_logml_timeline.txt $O/timeline_logml_24x208.txt
cp $R/gpurun_out/sbt_grad_timeline.txt $O/timeline_grad_24x208.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/scripts/everyday_calls_loop.py > $O/everyday_calls_loop.log 2>&1 || exit 1
F=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp $F $O/kernel_stats_everyday_calls_24x208.csv
rm -rf $O/prof
tail -3 $O/vignette_fit.txt
head -12 $O/kernel_stats_everyday_calls_24x208.csv
