#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python script.  Usage: gpurun -- bash scripts/gpu_prof_cmd.sh TAG script.py [args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python3 $R/"$@" > $R/gpurun_out/prof_${TAG}.log 2>&1
cd $R
find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_${TAG}.csv
cut -c1-200 gpurun_out/kernel_stats_${TAG}.csv | head -30
tail -5 gpurun_out/prof_${TAG}.log
