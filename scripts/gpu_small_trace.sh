#!/bin/bash
# timeline of the 64-particle calls: gpurun -- bash scripts/gpu_small_trace.sh [n] [P]
N=${1:-2048}; P=${2:-64}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for KIND in logml grad; do
  rm -rf $R/gpurun_out/sbt_$KIND
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sbt_$KIND -- python3 $R/scripts/small_batch_trace.py $KIND $N $P > $R/gpurun_out/sbt_$KIND.log 2>&1 || exit 1
  F=$(find $R/gpurun_out/sbt_$KIND -name "*kernel_trace.csv" | head -1)
  python3 $R/scripts/small_batch_trace.py --analyse $F > $R/gpurun_out/sbt_${KIND}_timeline.txt
  tail -14 $R/gpurun_out/sbt_${KIND}_timeline.txt
  rm -rf $R/gpurun_out/sbt_$KIND
done
