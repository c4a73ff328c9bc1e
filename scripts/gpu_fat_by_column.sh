#!/bin/bash
# gpurun -- bash scripts/gpu_fat_by_column.sh [lib ...] : kernel trace of three headline steps per build
# ("tree" or a path under the repo root), fat-step duration per column pair (scripts/fat_by_column.py)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
LIBS=${@:-tree}
for lib in $LIBS; do
  if [ $lib = tree ]; then unset NGP_LIB; else export NGP_LIB=$R/$lib; fi
  name=$(basename $lib .so)
  rm -rf $R/gpurun_out/fbc
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/fbc -- python3 $R/bench.py --steps 2 --warmup 1 --headline-only > $R/gpurun_out/fbc.log 2>&1 || { tail -5 $R/gpurun_out/fbc.log; exit 1; }
  F=$(find $R/gpurun_out/fbc -name "*kernel_trace.csv" | head -1)
  echo "== $name"
  python3 $R/scripts/fat_by_column.py $F | tee $R/gpurun_out/fat_by_column_$name.txt
  rm -rf $R/gpurun_out/fbc
done
