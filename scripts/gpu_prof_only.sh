#!/bin/bash
# rocprof kernel stats of one headline step.  Usage: gpurun -- bash scripts/gpu_prof_only.sh [tag]
TAG=${1:-x}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --headline-only > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_${TAG}.csv
python3 - gpurun_out/kernel_stats_${TAG}.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"{r['Name'][:58]:58s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
grep -o '"ms_per_step.\{22\}' gpurun_out/prof_${TAG}.log
