#!/bin/bash
# kernel trace of five logml+gradient calls at 64 particles.  Usage: gpurun -- bash scripts/gpu_grad_prof.sh [n]
N=${1:-2048}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/gradprof
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gradprof -- python3 $GRAFT_REPO_ROOT/scripts/latency_probe.py logml_grad $N > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/gradprof -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 5e6
print(f"kernel time per call: {tot:.2f} ms")
for r in rows[:12]:
    print(f"{r['Name'][:52]:52s} calls/5={int(r['Calls'])/5:6.1f} ms/call={float(r['TotalDurationNs'])/5e6:8.3f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
