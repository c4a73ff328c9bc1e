#!/bin/bash
# gradient parity tests, then a kernel trace of the P=64 latency probe
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "grad or hmc or fit" 2>&1 | tail -5 || exit 1
python scripts/latency_probe.py 2>&1 | tail -18
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gradprof -- python3 $GRAFT_REPO_ROOT/scripts/latency_probe.py logml_grad 2048 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/gradprof -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
