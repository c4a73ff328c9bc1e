#!/bin/bash
# microbench detail + bench + rocprof kernel stats.  Usage: gpurun -- bash scripts/gpu_bench.sh [tag]
set -o pipefail
TAG=${1:-r01}
mkdir -p gpurun_out
python - <<'PY' 2>&1 | tee gpurun_out/microbench_${TAG}.log
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib
c = _lib.Context(0)
for bpc in (1, 2, 4):
    print(c.microbench_mfma_f64_detail(40000, bpc))
w, cp = c.microbench_hbm(1 << 31)
print(f"hbm write {w:.0f} GB/s  copy {cp:.0f} GB/s")
PY
python bench.py --steps 2 --warmup 1 2>gpurun_out/bench_${TAG}.err | tee gpurun_out/bench_${TAG}.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --headline-only > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_${TAG}.csv
head -20 gpurun_out/kernel_stats_${TAG}.csv
