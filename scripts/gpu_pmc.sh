#!/bin/bash
# PMC passes (each its own run, --kernel-trace only, per the gpurun rules).  Usage: gpurun -- bash scripts/gpu_pmc.sh tag
TAG=${1:-pmc}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fit --particles 16"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/${TAG}_sq -- $CMD > $R/gpurun_out/${TAG}_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -- $CMD > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/${TAG}_write -- $CMD > $R/gpurun_out/${TAG}_write.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
for sub in ("sq","fetch","write"):
    files = glob.glob(f"gpurun_out/${TAG}_{sub}/**/*counter_collection.csv", recursive=True)
    if not files: print(sub, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0][-40:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in agg.items():
        print(sub, k, {a: f"{b:.4g}" for a, b in v.items()})
PY
