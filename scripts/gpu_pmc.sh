#!/bin/bash
# PMC passes of the shipped build (each its own run, --kernel-trace only, per the gpurun rules) ->
# gpurun_out/pmc_<config>[_grad].json + the raw per-kernel sums as text (copy both into profiles/rNN/).
# Usage: gpurun -- bash scripts/gpu_pmc.sh C3 [particles|-] [predict|grad] [scenarios|-] [prior|fitted]
CFG=${1:-C3}; PART=${2:--}; MODE=${3:-predict}; SCEN=${4:--}; ENS=${5:-prior}
NAME=${CFG}; [ "$MODE" = "grad" ] && NAME=${CFG}_grad; [ "$ENS" = "fitted" ] && NAME=${NAME}_fitted
TAG=pmc_${NAME}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/scripts/pmc_workload.py $CFG $PART $MODE $SCEN $ENS"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -- $CMD > $R/gpurun_out/${TAG}_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/${TAG}_write -- $CMD > $R/gpurun_out/${TAG}_write.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/${TAG}_sq -- $CMD > $R/gpurun_out/${TAG}_sq.log 2>&1
cd $R
WL=$(grep "^items" gpurun_out/${TAG}_fetch.log | tail -1)
python3 scripts/pmc_to_json.py $TAG $NAME "scripts/pmc_workload.py $CFG $PART $MODE $SCEN $ENS: $WL" | tee gpurun_out/${TAG}.txt
python3 - <<PY | tee -a gpurun_out/${TAG}.txt
import csv, glob, collections
files = glob.glob("gpurun_out/${TAG}_sq/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(files[0])):
    agg[row["Kernel_Name"].split("(")[0].replace("ngp::", "")[-60:]][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in agg.items():
    if v.get("SQ_WAVE_CYCLES", 0) > 1e8:
        print("sq", k, {a: f"{b:.4g}" for a, b in v.items()})
PY
rm -rf gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_sq
