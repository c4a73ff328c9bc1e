#!/bin/bash
# Clock the chip holds under the fat kernel: GRBM_GUI_ACTIVE / 8 / dispatch duration (guide: DVFS give-back),
# MFMA-busy share of those cycles.  Usage: gpurun -- bash scripts/gpu_clock.sh C3 16
CFG=${1:-C3}; PART=${2:-16}; SCEN=${3:--}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/clock_${CFG} -- python3 $R/scripts/pmc_workload.py $CFG $PART predict $SCEN > $R/gpurun_out/clock_${CFG}.log 2>&1
cd $R
python3 - <<PY | tee gpurun_out/clock_${CFG}.txt
import csv, glob, collections
f = glob.glob("gpurun_out/clock_${CFG}/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys()))
by = collections.defaultdict(dict)
for r in rows:
    key = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][-40:])
    by[key][r["Counter_Name"]] = float(r["Counter_Value"])
    for c in ("Start_Timestamp", "End_Timestamp"):
        if c in r: by[key][c] = float(r[c])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for (d, k), v in by.items():
    if "Start_Timestamp" not in v: continue
    dur = v["End_Timestamp"] - v["Start_Timestamp"]
    a = agg[k]; a[0] += 1; a[1] += dur; a[2] += v.get("GRBM_GUI_ACTIVE", 0.0); a[3] += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
for k, (n, dur, gui, mf) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    ghz = gui / 8 / dur if dur else 0
    print(f"{k}: {n} dispatches, {dur/1e6:.1f} ms, clock {ghz:.3f} GHz, MFMA busy {mf/1024/(gui/8) if gui else 0:.3f} of cycles per SIMD")
PY
