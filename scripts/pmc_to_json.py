"""rocprofv3 --pmc counter_collection.csv files (one pass each for FETCH_SIZE and WRITE_SIZE) ->
profiles/rNN/pmc_<config>[_grad].json, the file bench.py reads its roofline.traffic from.

Units are calibrated on the box, not assumed: the same passes contain ngp's stream_copy_kernel on
1 GiB (reads 2^30 B, writes 2^30 B per launch); the factor that makes FETCH_SIZE / WRITE_SIZE of
that kernel equal its known byte count is applied to every other kernel (on gfx950 the guide
expects FETCH_SIZE x 1024 x 2 and WRITE_SIZE x 1024 — the json records what was found)."""
import collections
import csv
import glob
import json
import subprocess
import sys

tag, config, workload = sys.argv[1], sys.argv[2], sys.argv[3]
GIB = float(1 << 30)


def load(sub, counter):
    files = glob.glob(f"gpurun_out/{tag}_{sub}/**/*counter_collection.csv", recursive=True)
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"]
        name = name.split("(")[0].replace("void ", "").replace("ngp::", "").strip()
        agg[name] += float(row["Counter_Value"])
        cnt[name] += 1
    return agg, cnt


fetch, nf = load("fetch", "FETCH_SIZE")
write, nw = load("write", "WRITE_SIZE")
cal_r = GIB / (fetch["stream_copy_kernel"] / nf["stream_copy_kernel"])
cal_w = GIB / (write["stream_copy_kernel"] / nw["stream_copy_kernel"])
kernels = {}
for k in fetch:
    if k.startswith("__amd") or k.startswith("stream_"):
        continue
    kernels[k] = {"launches": nf[k],
                  "read_bytes_per_launch": fetch[k] * cal_r / nf[k],
                  "written_bytes_per_launch": write.get(k, 0.0) * cal_w / max(nw.get(k, 1), 1)}
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = None
import re
m = re.search(r"items (\d+)", workload)
mg, mt = re.search(r"general_items (\d+)", workload), re.search(r"toeplitz_items (\d+)", workload)
items_by_kernel = {}
if mg and mt:   # a gradient job: its two leaves run different kernels over different items
    ng, nt = int(mg.group(1)), int(mt.group(1))
    for k in kernels:
        if k.startswith("chol_col") and ", true" in k or k.startswith("grad_kinv") or k.startswith("grad_alpha"):
            items_by_kernel[k] = ng
        elif k.startswith("chol_col") or k.startswith("aux_back") or k.startswith("toep_"):
            items_by_kernel[k] = nt if nt else ng
out = {"config": config, "workload": workload, "commit": commit,
       "items": int(m.group(1)) if m else None,   # items every launch of the profiled job covered
       "items_by_kernel": items_by_kernel,        # ... per kernel where the leaves differ
       "calibration": {"kernel": "stream_copy_kernel, 2^30 B read + 2^30 B written per launch",
                       "bytes_per_FETCH_SIZE_unit": cal_r, "bytes_per_WRITE_SIZE_unit": cal_w,
                       "guide_expectation": "2048 (1 KiB x 2, gfx950 halving) and 1024"},
       "kernels": kernels}
path = f"gpurun_out/pmc_{config}.json"
json.dump(out, open(path, "w"), indent=1)
print(path, json.dumps(out["calibration"]))
for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["read_bytes_per_launch"] * kv[1]["launches"])[:8]:
    print(f"  {k}: {v['launches']} launches, {v['read_bytes_per_launch']/1e6:.1f} MB read + "
          f"{v['written_bytes_per_launch']/1e6:.1f} MB written per launch")
