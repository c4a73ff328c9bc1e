#!/usr/bin/env python3
"""Per block column: duration of the fat / thin / chol_diag / diag_ahead launches of the C3 headline
step, from a rocprofv3 kernel trace (launch order = block column order within a chunk), beside
the flops of the column's fat step.  Usage: python scripts/fat_by_column.py kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cls = collections.OrderedDict((("fat", "chol_col_glds_kernel"), ("thin", "chol_col_thin_kernel"),
                               ("diag", "chol_diag"), ("ahead", "diag_ahead")))
seq = {k: [] for k in cls}
for r in rows:
    for k, pat in cls.items():
        if pat in r["Kernel_Name"]:
            seq[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            break
nfat = len(seq["fat"])
per_chunk = 16   # n0 = 2112: 33 block columns, column 0 alone, pairs from column 1
print(f"fat launches {nfat}, thin {len(seq['thin'])}, chol_diag {len(seq['diag'])}, diag_ahead {len(seq['ahead'])}")
nchunk = nfat // per_chunk
by = collections.defaultdict(list)
for i, d in enumerate(seq["fat"]):
    by[i % per_chunk].append(d)
tot = sum(sum(v) for v in by.values())
print("pair  launches  avg_us    share")
for k in sorted(by):
    v = by[k]
    print(f"{k:4d}  {len(v):8d}  {sum(v)/len(v):8.1f}  {sum(v)/tot:6.3f}")
