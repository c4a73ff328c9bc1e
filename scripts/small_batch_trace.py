"""The launch timeline of a small-batch call (the 64-particle logml / logml + gradient call a fit
repeats): run under `rocprofv3 --kernel-trace`, then `--analyse <kernel_trace.csv>` prints, for the
last call, every launch with its start offset, duration and the idle gap before it.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sbt -- \
        python3 $R/scripts/small_batch_trace.py logml 2048 64
    python3 scripts/small_batch_trace.py --analyse gpurun_out/sbt/**/*kernel_trace.csv
"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def analyse(path, verbose=True):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows]
    # calls are separated by long gaps (host work between calls)
    calls, cur = [], [ev[0]]
    for a, b in zip(ev, ev[1:]):
        if b[0] - a[1] > 300_000:      # > 0.3 ms idle: next call
            calls.append(cur)
            cur = []
        cur.append(b)
    calls.append(cur)
    last = calls[-1]
    t0 = last[0][0]
    busy = sum(e - s for s, e, _ in last)
    span = max(e for _, e, _ in last) - t0
    by = {}
    prev_end = t0
    for s, e, k in last:
        name = k.split("::")[-1][:40]
        g = by.setdefault(name, [0, 0.0, 0.0])
        g[0] += 1
        g[1] += (e - s) / 1e3
        g[2] += max(0, s - prev_end) / 1e3
        if verbose:
            print(f"{(s - t0) / 1e3:9.1f} us  +{max(0, s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {name}")
        prev_end = max(prev_end, e)
    print(f"calls seen {len(calls)}; last call: {len(last)} launches, span {span / 1e3:.1f} us, "
          f"sum of kernel time {busy / 1e3:.1f} us")
    for name, (n, t, gp) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"  {name:42s} {n:4d} launches {t:9.1f} us  (gaps before them {gp:7.1f} us)")


if __name__ == "__main__":
    if sys.argv[1] == "--analyse":
        analyse(sys.argv[2], verbose="-q" not in sys.argv)
        sys.exit(0)
    import __graft_entry__ as ge
    ge.build()
    from nowcastautogp_amd import _lib
    from nowcastautogp_amd.synthetic import make_workload
    kind, n, P = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    ctx = _lib.Context(0)
    w = make_workload("C3", n=n, P=P)
    import time
    for _ in range(4):
        if kind == "grad":
            ctx.logml_grad_batch(w.programs, w.t, w.y)
        else:
            ctx.logml_batch(w.programs, w.t, w.y)
        time.sleep(0.01)
    print("done")
