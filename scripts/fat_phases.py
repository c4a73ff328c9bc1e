"""Where the time of a fat step goes, wave by wave (diagnostic; not part of the product).

Runs on the DIAGNOSTIC library: the product's sources linked with scripts/stamps/ngp_stamps.hip,
which instantiates the column-sweep kernels (nowcastautogp_amd/csrc/ngp_col_kernels.h) with a
stamping probe and takes the place of the product's weak NoProbe launchers.  libngp.so never
contains any of it.  This script builds build/libngp_stamps.so itself when it is missing or stale:

    gpurun -- python3 scripts/fat_phases.py 15 16        # block column j, particles (x 200 scenarios)
    gpurun -- python3 scripts/fat_phases.py -17 16       # thin step of column 17
    gpurun -- python3 scripts/fat_phases.py 1016 16      # chol_diag of column 16

Every wave of the fat launch of block column j records 100-MHz timestamps (start, first chunk done,
half of the k-loop, k-loop done, M strips staged, epilogue issued, stores retired) and the CU/SIMD it
ran on; the table goes to gpurun_out/fat_phases_j<j>.npy and a summary to stdout.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAMPS_LIB = os.path.join(ROOT, "build", "libngp_stamps.so")


def build_stamps_lib():
    import subprocess
    csrc = os.path.join(ROOT, "nowcastautogp_amd", "csrc")
    srcs = [os.path.join(csrc, "ngp_kernels.hip"), os.path.join(csrc, "ngp_api.hip"),
            os.path.join(ROOT, "scripts", "stamps", "ngp_stamps.hip")]
    deps = srcs + [os.path.join(csrc, h) for h in ("ngp_col_kernels.h", "ngp_mfma.h", "ngp_internal.h")]
    if os.path.exists(STAMPS_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(STAMPS_LIB)
                                          for d in deps):
        return
    os.makedirs(os.path.dirname(STAMPS_LIB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                           "-fPIC", "-shared", "-o", STAMPS_LIB] + srcs, cwd=csrc)


build_stamps_lib()
os.environ["NGP_LIB"] = STAMPS_LIB
import numpy as np

from nowcastautogp_amd import _lib
from nowcastautogp_amd.synthetic import bench_items

WORDS = 32


def analyse_thin(a, j):
    """thin step of block column j (selected by passing -j): one row tile per wave"""
    t = a[:, 2:9].astype(np.int64)
    t0 = t[:, 0].min()
    t = (t - t0) * 0.01
    tc = ((a[:, 12:24].astype(np.int64) - t0) * 0.01).reshape(-1, 4, 3)
    ok = a[:, 12] != 0
    print(f"thin j={j}: {len(a)} wave records, launch span {t[:, 6].max():.0f} us")
    def med(x):
        return f"median {np.median(x):6.2f}  p10 {np.percentile(x, 10):6.2f}  p90 {np.percentile(x, 90):6.2f} us"
    print("   start -> LDS-DMA + operand stages landed:", med(t[:, 1] - t[:, 0]))
    print("   64-deep product                         :", med(t[:, 2] - t[:, 1]))
    print("   barrier before the epilogue             :", med(t[:, 3] - t[:, 2]))
    prev = t[ok, 3]
    for it in range(4):
        print(f"   pass {it}: K' arrived {np.median(tc[ok, it, 0] - prev):5.2f} | product + LDS "
              f"{np.median(tc[ok, it, 1] - tc[ok, it, 0]):5.2f} | stores issued "
              f"{np.median(tc[ok, it, 2] - tc[ok, it, 1]):5.2f}")
        prev = tc[ok, it, 2]
    print("   last pass -> stores retired             :", med(t[:, 6] - tc[:, 3, 2]))
    print("   whole wave                              :", med(t[:, 6] - t[:, 0]))


def analyse_diag(a, j):
    """chol_diag of block column j (selected by passing 1000 + j): one workgroup per item"""
    t = (a[:, 2:8].astype(np.int64) - int(a[:, 2].min())) * 0.01
    names = ["C_jj = K_jj - L_j L_j' (pending k) -> LDS", "64 x 64 Cholesky (16 rounds)",
             "store L, diagonal-block inverses", "block recursion for M = L^-1", "M strips, logdet"]
    print(f"chol_diag j={j}: {len(a)} workgroups")
    for k, nm in enumerate(names):
        d = t[:, k + 1] - t[:, k]
        print(f"   {nm:>44s}: median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f} us")
    print(f"   {'whole workgroup':>44s}: median {np.median(t[:, 5] - t[:, 0]):6.2f} us")


def analyse(a, j):
    if j >= 1000:
        return analyse_diag(a, j - 1000)
    if j < 0:
        return analyse_thin(a, -j)
    hw, ids = a[:, 0], a[:, 1]
    xcc = (hw >> 32) & 0xF
    h = hw & 0xFFFFFFFF
    simd, cu, sh, se = (h >> 4) & 3, (h >> 8) & 15, (h >> 12) & 1, (h >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    wave = ids & 0xFF
    t = a[:, 2:9].astype(np.int64)
    t0 = t[:, 0].min()
    t = (t - t0) * 0.01            # microseconds
    col = wave & 1
    print(f"j={j}: {len(a)} wave records, {len(np.unique(cuid))} CUs, launch span {t[:, 6].max():.0f} us")
    names = ["start->chunk0", "chunk0->half", "half->k end", "k end->M staged", "M staged->issued",
             "issued->retired"]
    for c in (0, 1):
        m = col == c
        d = np.diff(t[m], axis=1)
        print(f"  column-{c} waves ({m.sum()}):")
        for k, nm in enumerate(names):
            print(f"    {nm:>20s}: median {np.median(d[:, k]):7.2f}  p10 {np.percentile(d[:, k], 10):7.2f}"
                  f"  p90 {np.percentile(d[:, k], 90):7.2f} us")
        tot = t[m, 6] - t[m, 0]
        print(f"    {'whole wave':>20s}: median {np.median(tot):7.2f} us")
    m0 = (col == 0) & (a[:, 12] != 0)
    if m0.any():
        tc = (a[m0, 12:24].astype(np.int64) - t0) * 0.01
        tc = tc.reshape(-1, 4, 3)
        tm = t[m0]
        print("  column-0 epilogue, per 16-row pass (us, median): K' arrived | product + LDS | stores issued")
        prev = tm[:, 4]
        for it in range(4):
            print(f"    pass {it}: {np.median(tc[:, it, 0] - prev):6.2f} | "
                  f"{np.median(tc[:, it, 1] - tc[:, it, 0]):6.2f} | "
                  f"{np.median(tc[:, it, 2] - tc[:, it, 1]):6.2f}")
            prev = tc[:, it, 2]
    # per SIMD: is the k-loop of a wave faster while its partner is outside its own k-loop?
    kbeg, kend, wend = t[:, 1], t[:, 3], t[:, 6]
    key = cuid * 4 + simd
    order = np.argsort(key, kind="stable")
    rate_alone, rate_shared = [], []
    nchunk = 4 * j
    for k in np.unique(key)[:256]:
        idx = order[np.searchsorted(key[order], k, "left"):np.searchsorted(key[order], k, "right")]
        for i in idx:
            # fraction of wave i's k-loop during which some other wave of the SIMD is in its k-loop
            ov = 0.0
            for o in idx:
                if o != i:
                    ov += max(0.0, min(kend[i], kend[o]) - max(kbeg[i], kbeg[o]))
            dur = kend[i] - kbeg[i]
            (rate_shared if ov / dur > 0.9 else rate_alone if ov / dur < 0.6 else []).append(dur)
    if rate_shared:
        print(f"  k-loop (chunk 1..{nchunk}) wall time: partner in its k-loop >90% of it: "
              f"median {np.median(rate_shared):.1f} us ({len(rate_shared)} waves); <60%: "
              f"{np.median(rate_alone) if rate_alone else float('nan'):.1f} us ({len(rate_alone)} waves)")
    # idle estimate per SIMD: time inside the launch span when no resident wave is in a k-loop
    idle = []
    for k in np.unique(key)[:512]:
        idx = order[np.searchsorted(key[order], k, "left"):np.searchsorted(key[order], k, "right")]
        ev = sorted([(t[i, 0], 1) for i in idx] + [(kend[i], -1) for i in idx])
        lo, hi = min(t[i, 0] for i in idx), max(wend[i] for i in idx)
        busy, depth, last = 0.0, 0, lo
        for x, s in ev:
            if depth > 0:
                busy += x - last
            depth += s
            last = x
        idle.append(1.0 - busy / (hi - lo))
    print(f"  share of a SIMD's time with NO resident wave inside its k-loop: median {np.median(idle):.3f}"
          f" (p10 {np.percentile(idle, 10):.3f}, p90 {np.percentile(idle, 90):.3f})")
    resid = []
    for k in np.unique(key)[:512]:
        idx = order[np.searchsorted(key[order], k, "left"):np.searchsorted(key[order], k, "right")]
        lo, hi = min(t[i, 0] for i in idx), max(wend[i] for i in idx)
        resid.append(sum(wend[i] - t[i, 0] for i in idx) / (hi - lo))
    print(f"  resident waves per SIMD (time average): {np.mean(resid):.2f}")


def main():
    j = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    lib = C.CDLL(os.environ["NGP_LIB"])
    ctx = _lib.Context(0)
    scen = int(sys.argv[3]) if len(sys.argv) > 3 else None
    w, progs, Y, tt = bench_items("C3", 0, None, P, scen)
    job = ctx.stage_predict(progs, tt, Y, w.t_new)
    job.run()                                   # warm
    cap = len(progs) * 20 * 4
    assert lib.ngp_dbg_stamps_begin(C.c_int(j), C.c_uint(cap)) == 0
    job.run()
    out = job.fetch()
    buf = np.zeros((cap, WORDS), np.uint64)
    lib.ngp_dbg_stamps_fetch.restype = C.c_long
    n = lib.ngp_dbg_stamps_fetch(buf.ctypes.data_as(C.c_void_p), C.c_uint(cap))
    a = buf[:n]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", f"fat_phases_j{j}.npy"), a)
    print("items", len(progs), "failed", int(np.count_nonzero(out["info"])), "records", n)
    analyse(a, j)


if __name__ == "__main__":
    main()
