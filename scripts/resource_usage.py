#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of ngp_kernels.hip, one line per kernel
(hipcc -Rpass-analysis=kernel-resource-usage; runs without a GPU).  Usage:
    python scripts/resource_usage.py [filter-substring ...] > profiles/rNN/resource_usage_kernels.txt"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nowcastautogp_amd", "csrc")


def main():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "ngp_kernels.hip",
           "-o", os.path.join(ROOT, "build", "ngp_kernels_ru.o"), "-Rpass-analysis=kernel-resource-usage"]
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    err = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    filt = sys.argv[1:]
    print("# " + " ".join(cmd[:6] + cmd[8:]))
    cur = {}
    for line in err.splitlines():
        m = re.search(r"remark: (.*)", line)
        if not m:
            continue
        t = m.group(1).replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
        for key, tag in (("VGPRs:", "v"), ("AGPRs:", "a"), ("ScratchSize [bytes/lane]:", "s"),
                         ("Occupancy [waves/SIMD]:", "o"), ("LDS Size [bytes/block]:", "l")):
            if t.startswith(key):
                cur[tag] = t.split(":")[-1].strip()
        if "l" in cur and "name" in cur:
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True,
                                  text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name).replace("void ", "").replace("ngp::", "")
            if not filt or any(f in name for f in filt):
                print(f"{name:<72} VGPRs {cur.get('v','?'):>4}  AGPRs {cur.get('a','?'):>3}  scratch B/lane "
                      f"{cur.get('s','?'):>5}  waves/SIMD {cur.get('o','?'):>2}  LDS B/block {cur.get('l','?'):>6}")
            cur = {}


if __name__ == "__main__":
    main()
