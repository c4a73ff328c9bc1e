#!/bin/bash
# Build libngp of an older revision into build/libngp_a.so (the "a" side of scripts/gpu_ab.sh / gpu_abc.sh).
# Usage: bash scripts/build_baseline.sh <git-revision>       (runs here, without a GPU)
set -e
rev=${1:?usage: build_baseline.sh <git-revision>}
root=$(cd "$(dirname "$0")/.." && pwd)
d=$root/build/src_$rev
rm -rf "$d"; mkdir -p "$d/nowcastautogp_amd/csrc" "$d/include"
for f in ngp_kernels.hip ngp_api.hip ngp_col_kernels.h ngp_internal.h ngp_mfma.h; do
  git -C "$root" show "$rev:nowcastautogp_amd/csrc/$f" > "$d/nowcastautogp_amd/csrc/$f"
done
git -C "$root" show "$rev:include/ngp.h" > "$d/include/ngp.h"
(cd "$d/nowcastautogp_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o "$root/build/libngp_a.so" ngp_kernels.hip ngp_api.hip)
rm -rf "$d"
echo "build/libngp_a.so = $rev (its C-ABI must have every symbol nowcastautogp_amd/_lib.py binds)"
