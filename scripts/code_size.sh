#!/bin/bash
# Device code size (bytes) of the kernels of ngp_kernels.hip matching a pattern (default: the column kernels).
# Usage: bash scripts/code_size.sh [pattern]
set -e
root=$(cd "$(dirname "$0")/.." && pwd); mkdir -p $root/build; cd $root/nowcastautogp_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --offload-device-only -c ngp_kernels.hip -o $root/build/k_dev.o
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$root/build/k_dev.o --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=$root/build/k_gfx950.o
/opt/rocm/lib/llvm/bin/llvm-readelf -s --wide $root/build/k_gfx950.o | grep FUNC | awk '{print $3, $8}' | c++filt | sort -u | grep "${1:-chol_col}" | sed 's/(ngp::JobGeom.*//' | sort -n
