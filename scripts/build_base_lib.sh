#!/bin/bash
# build/libngp_base.so from the sources of a git revision (default HEAD): the "before" of an A/B
REV=${1:-HEAD}
set -e
cd "$(dirname "$0")/.."
rm -rf build/base_src && mkdir -p build/base_src/include build/base_src/nowcastautogp_amd/csrc
git archive $REV include nowcastautogp_amd/csrc | tar -x -C build/base_src
cd build/base_src/nowcastautogp_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o ../../../libngp_base.so ngp_kernels.hip ngp_api.hip
echo built build/libngp_base.so from $REV
