#!/bin/bash
# Three configurations interleaved on ONE box: libngp_a.so, libngp.so, libngp.so with an option.
# Usage: gpurun -- bash scripts/gpu_abc.sh --no-structured-storage [extra bench args]
opt=$1; shift
fmt='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"],1), {k:round(v,1) for k,v in d["kernels_ms_per_step"].items()})'
for r in 1 2; do
  echo "== a (libngp_a.so) round $r"
  NGP_LIB=$GRAFT_REPO_ROOT/build/libngp_a.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fit --headline-only "$@" 2>/dev/null | python -c "$fmt"
  echo "== b (libngp.so) round $r"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fit --headline-only "$@" 2>/dev/null | python -c "$fmt"
  echo "== b $opt round $r"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fit --headline-only $opt "$@" 2>/dev/null | python -c "$fmt"
done
