#!/bin/bash
# A/B of ONE build with and without a bench.py option, interleaved on the SAME box.
# Usage: gpurun -- bash scripts/gpu_ab_opt.sh --no-structured-storage [extra bench args]
opt=$1; shift
for r in 1 2; do
for v in with without; do
  o=$opt; [ $v = without ] && o=
  echo "== $v $opt round $r"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fit --headline-only $o "$@" 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernels_ms_per_step'].items()}, 'frac', round(d['roofline']['frac'],4), 'whole', round(d['roofline'].get('whole_path_frac',0),4))"
done; done
