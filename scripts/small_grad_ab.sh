#!/bin/bash
# The 64-particle logml + gradient call of a fit at long series, working tree against build/libngp_base.so.
# Usage: gpurun -- bash scripts/small_grad_ab.sh
R=$GRAFT_REPO_ROOT
cat > /tmp/sg.py <<'PY'
import sys, time, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from nowcastautogp_amd import _lib
from nowcastautogp_amd.synthetic import make_workload
ctx = _lib.Context(0)
for n, P in ((512, 64), (1024, 64), (2048, 64), (2048, 256)):
    for ens in ("prior", "fitted"):
        w = make_workload("C3", n=n, P=P, D=1, ensemble=ens)
        ka = _lib.KernelArray(w.programs)
        fn = lambda: ctx.logml_grad_flat(ka, w.t, w.y)
        for _ in range(3): fn()
        N = 20
        t0 = time.perf_counter()
        for _ in range(N): fn()
        print("n %5d  P %4d  %-6s  %.3f ms per logml + gradient call" % (n, P, ens, (time.perf_counter() - t0) / N * 1e3))
PY
for which in base new base new; do
  if [ $which = base ]; then export NGP_LIB=$R/build/libngp_base.so; else unset NGP_LIB; fi
  echo "== $which"; python3 /tmp/sg.py
done
