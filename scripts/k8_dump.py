#!/usr/bin/env python3
"""Results of a few calls that run the fat steps of the column sweep (value and gradient geometry, a
ragged tail, a small batch with split-k), written to an .npz: run once per build (NGP_LIB selects the
library) and compare the files bit for bit.  Usage: python scripts/k8_dump.py OUT.npz"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nowcastautogp_amd import _lib
from nowcastautogp_amd.synthetic import make_workload

ctx = _lib.Context(0)
out = {}
for tag, n, P, D in (("a", 2048, 64, 4), ("b", 1111, 40, 3), ("c", 520, 700, 1)):
    w = make_workload("C3", n=n, P=P, D=D, d=1, m=9)
    r = ctx.nowcast_batch(w.programs, w.t, w.y, w.t_add, w.y_add, w.t_new)
    for k in ("logml_base", "logml_full", "mu", "sigma", "info"):
        out[f"{tag}_now_{k}"] = np.asarray(r[k])
    for ens in ("prior", "fitted"):
        wg = make_workload("C3", n=n, P=min(P, 96), D=1, d=1, m=9, ensemble=ens)
        lm, grad, info = ctx.logml_grad_flat(_lib.KernelArray(wg.programs), wg.t, wg.y)
        out[f"{tag}_{ens}_lm"], out[f"{tag}_{ens}_grad"], out[f"{tag}_{ens}_info"] = lm, grad, info
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], {k: v.shape for k, v in list(out.items())[:4]})
