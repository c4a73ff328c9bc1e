#!/usr/bin/env python3
"""Which librccl / libamdhip64 / libhsa-runtime64 end up in the process when ngp_comm_create opens
RCCL at run time, with torch imported before or after libngp.  Usage: python scripts/rccl_probe.py [torch-first|torch-after|no-torch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "no-torch"
if mode == "torch-first":
    import torch  # noqa
from nowcastautogp_amd import _lib
ctx = _lib.Context(0)
if mode == "torch-after":
    import torch  # noqa
try:
    uid = _lib.comm_unique_id()
    comm = _lib.Comm(ctx, uid, 0, 1)
    print(mode, "comm ok")
    comm.close()
except Exception as e:
    print(mode, "FAILED:", e)
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if any(k in l for k in ("rccl", "amdhip64", "hsa-runtime"))})
for l in libs:
    print("   ", l)
