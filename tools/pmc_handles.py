#!/usr/bin/env python3
"""tools/pmc_handles.py -- run under `rocprofv3 --pmc FETCH_SIZE` (or WRITE_SIZE ...): 10 handles, each with its own
state allocation, 5 launches of the 12-plane setup per handle, handle after handle.  Timing is measured first,
WITHOUT meaning under the profiler; the point is whether HBM traffic differs from allocation to allocation."""
import os, sys
os.environ["CVS_PLACEMENT_SEARCH"] = "0"
os.environ["CVS_AUTOTUNE"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

n = 4096
img = torch.rand((n, n), device="cuda")
hs = []
for i in range(10):
    f = cv.SteerableFiltersG2(None)
    f.set_option(L.OPT_BLOCK_ORDER, 0)
    f.setup(img, flags=cv.SETUP_FULL)
    hs.append(f)
torch.cuda.synchronize()
for f in hs:
    for _ in range(5):
        f.setup(img, flags=cv.SETUP_FULL)
    torch.cuda.synchronize()
