#!/usr/bin/env python3
"""tools/dyn_trace.py -- 30 launches of the basis pass in the plain order, then 30 in the dynamic order (for rocprofv3 --kernel-trace)"""
import os, sys
os.environ.setdefault("CVS_AUTOTUNE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
img = torch.rand((4096, 4096), device="cuda")
f = cv.SteerableFiltersG2(None)
f.set_option(L.OPT_STRIP_ROWS, 10)
for order in (0, 2000000, 0, 2000000):
    f.set_option(L.OPT_BLOCK_ORDER, order)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        f.setup(img, flags=cv.SETUP_BASIS)
    a.record()
    for _ in range(30):
        f.setup(img, flags=cv.SETUP_BASIS)
    b.record()
    torch.cuda.synchronize()
    print("order %d: %.4f ms per launch (events)" % (order, a.elapsed_time(b) / 30), flush=True)
