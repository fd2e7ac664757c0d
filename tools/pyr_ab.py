#!/usr/bin/env python3
"""tools/pyr_ab.py -- cvs_pyr_down: strip march (CVS_PYR_STRIP=1, default) against the stand-alone kernel (0); the 4 levels below 8192^2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cvsteer_amd as cv
big = torch.rand((8192, 8192), device="cuda")
f = cv.SteerableFiltersG2(None)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
lv = f.pyramid(big, 5)
print("CVS_PYR_STRIP=%s: 5-level pyramid of 8192^2: %.4f ms; level 0 -> 1 alone: %.4f ms" % (os.environ.get("CVS_PYR_STRIP", "1"), t(lambda: f.pyramid(big, 5)), t(lambda: f.pyrDown(big))))
