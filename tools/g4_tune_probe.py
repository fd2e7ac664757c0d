#!/usr/bin/env python3
"""tools/g4_tune_probe.py -- what the online tuner sees for the G4 bank (CVS_TUNE_VERBOSE=1) next to a direct A/B of the same
candidates on the same handle, timed as 20 launches back to back."""
import os, sys
os.environ["CVS_TUNE_VERBOSE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

img = torch.rand((4096, 4096), device="cuda")
g, h = cv.alloc_planes(2, 4096, 4096, device="cuda")


def timeit(fn, steps=20, reps=5):
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / steps)
    return sorted(out)[len(out) // 2]


for trial in range(3):
    f4 = cv.SteerableFiltersG4(None, 6, 0.5)
    for name, fn, bpp in (("basis", lambda: f4.setup(img), 48), ("+steer", lambda: f4.setup_steer(img, 0.3, out=(g, h)), 56)):
        for _ in range(80):
            fn()
        torch.cuda.synchronize()
        li = f4.launch_info()
        t = timeit(fn)
        print("trial %d %s: tuner kept order %d strip %d split %d -> %.4f ms (%.3f)" % (trial, name, li["block_order"], li["strip_rows"], li["g4_split"], t, bpp * 4096 * 4096 / (t * 1e-3) / 8e12), flush=True)
    fa = cv.SteerableFiltersG4(None, 6, 0.5)
    fa.set_option(L.OPT_AUTOTUNE, 0)
    res = []
    for order in (0, 2000000, 1):
        fa.set_option(L.OPT_BLOCK_ORDER, order)
        for _ in range(5):
            fa.setup(img)
        t = timeit(lambda: fa.setup(img))
        res.append("order %d %.4f ms (%.3f)" % (order, t, 48 * 4096 * 4096 / (t * 1e-3) / 8e12))
    print("   pinned on a second handle: " + " | ".join(res), flush=True)
    del f4, fa
