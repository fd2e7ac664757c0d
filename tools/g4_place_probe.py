#!/usr/bin/env python3
"""tools/g4_place_probe.py -- does the opt-in placement search help the G4 bank?  Plain block vs searched window, several handles,
one process; CVS_TUNE_VERBOSE=1 shows the windows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cvsteer_amd as cv
from cvsteer_amd import _lib as L
def timeit(fn, steps=20, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps
n = 4096
img = torch.rand((n, n), device="cuda")
g, h = torch.empty_like(img), torch.empty_like(img)
hs = []
for mode in (0, 1, 0, 1):
    f = cv.SteerableFiltersG4(None)
    f.set_option(L.OPT_PLACEMENT_SEARCH, mode)
    f.set_option(L.OPT_AUTOTUNE, 0)
    f.setup(img)
    hs.append((mode, f))
for name, bpp, fn in (("basis", 48, lambda f: f.setup(img)), ("basis+steer", 56, lambda f: f.setup_steer(img, 0.3, out=(g, h)))):
    res = [[] for _ in hs]
    for rnd in range(4):
        for k, (mode, f) in enumerate(hs):
            res[k].append(timeit(lambda: fn(f)))
    for k, (mode, f) in enumerate(hs):
        ms = sorted(res[k])[len(res[k]) // 2]
        print("G4 %-11s handle %d placement %d (window %s): %.4f ms  %.3f of HBM" % (name, k, mode, f.launch_info()["window_found"], ms, bpp * n * n / ms / 1e6 / 8000), flush=True)
