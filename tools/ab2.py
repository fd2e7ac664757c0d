#!/usr/bin/env python3
"""tools/ab2.py -- how much does the relative placement of output planes matter?  state-plane pad x
stagger between caller-owned output planes (carved from one allocation), interleaved rounds."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

def timeit(fn, steps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps

n = 4096
img = torch.rand((n, n), device="cuda")
cfgs = []
for pad in (0, 64, 2112):
    for stag in (0, 64, 2112, 4160):
        big = torch.empty(8 * (n * n + stag) + 64, device="cuda")
        outs = [big[k * (n * n + stag): k * (n * n + stag) + n * n].view(n, n) for k in range(8)]
        f = cv.SteerableFiltersG2(None)
        f.set_option(L.OPT_PLANE_PAD, pad)
        cfgs.append((pad, stag, f, outs, big))
legs = {
    "M2 +steer": (lambda f, o: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(o[0], o[1])), 40),
    "M5 pipeline": (lambda f, o: f.pipeline(img, out=o), 84),
    "M3 steer scalar": (lambda f, o: f.steer(0.3, out=(o[0], o[1])), 36),
}
for name, (fn, bpp) in legs.items():
    res = {i: [] for i in range(len(cfgs))}
    for i, (pad, stag, f, outs, _) in enumerate(cfgs):
        f.setup(img, flags=cv.SETUP_FULL)
        for _ in range(3): fn(f, outs)
    torch.cuda.synchronize()
    for r in range(8):
        for i, (pad, stag, f, outs, _) in enumerate(cfgs):
            if name.startswith("M3"): pass
            res[i].append(timeit(lambda: fn(f, outs), 10))
    print(name)
    for i, (pad, stag, f, outs, _) in enumerate(cfgs):
        med = statistics.median(res[i])
        print("   state pad %5d  out stagger %5d : %.4f ms  %5.1f%%" % (pad, stag, med, bpp * n * n / med / 1e6 / 80), flush=True)
