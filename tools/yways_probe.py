#!/usr/bin/env python3
"""tools/yways_probe.py -- needs tools/patches/band_interleave_probe.patch (CVS_Y_WAYS): row bands of ONE image dispatched dealt from 2 / 4 / 8 equal parts of the
image in turn, against the plain order; plain state blocks (the library default), several handles, interleaved rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVS_AUTOTUNE"] = "0"
os.environ["CVS_PLACEMENT_SEARCH"] = "0"
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

n = 4096
imgs = [torch.rand((n, n), device="cuda") for _ in range(8)]
g, h = torch.empty_like(imgs[0]), torch.empty_like(imgs[0])
outs8 = [torch.empty_like(imgs[0]) for _ in range(8)]


def timeit(fn, steps=20, warm=3):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(steps):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


ways = [1, 2, 4, 8]
keep = []
for hnd in range(4):
    f = cv.SteerableFiltersG2(None)
    keep.append(f)
    for order in (0, 1):
        f.set_option(L.OPT_BLOCK_ORDER, order)
        legs = (("M2 resident", 40, lambda i: f.setup_steer(imgs[0], 0.3, flags=cv.SETUP_BASIS, out=(g, h))),
                ("M2 rotating", 40, lambda i: f.setup_steer(imgs[i & 7], 0.3, flags=cv.SETUP_BASIS, out=(g, h))),
                ("M4 full", 52, lambda i: f.setup(imgs[0], flags=cv.SETUP_FULL)),
                ("M5 pipeline", 84, lambda i: f.pipeline(imgs[0], out=outs8)))
        for name, bpp, fn in legs:
            res = {w: [] for w in ways}
            for rnd in range(3):
                for w in ways:
                    os.environ["CVS_Y_WAYS"] = str(w)
                    res[w].append(timeit(fn))
            print("handle %d order %d %-12s " % (hnd, order, name) + " | ".join("ways %d %.3f" % (w, bpp * n * n / sorted(res[w])[1] / 1e6 / 8000) for w in ways), flush=True)
