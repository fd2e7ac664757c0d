#!/usr/bin/env python3
"""tools/occ_probe.py -- workgroups per CU (CVS_OPT_WG_PER_CU) side by side on ONE handle, sustained launches, interleaved rounds: the
cap a caller may pin (the engine never does: DESIGN.md section 7).  Tuner off; 0 = as many as the registers allow."""
import os, sys, statistics
os.environ["CVS_AUTOTUNE"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
n = int(os.environ.get("OCC_N", "4096"))
gen = torch.Generator(device="cuda").manual_seed(3)
imgs = [torch.rand((n, n), device="cuda", generator=gen) for _ in range(8)]
g, h = cv.alloc_planes(2, n, n, device="cuda")
outs = cv.alloc_planes(8, n, n, device="cuda")
rot = {"i": 0}


def timeit(fn, steps=150):
    for _ in range(steps // 3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def nxt():
    rot["i"] = (rot["i"] + 1) & 7
    return imgs[rot["i"]]


caps = [0, 4, 3, 2]
for hi in range(int(os.environ.get("OCC_HANDLES", "2"))):
    f = cv.SteerableFiltersG2(None)
    f4 = cv.SteerableFiltersG4(None)
    legs = (("M1 basis", 32, f, lambda: f.setup(imgs[0], flags=cv.SETUP_BASIS)), ("M2 +steer", 40, f, lambda: f.setup_steer(imgs[0], 0.3, flags=cv.SETUP_BASIS, out=(g, h))),
            ("M4 full", 52, f, lambda: f.setup(imgs[0], flags=cv.SETUP_FULL)), ("M5 pipeline", 84, f, lambda: f.pipeline(imgs[0], out=outs)),
            ("M2 new images", 40, f, lambda: f.setup_steer(nxt(), 0.3, flags=cv.SETUP_BASIS, out=(g, h))), ("M5 new images", 84, f, lambda: f.pipeline(nxt(), out=outs)),
            ("G4 basis", 48, f4, lambda: f4.setup(imgs[0])))
    for name, bpp, hnd, fn in legs:
        res = {c: [] for c in caps}
        for r in range(3):
            for c in caps:
                hnd.set_option(L.OPT_WG_PER_CU, c)
                res[c].append(timeit(fn))
        hnd.set_option(L.OPT_WG_PER_CU, 0)
        print("handle %d %-14s " % (hi, name) + " | ".join("%s %.3f" % ("uncapped" if c == 0 else "%d WG/CU" % c, bpp * n * n / (statistics.median(v) * 1e-3) / 8e12) for c, v in res.items()), flush=True)
