#!/usr/bin/env python3
"""tools/ab_place.py -- CVS_OPT_PLACEMENT_SEARCH 0 (plain hipMalloc state block) vs 1 (one physical allocation per
plane, dealt across a run boundary when the allocation-time probe finds one): the many-plane legs on one resident
4096x4096 image and on 8 rotating images, interleaved rounds, autotune off.  CVS_TUNE_VERBOSE=1 shows the probe."""
import os, sys, statistics, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CVS_AUTOTUNE", "0")
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
n = int(os.environ.get("AB_N", "4096"))
imgs = [torch.rand((n, n), device="cuda") for _ in range(8)]
g, h = torch.empty_like(imgs[0]), torch.empty_like(imgs[0])
outs8 = [torch.empty_like(imgs[0]) for _ in range(8)]
def timeit(fn, steps=24):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(steps): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps
hs = {}
for mode in (0, 1, 0, 1):
    f = cv.SteerableFiltersG2(None); f.set_option(L.OPT_PLACEMENT_SEARCH, mode)
    t0 = time.perf_counter(); f.setup(imgs[0]); torch.cuda.synchronize()
    print("placement %d: first setup (allocation%s) %.1f ms" % (mode, " + probe" if mode else "", (time.perf_counter() - t0) * 1e3), flush=True)
    hs[(mode, len(hs))] = f
legs = (("M1 basis", 32, lambda f: (lambda i: f.setup(imgs[0], flags=cv.SETUP_BASIS))),
        ("M2 filter+steer", 40, lambda f: (lambda i: f.setup_steer(imgs[0], 0.3, flags=cv.SETUP_BASIS, out=(g, h)))),
        ("M2 rotating inputs", 40, lambda f: (lambda i: f.setup_steer(imgs[i & 7], 0.3, flags=cv.SETUP_BASIS, out=(g, h)))),
        ("M4 full setup", 52, lambda f: (lambda i: f.setup(imgs[0], flags=cv.SETUP_FULL))),
        ("M5 pipeline", 84, lambda f: (lambda i: f.pipeline(imgs[0], out=outs8))))
for name, bpp, mk in legs:
    res = {k: [] for k in hs}
    for k in hs: timeit(mk(hs[k]), 6)
    for r in range(5):
        for k in hs: res[k].append(timeit(mk(hs[k])))
    print("%-20s" % name + "  ".join("placement %d: %.4f ms %4.1f%%" % (k[0], statistics.median(res[k]), bpp * n * n / statistics.median(res[k]) / 1e6 / 80) for k in hs), flush=True)
