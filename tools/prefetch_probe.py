#!/usr/bin/env python3
"""tools/prefetch_probe.py -- does a pure-read pass over a fresh image (pulling it into the Infinity Cache) pay for itself?
8 rotating 4096^2 images; per step: [optional read pass (torch sum)] + fused filter + steer.  Events around the whole loop."""
import os, sys, statistics
os.environ.setdefault("CVS_AUTOTUNE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
n = 4096
imgs = [torch.rand((n, n), device="cuda") for _ in range(8)]
g, h = torch.empty_like(imgs[0]), torch.empty_like(imgs[0])
acc = torch.zeros((), device="cuda")
f = cv.SteerableFiltersG2(None)
f.set_option(L.OPT_STRIP_ROWS, 10)
rot = {"i": 0}


def timeit(fn, steps=24):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def step(prefetch, basis_only):
    rot["i"] = (rot["i"] + 1) & 7
    im = imgs[rot["i"]]
    if prefetch:
        im.max()
    if basis_only:
        f.setup(im, flags=cv.SETUP_BASIS)
    else:
        f.setup_steer(im, 0.3, flags=cv.SETUP_BASIS, out=(g, h))


for basis_only, bpp, name in ((False, 40, "M2"), (True, 32, "M1")):
    res = {0: [], 1: [], 2: []}
    for r in range(5):
        for pf in (0, 1):
            step(pf, basis_only); step(pf, basis_only)
            res[pf].append(timeit(lambda: step(pf, basis_only)))
        res[2].append(timeit(lambda: imgs[rot["i"]].max()))
    t0, t1, ts = (statistics.median(res[k]) for k in (0, 1, 2))
    print("%s rotating: filter alone %.4f ms (%.3f) | read pass + filter %.4f ms (%.3f) | the read pass alone on a resident image %.4f ms" %
          (name, t0, bpp * n * n / t0 / 1e6 / 8000, t1, bpp * n * n / t1 / 1e6 / 8000, ts), flush=True)
