#!/usr/bin/env python3
"""tools/pyr_probe.py -- what the fused pyramid emission costs on the 8192^2 level of config 3: the basis pass alone against the
basis pass that also writes the next level (cvs_setup_pyr), two images alternating (fresh inputs), a few launch configurations"""
import os, sys, statistics
os.environ.setdefault("CVS_AUTOTUNE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
n = int(os.environ.get("PYR_N", "8192"))
imgs = [torch.rand((n, n), device="cuda") for _ in range(2)]
nxt = torch.empty((n // 2, n // 2), device="cuda")
f = cv.SteerableFiltersG2(None)
flip = {"i": 0}


def timeit(fn, steps=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def plain():
    flip["i"] ^= 1
    f.setup(imgs[flip["i"]], flags=cv.SETUP_BASIS)


def pyr():
    flip["i"] ^= 1
    f.setup_pyr(imgs[flip["i"]], flags=cv.SETUP_BASIS, out=nxt)


for name, (o, sr, xw) in {"plain10": (0, 10, 0), "plain19": (0, 19, 0), "xcdcol10": (1000000, 10, 101), "dyn10": (2000000, 10, 0), "w504_10": (1, 10, 504)}.items():
    f.set_option(L.OPT_BLOCK_ORDER, o); f.set_option(L.OPT_STRIP_ROWS, sr); f.set_option(L.OPT_XCD_WEIGHTS, xw)
    res = {"basis": [], "basis+pyr": []}
    for fn in (plain, pyr):
        fn(); fn()
    for r in range(5):
        res["basis"].append(timeit(plain))
        res["basis+pyr"].append(timeit(pyr))
    tb, tp = statistics.median(res["basis"]), statistics.median(res["basis+pyr"])
    print("%-9s basis %.4f ms %.3f | basis + next level %.4f ms %.3f  (+%.1f %%)" % (name, tb, 32 * n * n / tb / 1e6 / 8000, tp, 33 * n * n / tp / 1e6 / 8000, 100 * (tp / tb - 1)), flush=True)
