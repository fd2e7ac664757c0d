#!/usr/bin/env python3
"""tools/ab_rot.py -- M2 with rotating inputs (input read from HBM, not the Infinity Cache) vs strip rows."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("CVS_PLACEMENT_SEARCH", "0"); _os.environ.setdefault("CVS_AUTOTUNE", "0")  # A/B runs compare like with like
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
n = 4096
imgs = [torch.rand((n, n), device="cuda") for _ in range(8)]
g, h = torch.empty_like(imgs[0]), torch.empty_like(imgs[0])
def timeit(fn, steps=24):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(steps): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps
vals = [10, 19, 28, 37, 64, 127]
hs = {}
for v in vals:
    f = cv.SteerableFiltersG2(None); f.set_strip_rows(v); hs[v] = f
for name, fn in (("rotating 8 inputs", lambda f: (lambda i: f.setup_steer(imgs[i & 7], 0.3, flags=cv.SETUP_BASIS, out=(g, h)))),
                 ("single input", lambda f: (lambda i: f.setup_steer(imgs[0], 0.3, flags=cv.SETUP_BASIS, out=(g, h))))):
    res = {v: [] for v in vals}
    for v in vals: timeit(fn(hs[v]), 8)
    for r in range(8):
        for v in vals: res[v].append(timeit(fn(hs[v])))
    print(name + ": " + " | ".join("sr=%d %.4f ms %4.1f%%" % (v, statistics.median(res[v]), 40 * n * n / statistics.median(res[v]) / 1e6 / 80) for v in vals), flush=True)
