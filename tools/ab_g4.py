#!/usr/bin/env python3
"""tools/ab_g4.py -- G4+H4 bank: one 11-plane kernel vs two half launches vs both halves in one launch
(CVS_OPT_G4_SPLIT = 0 / 1 / 2) x strip rows, interleaved rounds in one process."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("CVS_PLACEMENT_SEARCH", "0"); _os.environ.setdefault("CVS_AUTOTUNE", "0")  # A/B runs compare like with like
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
n = 4096
img = torch.rand((n, n), device="cuda")
g, h = torch.empty_like(img), torch.empty_like(img)
def timeit(fn, steps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps
cfgs = [(sp, sr) for sp in (0, 1, 2) for sr in (27, 40, 66, 131)]
hs = {}
for c in cfgs:
    f = cv.SteerableFiltersG4(None); f.set_option(L.OPT_G4_SPLIT, c[0]); f.set_strip_rows(c[1]); hs[c] = f
ref = None
for c in cfgs:   # all variants agree bit for bit
    hs[c].setup(img)
    b = hs[c].basis(7)
    if ref is None: ref = b
    assert torch.equal(b, ref), c
for name, fn, bpp in (("M6 basis", lambda f: f.setup(img), 48), ("M6 +steer", lambda f: f.setup_steer(img, 0.3, out=(g, h)), 56)):
    res = {c: [] for c in cfgs}
    for c in cfgs: timeit(lambda: fn(hs[c]), 5)
    for r in range(8):
        for c in cfgs: res[c].append(timeit(lambda: fn(hs[c])))
    print(name)
    for c in cfgs:
        med = statistics.median(res[c])
        print("   split=%d strip_rows=%3d : %.4f ms  %5.1f%%" % (c[0], c[1], med, bpp * n * n / med / 1e6 / 80), flush=True)
