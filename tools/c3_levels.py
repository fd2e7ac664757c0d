#!/usr/bin/env python3
"""tools/c3_levels.py -- BASELINE config 3 under the microscope: the five levels of the 8192^2 pyramid filtered one
after the other (G2+H2 basis, 32 B/pix), time per level inside the sequence (events between the launches), for the
engine's defaults and for pinned strip heights / launch orders."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
big = torch.rand((8192, 8192), device="cuda")
f0 = cv.SteerableFiltersG2(None)
lv = f0.pyramid(big, 5)
pix = [l.shape[0] * l.shape[1] for l in lv]
def measure(tag, opts):
    hs = [cv.SteerableFiltersG2(None) for _ in lv]
    for h in hs:
        for o, v in opts: h.set_option(o, v)
    def seq(evs=None):
        for i, (h, l) in enumerate(zip(hs, lv)):
            if evs: evs[i].record()
            h.setup(l, flags=cv.SETUP_BASIS)
        if evs: evs[len(lv)].record()
    for _ in range(6): seq()
    torch.cuda.synchronize()
    acc = [0.0] * len(lv); tot = 0.0; reps = 20
    for _ in range(reps):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(lv) + 1)]
        seq(evs); torch.cuda.synchronize()
        for i in range(len(lv)): acc[i] += evs[i].elapsed_time(evs[i + 1]) / reps
        tot += evs[0].elapsed_time(evs[len(lv)]) / reps
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): seq()
    b.record(); torch.cuda.synchronize()
    free = a.elapsed_time(b) / reps
    print("%-34s total %.4f ms (%.1f%%; without per-level events %.4f ms = %.1f%%)  levels: %s" % (
        tag, tot, 32 * sum(pix) / tot / 1e6 / 80, free, 32 * sum(pix) / free / 1e6 / 80,
        "  ".join("%.4f (%.0f%%)" % (t, 32 * p / t / 1e6 / 80) for t, p in zip(acc, pix))), flush=True)
measure("defaults", [])
measure("autotune off", [(L.OPT_AUTOTUNE, 0)])
for sr in (10, 19, 28):
    for order in (0, 1):
        measure("strip rows %d, order %d, no tune" % (sr, order), [(L.OPT_AUTOTUNE, 0), (L.OPT_STRIP_ROWS, sr), (L.OPT_BLOCK_ORDER, order)])
