#!/usr/bin/env python3
"""tools/tuner_value_probe.py -- what the online tuner's decision is worth on the SAME handle (same state block): every leg calls until
the tuner has decided (cvs_launch_info.tune_state), then tuner-on and tuner-off (CVS_OPT_AUTOTUNE 0 = the engine's default configuration)
take turns, sustained launches of 150.  Run it in many processes (the tuner's memory is per process): a pick that is more than 1 %
slower sustained than the default is a wrong decision.  PROBE_HANDLES handles per process (the later ones inherit the decisions)."""
import os, sys, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
n = 4096
img = torch.rand((n, n), device="cuda")
g, h = cv.alloc_planes(2, n, n, device="cuda")
outs = cv.alloc_planes(8, n, n, device="cuda")


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


for hi in range(int(os.environ.get("PROBE_HANDLES", "3"))):
    f = cv.SteerableFiltersG2(None)
    f4 = cv.SteerableFiltersG4(None)
    small = img[:1536, :2048].contiguous()
    legs = (("M4", 52, n * n, f, lambda: f.setup(img, flags=cv.SETUP_FULL)), ("M5", 84, n * n, f, lambda: f.pipeline(img, out=outs)),
            ("G4", 48, n * n, f4, lambda: f4.setup(img)), ("M4 1536x2048", 52, 1536 * 2048, f, lambda: f.setup(small, flags=cv.SETUP_FULL)))
    for name, bpp, npx, f, fn in legs:
        f.set_option(L.OPT_AUTOTUNE, 1)
        calls = 0
        for _ in range(60):          # the tuner says when it has decided
            for _ in range(5):
                fn()
            calls += 5
            torch.cuda.synchronize()     # (the tuner reads its samples back when their launches have finished, never by waiting)
            if f.launch_info()["tune_state"] != 1:
                break
        torch.cuda.synchronize()
        li = f.launch_info()
        res = {1: [], 0: []}
        for r in range(3):
            for mode in (1, 0):
                f.set_option(L.OPT_AUTOTUNE, mode)
                fn(); fn()
                res[mode].append(timeit(fn))
        a, b = (bpp * npx / (statistics.median(res[m]) * 1e-3) / 8e12 for m in (1, 0))
        print("handle %d %-13s tuned %.3f | default %.3f  (%+.1f %%)  decided after %3d calls (state %d): challenger kept %d -- order %d strip %d layout %d wg %d" %
              (hi, name, a, b, 100 * (a / b - 1), calls, li["tune_state"], li["tuned"], li["block_order"], li["strip_rows"], li["state_layout"], li["wg_per_cu"]), flush=True)
    del f, f4
