#!/usr/bin/env python3
"""tools/tuner_value_probe.py -- what the online tuner's decision is worth on the SAME handle (same state block): every leg settles
with the tuner on, then tuner-on and tuner-off (CVS_OPT_AUTOTUNE 0 = the engine's default configuration) take turns, sustained
launches.  PROBE_HANDLES handles per process."""
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
    legs = (("M1", 32, lambda: f.setup(img, flags=cv.SETUP_BASIS)), ("M2", 40, lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))),
            ("M4", 52, lambda: f.setup(img, flags=cv.SETUP_FULL)), ("M5", 84, lambda: f.pipeline(img, out=outs)))
    for name, bpp, fn in legs:
        f.set_option(L.OPT_AUTOTUNE, 1)
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
        li = f.launch_info()
        res = {1: [], 0: []}
        for r in range(3):
            for mode in (1, 0):
                f.set_option(L.OPT_AUTOTUNE, mode)
                fn(); fn()
                res[mode].append(timeit(fn))
        a, b = (bpp * n * n / (statistics.median(res[m]) * 1e-3) / 8e12 for m in (1, 0))
        print("handle %d %s  tuned %.3f | default %.3f  (%+.1f %%)   kept: order %d xcd %d strip %d layout %d wg/cu %d" %
              (hi, name, a, b, 100 * (a / b - 1), li["block_order"], li["xcd_weights"], li["strip_rows"], li["state_layout"], li["wg_per_cu"]), flush=True)
    del f
