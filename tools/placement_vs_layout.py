#!/usr/bin/env python3
"""tools/placement_vs_layout.py -- the opt-in allocation-time placement search (CVS_OPT_PLACEMENT_SEARCH = 1: one physical allocation
per plane, planar planes on a probed window; rounds 2-3) against the round-4 default (plain block, row-interleaved plane groups), same
process, same image and outputs, sustained launches; tuner off on both (engine defaults)."""
import os, sys, statistics
os.environ["CVS_AUTOTUNE"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

n = 4096
img = torch.rand((n, n), device="cuda")
g, h = cv.alloc_planes(2, n, n, device="cuda")
outs = cv.alloc_planes(8, n, n, device="cuda")


def timeit(fn, steps=200):
    for _ in range(steps // 3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


hs = {}
for name, placement in (("default (interleaved, plain block)", 0), ("placement search (planar, window)", 1), ("default, second handle", 0)):
    f = cv.SteerableFiltersG2(None)
    f.set_option(L.OPT_PLACEMENT_SEARCH, placement)
    f.setup(img, flags=cv.SETUP_FULL)
    hs[name] = f
for leg, bpp, call in (("M1", 32, lambda f: f.setup(img, flags=cv.SETUP_BASIS)), ("M2", 40, lambda f: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))),
                       ("M4", 52, lambda f: f.setup(img, flags=cv.SETUP_FULL)), ("M5", 84, lambda f: f.pipeline(img, out=outs))):
    res = {k: [] for k in hs}
    for r in range(3):
        for k, f in hs.items():
            res[k].append(timeit(lambda: call(f)))
    print("%s  " % leg + " | ".join("%s %.3f" % (k, bpp * n * n / (statistics.median(v) * 1e-3) / 8e12) for k, v in res.items()), flush=True)
for k, f in hs.items():
    li = f.launch_info()
    print("   %s: placement_mode %d window_found %d probe_ms %.1f state_layout %d" % (k, li["placement_mode"], li["window_found"], li["probe_ms"], li["state_layout"]))
