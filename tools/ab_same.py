#!/usr/bin/env python3
"""tools/ab_same.py -- interleaved A/B of launch options on ONE handle (one state allocation), so that the
allocation-dependent speed modes (tools/alloc_modes.py) cannot masquerade as an effect of the option.
usage: ab_same.py "8=0" "8=32" "8=0,7=8" ...   (option=value[,option=value]; options not named are reset to
order 0, strip rows auto, default weights).  AB_KIND=4 for the G4 bank, AB_HANDLES=n repeats on n handles, AB_STEPS=k launches per timed burst (default 20)."""
import os, sys, statistics
os.environ.setdefault("CVS_PLACEMENT_SEARCH", "0")
os.environ.setdefault("CVS_AUTOTUNE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

STEPS = int(os.environ.get("AB_STEPS", "20"))   # AB_STEPS=300: sustained launches (the card at its power cap) instead of short bursts


def timeit(fn, steps=None):
    steps = steps or STEPS
    for _ in range(steps // 3 if steps > 60 else 0):   # lead-in for the long form
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps

specs = sys.argv[1:] or ["8=0", "8=32"]
cfgs = {v: [tuple(int(x) for x in kv.split("=")) for kv in v.split(",") if kv] for v in specs}
base = [(L.OPT_BLOCK_ORDER, 0), (L.OPT_STRIP_ROWS, 0), (L.OPT_XCD_WEIGHTS, 0)]
n = int(os.environ.get("AB_N", "4096"))
g4 = os.environ.get("AB_KIND", "2") == "4"
img = torch.rand((n, n), device="cuda")
g, h = torch.empty_like(img), torch.empty_like(img)
outs = [torch.empty_like(img) for _ in range(8)]
imgs8 = [img] + [torch.rand((n, n), device="cuda") for _ in range(7)] if os.environ.get("AB_ROT") else None
rot = {"i": 0}
for hi in range(int(os.environ.get("AB_HANDLES", "2"))):
    f = cv.SteerableFiltersG4(None) if g4 else cv.SteerableFiltersG2(None)
    def apply(v):
        for o, val in base + cfgs[v]:
            f.set_option(o, val)
    legs = {"M6 basis": (lambda: f.setup(img), 48), "M6 +steer": (lambda: f.setup_steer(img, 0.3, out=(g, h)), 56)} if g4 else {
        "M1 basis": (lambda: f.setup(img, flags=cv.SETUP_BASIS), 32), "M2 +steer": (lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h)), 40),
        "M4 full": (lambda: f.setup(img, flags=cv.SETUP_FULL), 52), "M5 pipeline": (lambda: f.pipeline(img, out=outs), 84)}
    if imgs8 and not g4:   # AB_ROT=1: every launch filters a different image (inputs come from HBM, not the Infinity Cache)
        def step_rot():
            rot["i"] = (rot["i"] + 1) & 7
            f.setup_steer(imgs8[rot["i"]], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
        def step_rot1():
            rot["i"] = (rot["i"] + 1) & 7
            f.setup(imgs8[rot["i"]], flags=cv.SETUP_BASIS)
        legs = {"M1 rotating": (step_rot1, 32), "M2 rotating": (step_rot, 40)}
    print("handle %d" % hi)
    for name, (fn, bpp) in legs.items():
        res = {v: [] for v in specs}
        for v in specs:
            apply(v); fn(); fn()
        for r in range(8):
            for v in specs:
                apply(v)
                res[v].append(timeit(fn))
        print("  %-12s" % name + " | ".join("%s %.4f ms (%.1f%%)" % (v, statistics.median(res[v]), bpp * n * n / statistics.median(res[v]) / 1e6 / 80) for v in specs), flush=True)
