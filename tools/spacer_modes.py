#!/usr/bin/env python3
"""tools/spacer_modes.py -- does the DISTANCE (in allocated bytes) between a handle's state block and the planes it
shares a launch with decide the speed?  For each spacer size: allocate the spacer, then the state, then free the
spacer; time M1 / M2 / M4 / M5 interleaved over all handles."""
import os, sys, statistics
os.environ["CVS_PLACEMENT_SEARCH"] = "0"
os.environ["CVS_AUTOTUNE"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

def timeit(fn, steps=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps

n = 4096
img = torch.rand((n, n), device="cuda")
g, hq = torch.empty_like(img), torch.empty_like(img)
outs = [torch.empty_like(img) for _ in range(8)]
spacers_gb = [float(v) for v in sys.argv[1:]] or [0, 1, 2, 4, 6, 8, 12, 16, 24, 32]
hs, keep = [], []
for sgb in spacers_gb:
    sp = torch.empty(int(sgb * (1 << 30)), dtype=torch.uint8, device="cuda") if sgb > 0 else None
    f = cv.SteerableFiltersG2(None)
    f.set_option(L.OPT_BLOCK_ORDER, 0)
    f.setup(img, flags=cv.SETUP_FULL)
    torch.cuda.synchronize()
    hs.append(f)
    keep.append(sp)          # spacers stay allocated: every state block sits sum(spacers so far) further on
legs = [("M1", lambda f: f.setup(img, flags=cv.SETUP_BASIS), 32), ("M2", lambda f: f.setup_steer(img, 0.3, out=(g, hq)), 40),
        ("M4", lambda f: f.setup(img, flags=cv.SETUP_FULL), 52), ("M5", lambda f: f.pipeline(img, out=outs), 84)]
print("spacer before each state block (GiB):", " ".join("%g" % s for s in spacers_gb))
for name, fn, bpp in legs:
    res = [[] for _ in hs]
    for f in hs: fn(f)
    for r in range(5):
        for i, f in enumerate(hs):
            res[i].append(timeit(lambda: fn(f)))
    print(name + " % of 8 TB/s: " + "  ".join("%.1f" % (bpp * n * n / statistics.median(r) / 1e6 / 80) for r in res), flush=True)
