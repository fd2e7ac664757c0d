#!/usr/bin/env python3
"""tools/handle0_probe.py -- is the process's FIRST handle slower than later ones (bench.py: headline handle 0.80, the fresh
M2_untuned handle 0.815)?  Six handles created one after the other (tuner on / off alternating), all alive; M2 loop on each,
three interleaved rounds."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
n = 4096
img = torch.rand((n, n), device="cuda")
g, h = torch.empty_like(img), torch.empty_like(img)


def timeit(fn, steps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


hs = []
for i in range(6):
    f = cv.SteerableFiltersG2(None)
    f.set_option(L.OPT_AUTOTUNE, 1 - (i & 1))
    for _ in range(48):
        f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))
    torch.cuda.synchronize()
    hs.append(f)
res = [[] for _ in hs]
for r in range(5):
    for i, f in enumerate(hs):
        f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))
        res[i].append(timeit(lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))))
for i, f in enumerate(hs):
    li = f.launch_info()
    t = statistics.median(res[i])
    print("handle %d tuner %d: %.4f ms %.3f  (order %d strip %d)  state plane 0 at %#x" % (i, 1 - (i & 1), t, 40 * n * n / t / 1e6 / 8000, li["block_order"], li["strip_rows"], f.basis_view(0)[0]), flush=True)
print("img %#x g %#x h %#x" % (img.data_ptr(), g.data_ptr(), h.data_ptr()))
