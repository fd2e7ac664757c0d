#!/usr/bin/env python3
"""tools/size_trend.py -- % of 8 TB/s of the fused variants (M1 7 planes, M2 9, M4 12, M5 20) against image size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
def timeit(fn, steps=10):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps
for n in (2048, 2896, 4096, 5792, 8192, 11584):
    img = torch.rand((n, n), device="cuda")
    outs = [torch.empty_like(img) for _ in range(8)]
    f = cv.SteerableFiltersG2(None)
    res = []
    for name, bpp, fn in (("M1", 32, lambda: f.setup(img, flags=cv.SETUP_BASIS)), ("M2", 40, lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(outs[0], outs[1]))),
                          ("M4", 52, lambda: f.setup(img, flags=cv.SETUP_FULL)), ("M5", 84, lambda: f.pipeline(img, out=outs))):
        ms = timeit(fn)
        res.append("%s %.3f ms %4.1f%%" % (name, ms, bpp * n * n / ms / 1e6 / 80))
    print("%5d^2 (%5.1f Mpix): " % (n, n * n / 1e6) + "   ".join(res), flush=True)
    del f, img, outs
