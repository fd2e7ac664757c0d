#!/usr/bin/env python3
"""tools/c3_fused.py -- BASELINE config 3 as a whole (build the 5-level pyramid of one 8192^2 image AND filter every
level, G2+H2 basis): pyrDown launches + filter launches against the fused form in which the filter launch of level k
also writes level k+1 (cvs_setup_pyr)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
bigs = [torch.rand((8192, 8192), device="cuda") for _ in range(2)]
f0 = cv.SteerableFiltersG2(None)
lv = f0.pyramid(bigs[0], 5)
pix = sum(l.numel() for l in lv)
hs = [cv.SteerableFiltersG2(None) for _ in lv]
def t(fn, reps=20):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
def separate(i, rot):
    src = bigs[i & 1] if rot else bigs[0]
    cur = src
    for k in range(5):
        hs[k].setup(cur, flags=cv.SETUP_BASIS)
        if k < 4:
            lib_next = lv[k + 1]
            hs[k]._bind_stream(cur, lib_next)
            import ctypes as C
            from cvsteer_amd.api import _plane
            ps, pd = _plane(cur), _plane(lib_next)
            cv.lib().cvs_pyr_down(hs[k]._h, C.byref(ps), C.byref(pd))
            cur = lib_next
def fused(i, rot):
    cur = bigs[i & 1] if rot else bigs[0]
    for k in range(5):
        if k < 4:
            hs[k].setup_pyr(cur, flags=cv.SETUP_BASIS, out=lv[k + 1])
            cur = lv[k + 1]
        else:
            hs[k].setup(cur, flags=cv.SETUP_BASIS)
def filt_only(i, rot):
    for k in range(5):
        hs[k].setup(bigs[i & 1] if (rot and k == 0) else lv[k], flags=cv.SETUP_BASIS)
for rnd in range(2):
    for rot in (0, 1):
        tag = "two alternating 8192^2 images" if rot else "one image, re-filtered"
        for name, fn in (("filter only (5 launches)", filt_only), ("pyrDown + filter, separate (9 launches)", separate), ("filter launches emit the next level (5 launches)", fused)):
            ms = t(lambda i: fn(i, rot))
            print("%-32s %-50s %.4f ms  %.1f Gpix/s" % (tag, name, ms, pix / ms / 1e6), flush=True)
