#!/usr/bin/env python3
"""tools/fresh_probe.py -- launches on NEW images (8 rotating 4096^2 inputs; one object per image), tuner on: what it keeps and how
fast the settled loop runs.  CVS_NO_READ_AHEAD=1 takes the read-ahead candidates out (A/B); CVS_TUNE_VERBOSE=1 prints the tuner's table."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
n = 4096
imgs = [torch.rand((n, n), device="cuda") for _ in range(8)]
imgs_u8 = [(im * 255).to(torch.uint8) for im in imgs]
g, h = cv.alloc_planes(2, n, n, device="cuda")
f = cv.SteerableFiltersG2(None)
rot = {"i": 0}


def timeit(fn, steps=24):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def m2():
    rot["i"] = (rot["i"] + 1) & 7
    f.setup_steer(imgs[rot["i"]], 0.3, flags=cv.SETUP_BASIS, out=(g, h))


def m1():
    rot["i"] = (rot["i"] + 1) & 7
    f.setup(imgs[rot["i"]], flags=cv.SETUP_BASIS)


def m2u8():
    rot["i"] = (rot["i"] + 1) & 7
    f.setup_steer(imgs_u8[rot["i"]], 0.3, flags=cv.SETUP_BASIS, out=(g, h))


def obj():
    rot["i"] = (rot["i"] + 1) & 7
    fo = cv.SteerableFiltersG2(None)
    fo.setup_steer(imgs[rot["i"]], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
    del fo


for name, fn, bpp in (("M2 rotating", m2, 40), ("M1 rotating", m1, 32), ("M2 u8 rotating", m2u8, 37), ("M2 one object per image", obj, 40)):
    for _ in range(80):
        fn()
    torch.cuda.synchronize()
    t = statistics.median([timeit(fn) for _ in range(7)])
    li = f.launch_info()
    print("%-26s %.4f ms  %.3f of HBM   launch (order %d, strip %d, read-ahead %d)" % (name, t, bpp * n * n / t / 1e6 / 8000, li["block_order"], li["strip_rows"], li["read_ahead"]), flush=True)
