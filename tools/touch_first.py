#!/usr/bin/env python3
"""tools/touch_first.py -- does reading a fresh image once in ONE burst (a pass that pulls it into the Infinity Cache) pay
for itself in the filter launch that follows?  8 rotating 4096^2 inputs, M2 (filter + steer).  (a) plain; (b) a read pass
over the image right before each call, same stream; (c) the read pass of the NEXT image on a second stream while the
current one is filtered."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
n = int(os.environ.get("AB_N", "4096"))
imgs = [torch.rand((n, n), device="cuda") for _ in range(8)]
g, h = torch.empty_like(imgs[0]), torch.empty_like(imgs[0])
f = cv.SteerableFiltersG2(None)
sink = torch.zeros(1, device="cuda")
side = torch.cuda.Stream()
def t(fn, reps=40):
    for i in range(8): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
def plain(i): f.setup_steer(imgs[i & 7], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
def touch_same(i):
    sink.add_(imgs[i & 7].view(-1)[::1].sum())
    f.setup_steer(imgs[i & 7], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
def touch_only(i): sink.add_(imgs[i & 7].sum())
def touch_ahead(i):
    with torch.cuda.stream(side):
        sink.add_(imgs[(i + 1) & 7].sum())
    f.setup_steer(imgs[i & 7], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
for rnd in range(3):
    a = t(plain); c = t(touch_only); b = t(touch_same); d = t(touch_ahead)
    pct = lambda ms: 40 * n * n / ms / 1e6 / 80
    print("plain %.4f ms (%.1f%%) | read pass alone %.4f ms | read pass + filter %.4f ms (filter net %.4f ms = %.1f%%) | read pass of the next image on a second stream %.4f ms (%.1f%%)" % (a, pct(a), c, b, b - c, pct(b - c), d, pct(d)), flush=True)
