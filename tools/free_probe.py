#!/usr/bin/env python3
"""tools/free_probe.py -- does an allocate-and-free of a big dummy block change the speed of launches on EXISTING state blocks?
(tools/region_probe.py: processes that freed a 16-32 GiB dummy before measuring ran the full setup 7 % and the batch up to 25 % faster.)
Same handles throughout: measure, dummy of S GiB allocated and freed, measure again, for growing S."""
import os, sys, statistics
os.environ["CVS_AUTOTUNE"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import cvsteer_amd as cv
frames = torch.rand((32, 1080, 1920), device="cuda")
img = torch.rand((4096, 4096), device="cuda")
fb = cv.SteerableFiltersG2(None)
out = fb.pipeline_batch(frames)
f = cv.SteerableFiltersG2(None)
outs = cv.alloc_planes(8, 4096, 4096, device="cuda")
f.setup(img, flags=cv.SETUP_FULL)


def timeit(fn, steps):
    for _ in range(max(4, steps // 3)):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def measure(tag):
    c4 = statistics.median(timeit(lambda: fb.pipeline_batch(frames, out=out), 20) for _ in range(3))
    m4 = statistics.median(timeit(lambda: f.setup(img, flags=cv.SETUP_FULL), 100) for _ in range(3))
    m5 = statistics.median(timeit(lambda: f.pipeline(img, out=outs), 80) for _ in range(3))
    print("%-34s C4 %.3f | M4 %.3f | M5 %.3f" % (tag, 84 * 32 * 1080 * 1920 / (c4 * 1e-3) / 8e12, 52 * 4096 * 4096 / (m4 * 1e-3) / 8e12, 84 * 4096 * 4096 / (m5 * 1e-3) / 8e12), flush=True)


measure("start")
measure("again")
for gib in (1, 4, 16, 32, 64):
    d = torch.empty(gib << 30, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    measure("dummy of %d GiB allocated" % gib)
    del d
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    measure("... and freed")
