#!/usr/bin/env python3
"""tools/region_probe.py D_GB [free] -- does WHERE in VRAM a state block lands decide its speed?  A dummy allocation of D_GB GiB is made
first (and, with `free`, released again after the handles' state blocks exist); then config 4 (32 x 1080p, state kept) and the
4096^2 full setup / pipeline on fresh handles, tuner off, sustained launches."""
import os, sys, statistics
os.environ["CVS_AUTOTUNE"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import cvsteer_amd as cv
d_gb = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
free_after = len(sys.argv) > 2 and sys.argv[2] == "free"
frames = torch.rand((32, 1080, 1920), device="cuda")
img = torch.rand((4096, 4096), device="cuda")
dummy = torch.empty(int(d_gb * (1 << 30)), dtype=torch.uint8, device="cuda") if d_gb > 0 else None


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


fb = cv.SteerableFiltersG2(None)
out = fb.pipeline_batch(frames)
f = cv.SteerableFiltersG2(None)
outs = cv.alloc_planes(8, 4096, 4096, device="cuda")
f.setup(img, flags=cv.SETUP_FULL)
if free_after:
    del dummy
    torch.cuda.empty_cache()
torch.cuda.synchronize()
c4 = statistics.median(timeit(lambda: fb.pipeline_batch(frames, out=out), 20) for _ in range(3))
m4 = statistics.median(timeit(lambda: f.setup(img, flags=cv.SETUP_FULL), 100) for _ in range(3))
m5 = statistics.median(timeit(lambda: f.pipeline(img, out=outs), 80) for _ in range(3))
m1 = statistics.median(timeit(lambda: f.setup(img, flags=cv.SETUP_BASIS), 100) for _ in range(3))
print("dummy %5.1f GiB%s:  C4 %.3f | M4 %.3f | M5 %.3f | M1 %.3f   state at %#x / %#x" % (d_gb, " (freed)" if free_after else "", 84 * 32 * 1080 * 1920 / (c4 * 1e-3) / 8e12,
      52 * 4096 * 4096 / (m4 * 1e-3) / 8e12, 84 * 4096 * 4096 / (m5 * 1e-3) / 8e12, 32 * 4096 * 4096 / (m1 * 1e-3) / 8e12, fb.basis_view(0)[0], f.basis_view(0)[0]), flush=True)
