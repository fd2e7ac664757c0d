#!/usr/bin/env python3
"""tools/host_probe.py -- the host-plane path (64 MiB up, 2 x 64 MiB down per 4096x4096 image): sequential vs
overlapped, number of bands (CVS_HOST_BANDS), f32 vs 8-bit input."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
n = 4096
imgs = [np.random.default_rng(i).random((n, n), dtype=np.float32) for i in range(4)]
u8 = [(im * 255).astype(np.uint8) for im in imgs]
g, h = np.empty_like(imgs[0]), np.empty_like(imgs[0])
def run(tag, overlap, src):
    f = cv.SteerableFiltersG2(None); f.set_option(L.OPT_HOST_OVERLAP, overlap)
    f.setup_steer(src[0], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
    best = 1e9
    for r in range(3):
        t0 = time.perf_counter()
        for im in src: f.setup_steer(im, 0.3, flags=cv.SETUP_BASIS, out=(g, h))
        best = min(best, (time.perf_counter() - t0) / len(src))
    print("%-40s %.3f ms per image  %.2f Gpix/s" % (tag, best * 1e3, n * n / best / 1e9), flush=True)
run("sequential, f32 image", 0, imgs)
for nb in (2, 4, 8, 16, 32):
    os.environ["CVS_HOST_BANDS"] = str(nb)
    run("overlapped, %2d bands, f32 image" % nb, 1, imgs)
os.environ["CVS_HOST_BANDS"] = "8"
run("sequential, 8-bit image", 0, u8)
run("overlapped, 8 bands, 8-bit image", 1, u8)
