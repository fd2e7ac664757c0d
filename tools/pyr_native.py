#!/usr/bin/env python3
"""tools/pyr_native.py -- BASELINE config 3 through the native entry (cvs_batch_pyramid_setup, one rank): pyramid build +
G2+H2 basis on all 5 levels of an 8192x8192 image, against the per-level Python loop of bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import batch
big = torch.rand((8192, 8192), device="cuda")
nb = batch.NativeBatch.local((0,))
for _ in range(4): t = nb.pyramid_setup(big, 8192, 8192, 5, flags=cv.SETUP_BASIS)
acc = {"broadcast": 0.0, "compute": 0.0, "gather": 0.0}
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    t = nb.pyramid_setup(big, 8192, 8192, 5, flags=cv.SETUP_BASIS)
    for k in acc: acc[k] += t[k] / 10
wall = (time.perf_counter() - t0) / 10 * 1e3
ppix = sum((8192 >> l) ** 2 for l in range(5))
print("native: compute (pyramid build + 5 filter launches) %.3f ms, wall %.3f ms per call  -> %.1f %% of 8 TB/s on the 32 B/pix of the filters alone" % (acc["compute"], wall, 32 * ppix / acc["compute"] / 1e6 / 80))
f = cv.SteerableFiltersG2(None)
lv = f.pyramid(big, 5)
hp = [cv.SteerableFiltersG2(None) for _ in lv]
def filt():
    for hnd, l in zip(hp, lv): hnd.setup(l, flags=cv.SETUP_BASIS)
for _ in range(6): filt()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): filt()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
a.record()
for _ in range(10): f.pyramid(big, 5)
b.record(); torch.cuda.synchronize()
print("python loop: filters %.3f ms (%.1f %%), pyramid build %.3f ms" % (ms, 32 * ppix / ms / 1e6 / 80, a.elapsed_time(b) / 10))
for l, hnd in zip(lv, hp):
    a.record()
    for _ in range(10): hnd.setup(l, flags=cv.SETUP_BASIS)
    b.record(); torch.cuda.synchronize()
    m = a.elapsed_time(b) / 10
    print("   level %5d^2: %.4f ms  %4.1f %%" % (l.shape[0], m, 32 * l.shape[0] * l.shape[1] / m / 1e6 / 80))
