#!/usr/bin/env python3
"""tools/first_call_probe.py -- the reference's one-object-per-image pattern: create, one fused call on a new image, sync,
destroy; time per object with the placement probe on (1), off (0) and forced to find nothing (CVS_PLACEMENT_NO_WINDOW=1 is
not a product switch: the probe's verdict is what the box gives)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cvsteer_amd as cv
from cvsteer_amd import _lib as L
imgs = [torch.rand((4096, 4096), device="cuda") for _ in range(3)]
g, h = torch.empty_like(imgs[0]), torch.empty_like(imgs[0])
for placement in (1, 0):
    cv.lib().cvs_release_cached_memory()
    for i in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        f = cv.SteerableFiltersG2(None)
        f.set_option(L.OPT_PLACEMENT_SEARCH, placement)
        e0.record(); f.setup_steer(imgs[i % 3], 0.3, flags=cv.SETUP_BASIS, out=(g, h)); e1.record()
        torch.cuda.synchronize()
        del f
        print("placement %d object %d: call %.3f ms, wall %.3f ms" % (placement, i, e0.elapsed_time(e1), (time.perf_counter() - t0) * 1e3), flush=True)
