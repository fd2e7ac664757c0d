#!/usr/bin/env python3
"""tools/shape_sweep.py -- does the fused pipeline's throughput depend on the frame geometry?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("CVS_PLACEMENT_SEARCH", "0"); _os.environ.setdefault("CVS_AUTOTUNE", "0")  # A/B runs compare like with like
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

def timeit(fn, steps=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps

for (r, c) in [(1080, 1920), (1088, 1920), (1080, 2048), (1024, 2048), (2160, 3840), (4096, 4096), (4096, 1920), (1080, 4096), (2048, 2048)]:
    n = max(1, (64 << 20) // (r * c))
    frames = torch.rand((n, r, c), device="cuda")
    out = torch.empty((n, 8, r, c), device="cuda")
    f = cv.SteerableFiltersG2(None)
    line = "%5dx%-5d n=%3d " % (r, c, n)
    for sr in (10, 19, 37, 64):
        f.set_strip_rows(sr)
        ms = timeit(lambda: f.pipeline_batch(frames, out=out))
        line += "| sr=%2d %6.0f Mpix/s %4.1f%% " % (sr, n * r * c / ms / 1e3, 84 * n * r * c / ms / 1e6 / 80)
    print(line, flush=True)
