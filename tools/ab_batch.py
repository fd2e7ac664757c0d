#!/usr/bin/env python3
"""tools/ab_batch.py -- BASELINE config 4 shape (32 x 1080x1920 frames, fused pipeline, one batched launch):
launch options compared on ONE handle (see tools/ab_same.py)."""
import os, sys, statistics
os.environ["CVS_PLACEMENT_SEARCH"] = "0"
os.environ["CVS_AUTOTUNE"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

def timeit(fn, steps=5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps

specs = sys.argv[1:] or ["8=0", "8=1"]
cfgs = {v: [tuple(int(x) for x in kv.split("=")) for kv in v.split(",") if kv] for v in specs}
base = [(L.OPT_BLOCK_ORDER, 0), (L.OPT_STRIP_ROWS, 0), (L.OPT_XCD_WEIGHTS, 0)]
nfr = 32
frames = torch.rand((nfr, 1080, 1920), device="cuda")
fout = torch.empty((nfr, 8, 1080, 1920), device="cuda")
fo3 = torch.empty((nfr, 3, 1080, 1920), device="cuda")
f = cv.SteerableFiltersG2(None)
fp = nfr * 1080 * 1920
for name, persist, fn, bpp in (("all state kept (84 B/pix)", True, lambda: f.pipeline_batch(frames, out=fout), 84),
                               ("feature maps only (16 B/pix)", False, lambda: f.pipeline_batch(frames, out=fo3, outputs=(5, 6, 7)), 16)):
    f.set_persist(persist)
    res = {v: [] for v in specs}
    for r in range(7):
        for v in specs:
            for o, val in base + cfgs[v]: f.set_option(o, val)
            if r == 0: fn()
            res[v].append(timeit(fn))
    print("%-30s" % name + " | ".join("%s %.4f ms/frame (%.0f Mpix/s, %.1f%%)" % (v, statistics.median(res[v]) / nfr, fp / statistics.median(res[v]) / 1e3,
                                                                          bpp * fp / statistics.median(res[v]) / 1e6 / 80) for v in specs), flush=True)
