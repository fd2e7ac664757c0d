#!/usr/bin/env python3
"""tools/c4_probe.py -- config-4 leg under the microscope: host time per call vs device time per launch, one frame
set vs two alternating sets, strip heights, state kept vs outputs only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
nfr = 32
fs = [torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)]
out = torch.empty((nfr, 8, 1080, 1920), device="cuda")
pix = nfr * 1080 * 1920
def run(tag, f, fn, steps=20):
    for _ in range(6): fn(0); fn(1)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for i in range(steps): fn(i)
    b.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    print("%-58s host %.3f ms/call  device %.3f ms/launch  %5.1f%% of 8 TB/s (84 B/pix)" % (tag, (t1 - t0) / steps * 1e3, ms, 84 * pix / ms / 1e6 / 80), flush=True)
for sr in (0, 10, 19, 28):
    f = cv.SteerableFiltersG2(None)
    if sr: f.set_strip_rows(sr)
    run("state kept, one frame set, strip rows %d" % sr, f, lambda i: f.pipeline_batch(fs[0], out=out))
    run("state kept, two alternating sets, strip rows %d" % sr, f, lambda i: f.pipeline_batch(fs[i & 1], out=out))
for order in (0, 1):
    f = cv.SteerableFiltersG2(None)
    f.set_option(L.OPT_BLOCK_ORDER, order); f.set_option(L.OPT_AUTOTUNE, 0)
    run("state kept, two sets, block order %d, no autotune" % order, f, lambda i: f.pipeline_batch(fs[i & 1], out=out))
f = cv.SteerableFiltersG2(None)
f.set_option(L.OPT_STORE_POLICY, 1)
run("state kept, two sets, plain stores", f, lambda i: f.pipeline_batch(fs[i & 1], out=out))
# frame by frame through the single-image pipeline (20 launches of 1080p)
f = cv.SteerableFiltersG2(None)
def one_by_one(i):
    for k in range(nfr):
        f.pipeline(fs[i & 1][k], out=[out[k][j] for j in range(8)])
run("state kept, frame by frame (32 launches)", f, one_by_one, steps=5)
