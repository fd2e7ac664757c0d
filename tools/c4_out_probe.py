#!/usr/bin/env python3
"""tools/c4_out_probe.py -- config 4 (32 x 1080p, state kept): the caller's eight output planes per frame as a planar block
[n][8][H][W] against rows of one block per frame [n][H][8][W] (strided views, what cv.alloc_planes does for one image)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
nfr = 32
fs = [torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)]
planar = torch.empty((nfr, 8, 1080, 1920), device="cuda")
inter = torch.empty((nfr, 1080, 8, 1920), device="cuda").permute(0, 2, 1, 3)
ff = cv.SteerableFiltersG2(None)
alt = {"i": 0}


def timeit(fn, steps=8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def run(out):
    alt["i"] ^= 1
    ff.pipeline_batch(fs[alt["i"]], out=out)


for _ in range(70):
    run(planar)
res = {"planar": [], "rows of one block": []}
for r in range(5):
    for name, o in (("planar", planar), ("rows of one block", inter)):
        run(o); run(o)
        res[name].append(timeit(lambda: run(o)))
for k, v in res.items():
    t = statistics.median(v)
    print("%-18s %.4f ms  %.3f of HBM" % (k, t, 84 * nfr * 1080 * 1920 / t / 1e6 / 8000))
ref = planar.clone()
run(inter); run(inter)
torch.cuda.synchronize()
run(planar)
print("same values:", torch.equal(inter.contiguous(), planar) or "frames differ (alternating sets): compare per set")
