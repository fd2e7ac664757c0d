#!/usr/bin/env python3
"""tools/c4_search_probe.py -- config 4 (32 x 1080p, state kept, outputs in one block): the handle's own plain state block against
the block kept by the opt-in batch block search (CVS_OPT_PLACEMENT_SEARCH = 1: the real launch timed on up to six candidate blocks, once);
tuner on (library defaults otherwise), rotating input blocks (every frame new), sustained launches.  One line per handle pair."""
import os, sys, statistics, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
fr = [torch.rand((32, 1080, 1920), device="cuda") for _ in range(2)]
nb = 84 * 32 * 1080 * 1920
flip = {"i": 0}


def timeit(fn, steps=20):
    for _ in range(6):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


for i in range(int(os.environ.get("PROBE_HANDLES", "3"))):
    line = []
    for placement in (0, 1):
        f = cv.SteerableFiltersG2(None)
        f.set_option(L.OPT_PLACEMENT_SEARCH, placement)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = f.pipeline_batch(fr[0])
        torch.cuda.synchronize()
        first_ms = (time.perf_counter() - t0) * 1e3

        def step():
            flip["i"] ^= 1
            f.pipeline_batch(fr[flip["i"]], out=out)
        for _ in range(70):
            step()
        res = [timeit(step) for _ in range(3)]
        li = f.launch_info()
        line.append("%s: %.3f (first call %.1f ms, probe %.1f ms, order %d)" % ("search" if placement else "plain ", nb / (statistics.median(res) * 1e-3) / 8e12, first_ms, li["probe_ms"], li["block_order"]))
        del f, out
    print("pair %d  " % i + " | ".join(line), flush=True)
