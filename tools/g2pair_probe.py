#!/usr/bin/env python3
"""tools/g2pair_probe.py (needs tools/patches/g2_half_banks_probe.patch applied) -- G2 basis (M1) and fused filter + steer (M2)
as ONE 7-plane kernel (product) against two half banks in one launch (G: g2a..g2c (+ g), H: h2a..h2d (+ h); CVS_G2_PAIR=1), on
the same handles, interleaved rounds, resident and rotating inputs; values compared as well."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVS_AUTOTUNE"] = "0"
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

n = 4096
imgs = [torch.rand((n, n), device="cuda") for _ in range(8)]
g, h = torch.empty_like(imgs[0]), torch.empty_like(imgs[0])


def timeit(fn, steps=20, warm=3):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(steps):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


keep = []
for hnd in range(3):
    for place in (0, 1):
        f = cv.SteerableFiltersG2(None)
        f.set_option(L.OPT_PLACEMENT_SEARCH, place)
        keep.append(f)
        os.environ["CVS_G2_PAIR"] = "0"
        g0, h0 = f.setup_steer(imgs[0], 0.3, flags=cv.SETUP_BASIS)
        ref = [f.basis(p).clone() for p in range(7)] + [g0.clone(), h0.clone()]
        os.environ["CVS_G2_PAIR"] = "1"
        g1, h1 = f.setup_steer(imgs[0], 0.3, flags=cv.SETUP_BASIS)
        got = [f.basis(p) for p in range(7)] + [g1, h1]
        same = all(torch.equal(a, b) for a, b in zip(got, ref))
        legs = (("M1 basis", 32, lambda i: f.setup(imgs[0], flags=cv.SETUP_BASIS)),
                ("M2 +steer", 40, lambda i: f.setup_steer(imgs[0], 0.3, flags=cv.SETUP_BASIS, out=(g, h))),
                ("M2 rotating", 40, lambda i: f.setup_steer(imgs[i & 7], 0.3, flags=cv.SETUP_BASIS, out=(g, h))))
        for order in (0, 1):
            f.set_option(L.OPT_BLOCK_ORDER, order)
            for name, bpp, fn in legs:
                res = {0: [], 1: []}
                for rnd in range(5):
                    for m in (0, 1):
                        os.environ["CVS_G2_PAIR"] = str(m)
                        res[m].append(timeit(fn))
                print("handle %d placement %d order %d %-12s one kernel %.4f ms %.3f | half banks %.4f ms %.3f   (equal: %s)" % (
                    hnd, place, order, name, sorted(res[0])[2], bpp * n * n / sorted(res[0])[2] / 1e6 / 8000,
                    sorted(res[1])[2], bpp * n * n / sorted(res[1])[2] / 1e6 / 8000, same), flush=True)
