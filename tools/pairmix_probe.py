#!/usr/bin/env python3
"""tools/pairmix_probe.py (needs tools/patches/g4_pair_halves_on_one_xcd_probe.patch applied) -- G4 pair kernel: half banks as grid.z (all G tiles, then all H tiles) against both halves of a tile
on one XCD, eight workgroups apart (CVS_PAIR_MIX=1), same handles, interleaved rounds; values compared as well."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVS_AUTOTUNE"] = "0"
import torch
import cvsteer_amd as cv

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
    f = cv.SteerableFiltersG4(None)
    keep.append(f)
    os.environ["CVS_PAIR_MIX"] = "0"
    f.setup(imgs[0])
    ref = [f.basis(p).clone() for p in range(11)]
    os.environ["CVS_PAIR_MIX"] = "1"
    f.setup(imgs[0])
    same = all(torch.equal(f.basis(p), ref[p]) for p in range(11))
    legs = (("M6 basis", 48, lambda i: f.setup(imgs[0])), ("M6 +steer", 56, lambda i: f.setup_steer(imgs[0], 0.3, out=(g, h))),
            ("M6 basis, rotating inputs", 48, lambda i: f.setup(imgs[i & 7])))
    for name, bpp, fn in legs:
        res = {0: [], 1: []}
        for rnd in range(5):
            for m in (0, 1):
                os.environ["CVS_PAIR_MIX"] = str(m)
                res[m].append(timeit(fn))
        print("handle %d %-26s z-major %.4f ms %.3f | halves on one XCD %.4f ms %.3f   (planes equal: %s)" % (
            hnd, name, sorted(res[0])[2], bpp * n * n / sorted(res[0])[2] / 1e6 / 8000, sorted(res[1])[2], bpp * n * n / sorted(res[1])[2] / 1e6 / 8000, same), flush=True)
