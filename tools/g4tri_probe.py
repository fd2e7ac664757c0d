#!/usr/bin/env python3
"""tools/g4tri_probe.py (needs tools/patches/g4_three_parts_probe.patch applied) -- G4 basis: the product's pair launch (five planes,
then six) against three parts in one launch (g4a..g4e, h4a..h4c, h4d..h4f: five, three, three planes; CVS_G4_TRI=1), same
handles, interleaved rounds; values compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVS_AUTOTUNE"] = "0"
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

n = 4096
imgs = [torch.rand((n, n), device="cuda") for _ in range(8)]


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
    os.environ["CVS_G4_TRI"] = "0"
    f.setup(imgs[0])
    ref = [f.basis(p).clone() for p in range(11)]
    os.environ["CVS_G4_TRI"] = "1"
    f.setup(imgs[0])
    same = all(torch.equal(f.basis(p), ref[p]) for p in range(11))
    for sr in (0, 66, 131):
        if sr:
            f.set_strip_rows(sr)
        for name, fn in (("M6 basis", lambda i: f.setup(imgs[0])), ("M6 basis, rotating inputs", lambda i: f.setup(imgs[i & 7]))):
            res = {0: [], 1: []}
            for rnd in range(5):
                for m in (0, 1):
                    os.environ["CVS_G4_TRI"] = str(m)
                    res[m].append(timeit(fn))
            print("handle %d strip %3d %-26s pair %.4f ms %.3f | three parts %.4f ms %.3f   (equal: %s)" % (
                hnd, sr, name, sorted(res[0])[2], 48 * n * n / sorted(res[0])[2] / 1e6 / 8000, sorted(res[1])[2], 48 * n * n / sorted(res[1])[2] / 1e6 / 8000, same), flush=True)
