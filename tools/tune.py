#!/usr/bin/env python3
"""tools/tune.py -- sweep the basis kernel's launch knobs on the GPU box (strip rows x store policy).
A coarse sequential sweep: every configuration is its own handle (its own state allocation -- see
tools/alloc_modes.py for what that alone can do), so differences under ~15 % need tools/ab.py to confirm."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("CVS_PLACEMENT_SEARCH", "0"); _os.environ.setdefault("CVS_AUTOTUNE", "0")  # A/B runs compare like with like
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

def timeit(fn, steps=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    m = int(sys.argv[2]) if len(sys.argv) > 2 else n     # rows x cols = n x m
    kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ["G2", "G4"]
    img = torch.rand((n, m), device="cuda")
    g, h = torch.empty_like(img), torch.empty_like(img)
    for kind, cls, bpp in (("G2", cv.SteerableFiltersG2, 32), ("G4", cv.SteerableFiltersG4, 48)):
        if kind not in kinds:
            continue
        f = cls(None)
        nt_, halo = (9, 8) if kind == "G2" else (13, 12)
        for pol, split in ((2, 1), (2, 0), (2, 2)) if kind == "G4" else ((1, 1), (2, 1)):
            f.set_option(L.OPT_STORE_POLICY, pol)
            f.set_option(L.OPT_G4_SPLIT, split)
            for k in (2, 3, 4, 6, 8, 11, 15, 22):
                sr = k * nt_ - halo
                f.set_strip_rows(sr)
                ms = timeit(lambda: f.setup(img, flags=cv.SETUP_BASIS))
                ms2 = timeit(lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h)))
                print("%s split=%d policy=%s strip_rows=%3d  basis %.4f ms %7.0f Mpix/s %6.0f GB/s (%.1f%%) | +steer %.4f ms %7.0f Mpix/s %6.0f GB/s" % (
                    kind, split, "plain" if pol == 1 else "nt", sr, ms, n*m/ms/1e3, bpp*n*m/ms/1e6, bpp*n*m/ms/1e6/80,
                    ms2, n*m/ms2/1e3, (bpp+8)*n*m/ms2/1e6), flush=True)

if __name__ == "__main__":
    main()
