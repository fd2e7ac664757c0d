#!/usr/bin/env python3
"""tools/ab_point.py -- interleaved A/B of the pointwise kernels' grid shape (experiment; env read per launch)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv

def timeit(fn, steps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps

def main():
    n = 4096
    img = torch.rand((n, n), device="cuda")
    f = cv.SteerableFiltersG2(img)
    outs = [torch.empty_like(img) for _ in range(5)]
    th = f.getDominantOrientationAngle()
    mag, ph = torch.empty_like(img), torch.empty_like(img)
    feats = [torch.empty_like(img) for _ in range(3)]
    cfgs = [(64, 16), (64, 8), (64, 32), (64, 64), (16, 16), (4, 16), (1024, 16), (1024, 64), (1024, 100000)]
    legs = {
        "M3 steer scalar": (lambda: f.steer(0.3, out=outs[:2]), 36),
        "M3 steer scalar full": (lambda: f.steer(0.3, full=True, out=outs), 60),
        "M3 steer map full": (lambda: f.steer(th, full=True, out=outs), 64),
        "mag_phase": (lambda: f.computeMagnitudeAndPhase(outs[0], outs[1]), 16),
        "find x3": (lambda: f.find(mag, ph), 20),
    }
    for name, (fn, bpp) in legs.items():
        try:
            fn()
        except Exception as exc:
            print(name, "skipped:", exc); continue
        res = {c: [] for c in cfgs}
        for r in range(8):
            for c in cfgs:
                os.environ["CVS_POINT_GX"], os.environ["CVS_POINT_CAP"] = str(c[0]), str(c[1])
                res[c].append(timeit(fn))
        line = "%-22s" % name
        for c in cfgs:
            med = statistics.median(res[c])
            line += " | gx%d cap%d %.4f (%4.1f%%)" % (c[0], c[1], med, bpp * n * n / med / 1e6 / 80)
        print(line, flush=True)

if __name__ == "__main__":
    main()
