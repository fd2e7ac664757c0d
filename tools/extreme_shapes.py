#!/usr/bin/env python3
"""tools/extreme_shapes.py -- extreme aspect ratios through both banks, against the oracle: one row of a million pixels, one column
of two million, thin and wide bands on the strip kernels, plus the fused steer on them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cvsteer_amd as cv
from oracle import pyoracle as ora
rng = np.random.default_rng(7)
fails = 0
for rows, cols in ((1, 1_000_000), (3, 2_000_003), (2_000_000, 1), (100_000, 3), (13, 100_000), (19, 65_537), (100_000, 5), (40_000, 64), (7, 7), (2, 2), (1, 2), (65_536, 65)):
    img = rng.random((rows, cols), dtype=np.float32)
    for kind in (2, 4):
        w, s = (4, 0.67) if kind == 2 else (6, 0.5)
        nb = 7 if kind == 2 else 11
        try:
            f = cv.SteerableFiltersG2(None) if kind == 2 else cv.SteerableFiltersG4(None)
            x = torch.from_numpy(img).cuda()
            g, h = f.setup_steer(x, 0.3)
            got = np.stack([f.basis(p).cpu().numpy() for p in range(nb)])
            truth = ora.basis(kind, img, w, s, f64=True)
            err = float(np.abs(got - truth).max())
            og, oh = (ora.g2_steer_scalar if kind == 2 else ora.g4_steer_scalar)(got, 0.3)
            serr = max(float(np.abs(g.cpu().numpy() - og).max()), float(np.abs(h.cpu().numpy() - oh).max()))
            ok = err <= 1e-5 and serr <= 4e-6
            print("%9d x %9d kind %d: basis err %.2e steer err %.2e %s  [%s]" % (rows, cols, kind, err, serr, "ok" if ok else "FAIL", f.launch_info()), flush=True)
            fails += not ok
            del f, x, g, h
        except Exception as ex:
            fails += 1
            print("%9d x %9d kind %d: ERROR %s: %s" % (rows, cols, kind, type(ex).__name__, ex), flush=True)
print("extreme shapes: %d failures" % fails)
sys.exit(1 if fails else 0)
