#!/usr/bin/env python3
"""tools/layout_probe.py -- A/B of the state-plane layout (CVS_STATE_LAYOUT: 0 = planar, 1 = rows of the planes
interleaved) on handles that live side by side in one process; launch configurations fixed (autotune off), rounds
interleaved.  usage: layout_probe.py [n_pairs]"""
import os, sys, statistics
os.environ.setdefault("CVS_PLACEMENT_SEARCH", "0")
os.environ.setdefault("CVS_AUTOTUNE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L


def timeit(fn, steps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


n = int(os.environ.get("AB_N", "4096"))
imgs8 = [torch.rand((n, n), device="cuda") for _ in range(8)]
img = imgs8[0]
g, h = torch.empty_like(img), torch.empty_like(img)
outs = [torch.empty_like(img) for _ in range(8)]
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
handles = []
for i in range(pairs):
    for lay in (0, 1):
        os.environ["CVS_STATE_LAYOUT"] = str(lay)
        f = cv.SteerableFiltersG2(None)
        f.setup(img, flags=cv.SETUP_FULL)
        handles.append((lay, f))
torch.cuda.synchronize()
cfgs = {"plain10": [(L.OPT_BLOCK_ORDER, 0), (L.OPT_STRIP_ROWS, 10), (L.OPT_XCD_WEIGHTS, 0)],
        "w504_10": [(L.OPT_BLOCK_ORDER, 1), (L.OPT_STRIP_ROWS, 10), (L.OPT_XCD_WEIGHTS, 504)],
        "xcdcol10": [(L.OPT_BLOCK_ORDER, 1000000), (L.OPT_STRIP_ROWS, 10), (L.OPT_XCD_WEIGHTS, 101)],
        "w403_19": [(L.OPT_BLOCK_ORDER, 1), (L.OPT_STRIP_ROWS, 19), (L.OPT_XCD_WEIGHTS, 403)],
        "plain19": [(L.OPT_BLOCK_ORDER, 0), (L.OPT_STRIP_ROWS, 19), (L.OPT_XCD_WEIGHTS, 0)]}
rot = {"i": 0}


def legs_of(f):
    def rot2():
        rot["i"] = (rot["i"] + 1) & 7
        f.setup_steer(imgs8[rot["i"]], 0.3, flags=cv.SETUP_BASIS, out=(g, h))

    def rot1():
        rot["i"] = (rot["i"] + 1) & 7
        f.setup(imgs8[rot["i"]], flags=cv.SETUP_BASIS)

    return {"M1": (lambda: f.setup(img, flags=cv.SETUP_BASIS), 32), "M2": (lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h)), 40),
            "M4": (lambda: f.setup(img, flags=cv.SETUP_FULL), 52), "M5": (lambda: f.pipeline(img, out=outs), 84),
            "M1rot": (rot1, 32), "M2rot": (rot2, 40)}


for cname, opts in cfgs.items():
    print("== config %s" % cname, flush=True)
    for leg in ("M1", "M2", "M4", "M5", "M1rot", "M2rot"):
        res = [[] for _ in handles]
        for hi, (lay, f) in enumerate(handles):
            for o, v in opts:
                f.set_option(o, v)
            fn, bpp = legs_of(f)[leg]
            fn(); fn()
        for r in range(5):
            for hi, (lay, f) in enumerate(handles):
                fn, bpp = legs_of(f)[leg]
                fn()
                res[hi].append(timeit(fn))
        bpp = legs_of(handles[0][1])[leg][1]
        print("  %-6s " % leg + " | ".join("h%d L%d %.4f ms %.3f" % (hi, lay, statistics.median(res[hi]), bpp * n * n / statistics.median(res[hi]) / 1e6 / 8000)
                                          for hi, (lay, f) in enumerate(handles)), flush=True)

# ---- do the CALLER's planes gain from the same layout?  g / h of the fused steer and the 8 outputs of the pipeline as rows of
# ONE block ([row][plane][column], strided views) against separate allocations; interleaved-state handles only
gh = torch.empty((n, 2, n), device="cuda")
gi, hi_ = gh[:, 0, :], gh[:, 1, :]
o8 = torch.empty((n, 8, n), device="cuda")
outs_i = [o8[:, k, :] for k in range(8)]
print("== caller planes: separate allocations vs rows of one block (config w504_10 / plain10)", flush=True)
for cname in ("w504_10", "plain10", "xcdcol10"):
    for lay, f in handles:
        if lay != 1:
            continue
        for o, v in cfgs[cname]:
            f.set_option(o, v)
        legs = {"M2 separate": (lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h)), 40),
                "M2 one block": (lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(gi, hi_)), 40),
                "M5 separate": (lambda: f.pipeline(img, out=outs), 84),
                "M5 one block": (lambda: f.pipeline(img, out=outs_i), 84)}
        res = {k: [] for k in legs}
        for k, (fn, bpp) in legs.items():
            fn(); fn()
        for r in range(5):
            for k, (fn, bpp) in legs.items():
                fn()
                res[k].append(timeit(fn))
        print("  %-9s " % cname + " | ".join("%s %.4f ms %.3f" % (k, statistics.median(v), legs[k][1] * n * n / statistics.median(v) / 1e6 / 8000) for k, v in res.items()), flush=True)
        break
