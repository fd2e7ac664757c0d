#!/usr/bin/env python3
"""tools/order_probe.py -- launch orders side by side on ONE handle per layout (round 4): plain, weighted, XCD columns and the
dynamic order (tiles taken from per-XCD queues), 10- and 19-row strips; legs M1 / M2 / M4 / M5 / rotating / G4 / C4 batch.
Interleaved rounds, median.  usage: order_probe.py [legs...]"""
import os, sys, statistics
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


n = 4096
imgs8 = [torch.rand((n, n), device="cuda") for _ in range(8)]
img = imgs8[0]
g, h = torch.empty_like(img), torch.empty_like(img)
outs = cv.alloc_planes(8, n, n, device="cuda")
cfgs = {"plain10": (0, 10, 0), "w504_10": (1, 10, 504), "xcdcol10": (1000000, 10, 101), "dyn10": (2000000, 10, 0), "plain19": (0, 19, 0)}
want = set(sys.argv[1:])
rot = {"i": 0}


def run(name, f, legs, cfgs, rounds=5, steps=20, npix=n * n):
    for leg, (fn, bpp) in legs.items():
        if want and leg not in want:
            continue
        res = {c: [] for c in cfgs}
        for c, (o, sr, xw) in cfgs.items():
            f.set_option(L.OPT_BLOCK_ORDER, o); f.set_option(L.OPT_STRIP_ROWS, sr); f.set_option(L.OPT_XCD_WEIGHTS, xw)
            fn(); fn()
        for r in range(rounds):
            for c, (o, sr, xw) in cfgs.items():
                f.set_option(L.OPT_BLOCK_ORDER, o); f.set_option(L.OPT_STRIP_ROWS, sr); f.set_option(L.OPT_XCD_WEIGHTS, xw)
                fn(); fn()
                res[c].append(timeit(fn, steps))
        print("%-8s %-6s " % (name, leg) + " | ".join("%s %.4f %.3f" % (c, statistics.median(v), bpp * npix / statistics.median(v) / 1e6 / 8000) for c, v in res.items()), flush=True)


for lay in (1,):
    f = cv.SteerableFiltersG2(None)
    f.set_option(L.OPT_STATE_LAYOUT, lay)

    def rot2():
        rot["i"] = (rot["i"] + 1) & 7
        f.setup_steer(imgs8[rot["i"]], 0.3, flags=cv.SETUP_BASIS, out=(g, h))

    def rot1():
        rot["i"] = (rot["i"] + 1) & 7
        f.setup(imgs8[rot["i"]], flags=cv.SETUP_BASIS)

    legs = {"M1": (lambda: f.setup(img, flags=cv.SETUP_BASIS), 32), "M2": (lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h)), 40),
            "M4": (lambda: f.setup(img, flags=cv.SETUP_FULL), 52), "M5": (lambda: f.pipeline(img, out=outs), 84),
            "M1rot": (rot1, 32), "M2rot": (rot2, 40)}
    run("G2 L%d" % lay, f, legs, cfgs)
    del f
for lay in (0, 1):
    f4 = cv.SteerableFiltersG4(None)
    f4.set_option(L.OPT_STATE_LAYOUT, lay)
    c4 = {"plain40": (0, 40, 0), "w504_40": (1, 40, 504), "dyn40": (2000000, 40, 0), "plain27": (0, 27, 0), "plain53": (0, 53, 0)}
    run("G4 L%d" % lay, f4, {"M6": (lambda: f4.setup(img), 48), "M6s": (lambda: f4.setup_steer(img, 0.3, out=(g, h)), 56)}, c4)
    del f4
del imgs8, outs
nfr = 32
fsets = [torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)]
fout = torch.empty((nfr, 8, 1080, 1920), device="cuda")
fo3 = torch.empty((nfr, 3, 1080, 1920), device="cuda")
alt = {"i": 0}
for lay in (0, 1):
    ff = cv.SteerableFiltersG2(None)
    ff.set_option(L.OPT_STATE_LAYOUT, lay)

    def c4s():
        alt["i"] ^= 1
        ff.pipeline_batch(fsets[alt["i"]], out=fout)

    cb = {"plain10": (0, 10, 0), "w504_10": (1, 10, 504), "dyn10": (2000000, 10, 0), "plain19": (0, 19, 0)}
    run("C4 L%d" % lay, ff, {"C4": (c4s, 84)}, cb, rounds=3, steps=6, npix=nfr * 1080 * 1920)
    ff.set_persist(False)

    def c4f():
        alt["i"] ^= 1
        ff.pipeline_batch(fsets[alt["i"]], out=fo3, outputs=(5, 6, 7))

    run("C4f L%d" % lay, ff, {"C4f": (c4f, 16)}, cb, rounds=3, steps=6, npix=nfr * 1080 * 1920)
    del ff
