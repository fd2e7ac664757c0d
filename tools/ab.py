#!/usr/bin/env python3
"""tools/ab.py -- interleaved A/B rounds in ONE process (cdna guide rule 24): option values x legs."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("CVS_PLACEMENT_SEARCH", "0"); _os.environ.setdefault("CVS_AUTOTUNE", "0")  # A/B runs compare like with like
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

def timeit(fn, steps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps

def main():
    # either: ab.py OPT v1,v2,...      (one option, several values)
    # or:     ab.py cfg "8=0" "8=1,10=0" ...   (several option sets, "opt=value,opt=value")
    n = int(os.environ.get("AB_N", "4096"))
    if len(sys.argv) > 1 and sys.argv[1] == "cfg":
        values = sys.argv[2:]
        cfgs = {v: [tuple(int(x) for x in kv.split("=")) for kv in v.split(",") if kv] for v in values}
    else:
        opt = int(sys.argv[1]) if len(sys.argv) > 1 else L.OPT_BLOCK_ORDER
        values = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1]
        cfgs = {v: [(opt, v)] for v in values}
    img = torch.rand((n, n), device="cuda")
    g, h = torch.empty_like(img), torch.empty_like(img)
    outs = [torch.empty_like(img) for _ in range(8)]
    hs = {}
    g4 = os.environ.get("AB_KIND", "2") == "4"
    for v in values:
        f = cv.SteerableFiltersG4(None) if g4 else cv.SteerableFiltersG2(None)
        for o, val in cfgs[v]:
            f.set_option(o, val)
        hs[v] = f
    legs = {
        "M1 basis": (lambda f: f.setup(img, flags=cv.SETUP_BASIS), 32),
        "M2 +steer": (lambda f: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h)), 40),
        "M4 full": (lambda f: f.setup(img, flags=cv.SETUP_FULL), 52),
        "M5 pipeline": (lambda f: f.pipeline(img, out=outs), 84),
    }
    if g4:
        legs = {"M6 basis": (lambda f: f.setup(img), 48), "M6 +steer": (lambda f: f.setup_steer(img, 0.3, out=(g, h)), 56)}
    for name, (fn, bpp) in legs.items():
        res = {v: [] for v in values}
        for v in values:
            for _ in range(5): fn(hs[v])
        torch.cuda.synchronize()
        for r in range(10):
            for v in values:
                res[v].append(timeit(lambda: fn(hs[v])))
        line = "%-12s" % name
        for v in values:
            med, mn = statistics.median(res[v]), min(res[v])
            line += " | %s med %.4f ms (%5.1f%%) min %.4f" % (v, med, bpp * n * n / med / 1e6 / 80, mn)
        print(line, flush=True)

if __name__ == "__main__":
    main()
