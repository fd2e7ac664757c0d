#!/usr/bin/env python3
"""tools/c4_config_sweep.py -- 32 x 1080p frame batch with state kept, two alternating frame sets (fresh inputs): strip height x
tile deal x order on ONE handle at a time (several handles = several state blocks of the allocation lottery)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cvsteer_amd as cv
from cvsteer_amd import _lib as L
nfr = 32
sets = [torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)]
out = torch.empty((nfr, 8, 1080, 1920), device="cuda")
pix = nfr * 1080 * 1920
def timeit(fn, steps=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps
alt = {"i": 0}
keep = []
for hnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    f = cv.SteerableFiltersG2(None); f.set_option(L.OPT_AUTOTUNE, 0)
    keep.append(f)
    def run():
        alt["i"] ^= 1; f.pipeline_batch(sets[alt["i"]], out=out)
    res = {}
    for rnd in range(3):
        for sr, xw in ((19, 403), (10, 403), (10, 504), (19, 504)):
            f.set_strip_rows(sr); f.set_option(L.OPT_XCD_WEIGHTS, xw)
            for order in (0, 1, 1000000):
                f.set_option(L.OPT_BLOCK_ORDER, order)
                res.setdefault((sr, xw, order), []).append(84 * pix / timeit(run) / 1e6 / 8000)
    print("handle", hnd, " ".join("%s:%.3f" % (k, sorted(v)[1]) for k, v in res.items()), flush=True)
