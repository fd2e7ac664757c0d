#!/usr/bin/env python3
"""tools/zways_probe.py -- frame batches (32 x 1080p, state kept / three maps only): frames dispatched in order against
frames dealt from 2 / 4 / 8 / 16 equal parts of the batch in turn (CVS_BATCH_WAYS), on the SAME handles and buffers,
interleaved rounds; several handles = several state blocks of the allocation lottery."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVS_AUTOTUNE"] = "0"
import torch
import cvsteer_amd as cv

nfr = 32
sets = [torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)]
pix = nfr * 1080 * 1920
fo3 = torch.empty((nfr, 3, 1080, 1920), device="cuda")
fo8 = torch.empty((nfr, 8, 1080, 1920), device="cuda")
alt = {"i": 0}


def timeit(fn, steps=6, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


ways = [1, 2, 4, 8, 16]
for persist, out, sel, bpp in ((True, fo8, None, 84), (False, fo3, (5, 6, 7), 16)):
    for hnd in range(3 if persist else 1):
        f = cv.SteerableFiltersG2(None)
        f.set_persist(persist)

        def run():
            alt["i"] ^= 1
            f.pipeline_batch(sets[alt["i"]], out=out, outputs=sel)

        res = {w: [] for w in ways}
        for rnd in range(5):
            for w in ways:
                os.environ["CVS_BATCH_WAYS"] = str(w)
                res[w].append(timeit(run))
        os.environ["CVS_BATCH_WAYS"] = "1"
        print("persist=%d handle %d: " % (persist, hnd) + " | ".join(
            "ways %2d %.4f ms %.3f" % (w, sorted(res[w])[2], bpp * pix / sorted(res[w])[2] / 1e6 / 8000) for w in ways), flush=True)
        if persist:
            keep = f   # keep the block allocated so that the next handle gets another one
