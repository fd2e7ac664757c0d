"""round 5: two strip heights in one launch -- the first P % of the rows in strips of T rows (a wave that walks on
needs no new window priming: the row pass runs 1 + 8/T times per row instead of 1.8), the rest in 10-row strips that fill the end of
the launch -- against 10-row strips throughout, with and without the dynamic tail; and a cap on the workgroups per CU for launches on
new images.  4096^2, tuner off, same handle, alternating settings."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
n = 4096
imgs = [torch.rand((n, n), device="cuda") for _ in range(8)]
g, h = cv.alloc_planes(2, n, n, device="cuda")
outs8 = cv.alloc_planes(8, n, n, device="cuda")
def timeit(fn, steps=24, lead=40):
    for _ in range(lead): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
def ab(name, mk, nbytes, settings, order=0):
    f = cv.SteerableFiltersG2(None, 4, 0.67)
    f.set_option(L.OPT_AUTOTUNE, 0); f.set_option(L.OPT_BLOCK_ORDER, order)
    fn = mk(f)
    res, ref = {}, None
    for rnd in range(3):
        for st in settings:
            os.environ["CVS_OPTS"] = st
            res.setdefault(st, []).append(timeit(fn))
            if rnd == 0:
                cur = [g.clone(), f.basis(2).clone(), f.basis(6).clone()]
                if ref is None: ref = cur
                assert all(torch.equal(a, b) for a, b in zip(cur, ref)), st
    print(name)
    for st in settings:
        m = statistics.median(res[st])
        print("   %-28s %.4f ms %.3f  (%s)" % (st or "(default)", m, nbytes / m / 8e9, " ".join("%.3f" % (nbytes / x / 8e9) for x in res[st])), flush=True)
k = [0]
tall = ["", "tall=19,tall_pct=80", "tall=37,tall_pct=80", "tall=37,tall_pct=60", "tall=64,tall_pct=80", "tall=37,tall_pct=90", "tall=28,tall_pct=85"]
ab("M1 basis pass, resident image, plain order", lambda f: (lambda: f.setup(imgs[0], flags=cv.SETUP_BASIS)), 32 * n * n, tall)
ab("M1 basis pass, resident image, dynamic tail", lambda f: (lambda: f.setup(imgs[0], flags=cv.SETUP_BASIS)), 32 * n * n, tall[:4], order=2000000)
ab("M2 fused steer, resident image", lambda f: (lambda: f.setup_steer(imgs[0], 0.3, flags=cv.SETUP_BASIS, out=(g, h))), 40 * n * n, tall[:5])
ab("M4 full setup, resident image", lambda f: (lambda: f.setup(imgs[0], flags=cv.SETUP_FULL)), 52 * n * n, tall[:4])
ab("M5 pipeline, resident image", lambda f: (lambda: f.pipeline(imgs[0], out=outs8)), 84 * n * n, tall[:4])
caps = ["", "wgcap=5", "wgcap=4", "wgcap=3"]
def rot(f):
    def fn():
        k[0] = (k[0] + 1) & 7
        f.setup_steer(imgs[k[0]], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
    return fn
ab("M2 fused steer on 8 rotating images (requested ahead)", rot, 40 * n * n, caps)
ab("M2 fused steer, resident image, capped", lambda f: (lambda: f.setup_steer(imgs[0], 0.3, flags=cv.SETUP_BASIS, out=(g, h))), 40 * n * n, caps)
ab("M1 basis pass, resident image, capped", lambda f: (lambda: f.setup(imgs[0], flags=cv.SETUP_BASIS)), 32 * n * n, caps)
