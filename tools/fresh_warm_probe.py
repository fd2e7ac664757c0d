"""round 5: launches on NEW images -- every wave also touches the first 2W+1 rows of the tile `warm` row bands further down (the
tile expected to follow it on its CU), CVS_OPTS warm=N, against off and against the read-ahead pass.  8 rotating 4096^2 images
(M2) and 2 rotating 8192^2 images; tuner off; interleaved rounds on one box."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
def run(n, nimg, warms, steps, kind=2):
    imgs = [torch.rand((n, n), device="cuda") for _ in range(nimg)]
    g, h = cv.alloc_planes(2, n, n, device="cuda")
    f = cv.SteerableFiltersG2(None, 4, 0.67) if kind == 2 else cv.SteerableFiltersG4(None, 6, 0.5)
    f.set_option(L.OPT_AUTOTUNE, 0)
    bpp = 40 if kind == 2 else 56
    ref = None
    res = {}
    k = [0]
    def step():
        k[0] = (k[0] + 1) % nimg
        f.setup_steer(imgs[k[0]], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
    for rnd in range(4):
        for w in warms:
            os.environ["CVS_OPTS"] = ("warm=%d,warm_k=%d" % w) if w[0] >= 0 else "read_ahead_force=1"
            for _ in range(40): step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(steps): step()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(w, []).append(e0.elapsed_time(e1) / steps)
            if rnd == 0:
                k[0] = 0; step(); cur = (g.clone(), h.clone(), f.basis(3).clone())
                if ref is None: ref = cur
                assert all(torch.equal(a, b) for a, b in zip(cur, ref)), w
    for w in warms:
        ms = statistics.median(res[w])
        print("kind %d %d^2, %d rotating images, cold bands %3d k %2d : %.4f ms %.3f of HBM  (%s)" % (kind, n, nimg, w[0], w[1], ms, bpp * n * n / ms / 8e9, " ".join("%.3f" % (bpp * n * n / x / 8e9) for x in res[w])), flush=True)
# 8192^2: 820 bands, 32 KiB per row, 320 KiB per band: which share of the image to warm, from how many cold bands
run(8192, 2, [(0, 0), (160, 3), (205, 3), (164, 4), (117, 4), (82, 6), (82, 9), (41, 12)], 10)
run(6144, 4, [(0, 0), (62, 9), (103, 5), (154, 3), (205, 2)], 12)
run(4096, 8, [(0, 0), (41, 9), (82, 4)], 24)
