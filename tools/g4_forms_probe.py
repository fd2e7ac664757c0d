"""round 5: the G4 bank after the LDS-DMA input path -- both half banks in one launch (split 2, the product's) against the single
11-plane kernel (split 0: one staging of every row, the image fetched once; 158 VGPRs = three waves per SIMD now, two before),
strip heights and launch orders, basis pass and fused steer at 4096^2.  Same process, handles created under CVS_OPTS g4_split=...,
interleaved rounds; HIP events around 20 launches after a 20 ms lead-in."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
n = 4096
img = torch.rand((n, n), device="cuda")
g, h = torch.empty_like(img), torch.empty_like(img)
cfgs = [(2, 40, 0), (0, 40, 0), (0, 27, 0), (0, 53, 0), (2, 53, 0), (0, 66, 0), (2, 40, 2000000), (0, 40, 2000000), (0, 53, 2000000), (0, 40, 1000000), (2, 40, 1000000)]
hs = {}
for split, strip, order in cfgs:
    os.environ["CVS_OPTS"] = "g4_split=%d,autotune=0" % split
    f = cv.SteerableFiltersG4(None, 6, 0.5)
    f.set_strip_rows(strip); f.set_option(L.OPT_BLOCK_ORDER, order)
    hs[(split, strip, order)] = f
ref = None
def timeit(fn, steps=20):
    for _ in range(150): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
res = {}
for rnd in range(3):
    for key, f in hs.items():
        res.setdefault((key, "basis"), []).append(timeit(lambda: f.setup(img)))
        res.setdefault((key, "steer"), []).append(timeit(lambda: f.setup_steer(img, 0.3, out=(g, h))))
        if rnd == 0:
            cur = [f.basis(p).clone() for p in (0, 4, 5, 10)] + [g.clone(), h.clone()]
            if ref is None: ref = cur
            assert all(torch.equal(a, b) for a, b in zip(cur, ref)), key
            assert f.launch_info()["g4_split"] == key[0]
for key in hs:
    b, s = statistics.median(res[(key, "basis")]), statistics.median(res[(key, "steer")])
    print("split %d strip %2d order %7d : basis %.4f ms %.3f | +steer %.4f ms %.3f   (rounds: %s)" % (key + (b, 48 * n * n / b / 8e9, s, 56 * n * n / s / 8e9,
          " ".join("%.3f" % (48 * n * n / x / 8e9) for x in res[(key, "basis")]))), flush=True)
