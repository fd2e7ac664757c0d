import os, sys, statistics
sys.path.insert(0, "/root/repo")
import torch, cvsteer_amd as cv
bigs = [torch.rand((8192, 8192), device="cuda") for _ in range(2)]
hp = [cv.SteerableFiltersG2(None) for _ in range(5)]
lv = None
flip = {"i": 0}
def whole():
    global lv
    flip["i"] ^= 1
    lv = cv.pyramid_setup(hp, bigs[flip["i"]], level_images=lv[1:] if lv else None)
def timeit(fn, steps=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps
for _ in range(50): whole()
r = [timeit(whole) for _ in range(7)]
ppix = sum((8192 >> l) ** 2 for l in range(5))
wb = 32 * ppix + 4 * (ppix - 8192 * 8192)
t = statistics.median(r)
print("CVS_PYR_NT=%s whole %.4f ms  %.3f of HBM  (min %.4f)" % (os.environ.get("CVS_PYR_NT", "0"), t, wb / t / 1e6 / 8000, min(r)))
