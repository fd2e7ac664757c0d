"""round 5: a card that has been idle for 30 ms runs the next launches at the shader clock the power management had
dropped to.  Does a short "wake" in front of the timed launches bring the clock back -- k launches of the (cheap, memory-bound) basis
pass on the image, issued right after the pause?  20 timed launches of the pipeline / the G4 bank / the fused steer; fraction of HBM."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
n = 4096
img = torch.rand((n, n), device="cuda")
g, h = cv.alloc_planes(2, n, n, device="cuda")
outs = cv.alloc_planes(8, n, n, device="cuda")
f, fw, f4 = cv.SteerableFiltersG2(None), cv.SteerableFiltersG2(None), cv.SteerableFiltersG4(None)
legs = (("M5 pipeline", 84, lambda: f.pipeline(img, out=outs)), ("M6 G4 basis", 48, lambda: f4.setup(img)), ("M2 fused steer", 40, lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))))
wake = lambda: fw.setup(img, flags=cv.SETUP_BASIS)
for _ in range(100):
    for _, _, fn in legs: fn()
    wake()
torch.cuda.synchronize()
def region(fn, idle, k):
    for _ in range(60): fn()
    torch.cuda.synchronize()
    if idle: time.sleep(idle)
    for _ in range(k): wake()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20
for name, bpp, fn in legs:
    row = []
    for idle, k in ((0, 0), (0.03, 0), (0.03, 1), (0.03, 3), (0.03, 10), (0.03, 30)):
        ms = statistics.median(region(fn, idle, k) for _ in range(7))
        row.append("%s wake %2d: %.3f" % ("led in," if not idle else "30 ms idle,", k, bpp * n * n / ms / 8e9))
    print(name.ljust(16) + " | ".join(row), flush=True)
