"""round 5 (repeated after the packed arithmetic, which needs fewer VGPRs again): workgroups per CU (dynamic LDS nobody touches, CVS_OPTS wgcap=N) per entry point, after the LDS-DMA input path (fewer VGPRs:
5-6 workgroups per CU fit where 4-5 did).  Tuner off, same handle, alternating settings, 3 rounds; run it in several processes."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
def timeit(fn, steps, lead=40):
    for _ in range(lead): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
caps = ["", "wgcap=5", "wgcap=4", "wgcap=3", "wgcap=2"]
def ab(name, fn, nbytes, steps=24):
    res = {}
    for rnd in range(3):
        for st in caps:
            os.environ["CVS_OPTS"] = ("autotune=0," + st).rstrip(",")
            res.setdefault(st, []).append(timeit(fn, steps))
    print(name.ljust(40) + " | ".join("%s %.3f" % (st or "none", nbytes / statistics.median(res[st]) / 8e9) for st in caps), flush=True)
os.environ["CVS_OPTS"] = "autotune=0"
n = 4096
imgs = [torch.rand((n, n), device="cuda") for _ in range(8)]
g, h = cv.alloc_planes(2, n, n, device="cuda")
outs8 = cv.alloc_planes(8, n, n, device="cuda")
f = cv.SteerableFiltersG2(None, 4, 0.67)
k = [0]
def nxt():
    k[0] = (k[0] + 1) & 7
    return imgs[k[0]]
ab("M1 resident", lambda: f.setup(imgs[0], flags=cv.SETUP_BASIS), 32 * n * n)
ab("M2 resident", lambda: f.setup_steer(imgs[0], 0.3, flags=cv.SETUP_BASIS, out=(g, h)), 40 * n * n)
ab("M4 resident", lambda: f.setup(imgs[0], flags=cv.SETUP_FULL), 52 * n * n)
ab("M5 resident", lambda: f.pipeline(imgs[0], out=outs8), 84 * n * n, 16)
ab("M1 new images", lambda: f.setup(nxt(), flags=cv.SETUP_BASIS), 32 * n * n)
ab("M2 new images", lambda: f.setup_steer(nxt(), 0.3, flags=cv.SETUP_BASIS, out=(g, h)), 40 * n * n)
ab("M4 new images", lambda: f.setup(nxt(), flags=cv.SETUP_FULL), 52 * n * n)
f4 = cv.SteerableFiltersG4(None, 6, 0.5)
ab("M6 G4 basis resident", lambda: f4.setup(imgs[0]), 48 * n * n, 16)
ab("M6 G4 + steer resident", lambda: f4.setup_steer(imgs[0], 0.3, out=(g, h)), 56 * n * n, 16)
del imgs, outs8
big = [torch.rand((8192, 8192), device="cuda") for _ in range(2)]
gb, hb = torch.empty_like(big[0]), torch.empty_like(big[0])
fb = cv.SteerableFiltersG2(None, 4, 0.67)
fl = [0]
def nb():
    fl[0] ^= 1
    return big[fl[0]]
ab("M1 8192 resident", lambda: fb.setup(big[0], flags=cv.SETUP_BASIS), 32 * 8192 * 8192, 8)
ab("M2 8192 resident", lambda: fb.setup_steer(big[0], 0.3, flags=cv.SETUP_BASIS, out=(gb, hb)), 40 * 8192 * 8192, 8)
ab("M2 8192 new images", lambda: fb.setup_steer(nb(), 0.3, flags=cv.SETUP_BASIS, out=(gb, hb)), 40 * 8192 * 8192, 8)
ab("M1 8192 new images", lambda: fb.setup(nb(), flags=cv.SETUP_BASIS), 32 * 8192 * 8192, 8)
del big, gb, hb, fb
frames = torch.rand((32, 1080, 1920), device="cuda")
ff = cv.SteerableFiltersG2(None, 4, 0.67)
fo8 = torch.empty((32, 8, 1080, 1920), device="cuda")
fo3 = torch.empty((32, 3, 1080, 1920), device="cuda")
fpix = 32 * 1080 * 1920
ab("C4 32x1080p state kept", lambda: ff.pipeline_batch(frames, out=fo8), 84 * fpix, 8)
ff.set_persist(False)
ab("C4 32x1080p three maps", lambda: ff.pipeline_batch(frames, out=fo3, outputs=(5, 6, 7)), 16 * fpix, 8)
del frames, fo8, fo3, ff
bigs = [torch.rand((8192, 8192), device="cuda") for _ in range(2)]
fp3 = cv.SteerableFiltersG2(None, 4, 0.67)
lv = fp3.pyramid(bigs[0], 5)
ppix = sum(l.shape[0] * l.shape[1] for l in lv)
hp = [cv.SteerableFiltersG2(None, 4, 0.67) for _ in lv]
def pyr():
    fl[0] ^= 1
    cv.pyramid_setup(hp, bigs[fl[0]], level_images=lv[1:], flags=cv.SETUP_BASIS)
ab("C3 pyramid 8192 5 levels", pyr, 32 * ppix + 4 * (ppix - 8192 * 8192), 8)
