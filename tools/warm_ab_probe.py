"""round 5: what requesting the rest of a NEW image ahead of need (CVS_OPTS warm=K, default 4) is worth on the bench's legs that
run on new images -- config 3 (5-level pyramid of alternating 8192^2 images, one native call), one object per image, the fused
steer on rotating images, G4 -- same process, same handles, alternating settings; fraction of the HBM roofline."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
dev = "cuda"
def timeit(fn, steps, lead=30):
    for _ in range(lead): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
def ab(name, fn, nbytes, steps, settings=(0, 4, 2, 8)):
    res = {}
    for rnd in range(3):
        for w in settings:
            os.environ["CVS_OPTS"] = "warm=%d,autotune=0" % w
            res.setdefault(w, []).append(timeit(fn, steps))
    print(name.ljust(44) + " | ".join("warm %d: %.4f ms %.3f" % (w, statistics.median(res[w]), nbytes / statistics.median(res[w]) / 8e9) for w in settings), flush=True)
os.environ["CVS_OPTS"] = "autotune=0"
# config 3
bigs = [torch.rand((8192, 8192), device=dev) for _ in range(2)]
fp3 = cv.SteerableFiltersG2(None, 4, 0.67)
lv = fp3.pyramid(bigs[0], 5)
ppix = sum(l.shape[0] * l.shape[1] for l in lv)
hp = [cv.SteerableFiltersG2(None, 4, 0.67) for _ in lv]
flip = [0]
def pyr():
    flip[0] ^= 1
    cv.pyramid_setup(hp, bigs[flip[0]], level_images=lv[1:], flags=cv.SETUP_BASIS)
ab("C3 pyramid 8192^2 x 5 levels, one call", pyr, 32 * ppix + 4 * (ppix - 8192 * 8192), 10)
def lvl0():
    flip[0] ^= 1
    hp[0].setup_pyr(bigs[flip[0]], flags=cv.SETUP_BASIS, out=lv[1])
ab("   level 0 alone (filter + next level)", lvl0, 33 * 8192 * 8192, 10)
def lvl0m1():
    flip[0] ^= 1
    hp[0].setup(bigs[flip[0]], flags=cv.SETUP_BASIS)
ab("   M1 on alternating 8192^2 images", lvl0m1, 32 * 8192 * 8192, 10)
del bigs, lv, hp, fp3
n = 4096
imgs = [torch.rand((n, n), device=dev) for _ in range(8)]
g, h = cv.alloc_planes(2, n, n, device=dev)
k = [0]
def objs():
    k[0] = (k[0] + 1) & 7
    fo = cv.SteerableFiltersG2(None, 4, 0.67)
    fo.setup_steer(imgs[k[0]], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
    del fo
ab("M2 one object per image (8 images)", objs, 40 * n * n, 48)
f = cv.SteerableFiltersG2(None, 4, 0.67)
def rot():
    k[0] = (k[0] + 1) & 7
    f.setup_steer(imgs[k[0]], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
ab("M2 rotating 8 images, one handle", rot, 40 * n * n, 24)
outs8 = cv.alloc_planes(8, n, n, device=dev)
def pipe():
    k[0] = (k[0] + 1) & 7
    f.pipeline(imgs[k[0]], out=outs8)
ab("M5 pipeline on rotating images", pipe, 84 * n * n, 16)
def full():
    k[0] = (k[0] + 1) & 7
    f.setup(imgs[k[0]], flags=cv.SETUP_FULL)
ab("M4 full setup on rotating images", full, 52 * n * n, 24)
f4 = cv.SteerableFiltersG4(None, 6, 0.5)
def g4():
    k[0] = (k[0] + 1) & 7
    f4.setup(imgs[k[0]])
ab("M6 G4 basis on rotating images", g4, 48 * n * n, 16)
u8 = [(im * 255).to(torch.uint8) for im in imgs]
def rot8():
    k[0] = (k[0] + 1) & 7
    f.setup_steer(u8[k[0]], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
ab("M2 rotating 8-bit images (16 MiB each)", rot8, 37 * n * n, 24)
