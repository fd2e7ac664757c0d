import os, sys, ctypes as C
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/cvsteer_amd") else ".")
import numpy as np, torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
from cvsteer_amd.api import Plane, _plane
lib = cv.lib()
def create(kind, w, s, dev=0):
    h = C.c_void_p()
    rc = lib.cvs_create(kind, w, C.c_float(s), dev, C.byref(h))
    return rc, h
print("create bad kind", create(7, 4, 0.67)[0])
for w in (0, -1, 33, 1000): print("create width", w, create(2, w, 0.67)[0])
for s in (0.0, -1.0, float("nan"), float("inf")): print("create spacing", s, create(2, 4, s)[0])
print("create bad device", create(2, 4, 0.67, 99)[0], create(2, 4, 0.67, -1)[0])
rc, h = create(2, 4, 0.67); print("create ok", rc)
img = torch.rand((40, 50), device="cuda")
g = torch.empty_like(img); hq = torch.empty_like(img)
def P(t, **kw):
    p = _plane(t)
    for k, v in kw.items(): setattr(p, k, v)
    return p
cases = {
 "rows 0": P(img, rows=0), "rows -3": P(img, rows=-3), "cols 0": P(img, cols=0), "null data": P(img, data=0),
 "step too small": P(img, step=50 * 4 - 4), "step % 4": P(img, step=50 * 4 + 2), "mem 99": P(img, mem=99), "mem 0x300": P(img, mem=0x300),
 "misaligned": P(img, data=img.data_ptr() + 2),
}
for name, p in cases.items():
    rc = lib.cvs_setup(h, C.byref(p), cv.SETUP_FULL)
    print("setup %-16s -> %d  %s" % (name, rc, lib.cvs_last_error(h).decode()[:60]))
print("setup NULL plane ->", lib.cvs_setup(h, None, cv.SETUP_FULL))
print("setup NULL handle ->", lib.cvs_setup(None, C.byref(_plane(img)), cv.SETUP_FULL))
print("setup flags 0xffff ->", lib.cvs_setup(h, C.byref(_plane(img)), 0xffff))
pi, pg, ph = _plane(img), _plane(g), _plane(hq)
print("setup_steer ok ->", lib.cvs_setup_steer(h, C.byref(pi), cv.SETUP_BASIS, C.c_float(0.3), C.byref(pg), C.byref(ph)))
print("setup_steer theta nan ->", lib.cvs_setup_steer(h, C.byref(pi), cv.SETUP_BASIS, C.c_float(float("nan")), C.byref(pg), C.byref(ph)))
small = torch.empty((39, 50), device="cuda")
print("setup_steer wrong-size g ->", lib.cvs_setup_steer(h, C.byref(pi), cv.SETUP_BASIS, C.c_float(0.3), C.byref(_plane(small)), C.byref(ph)))
print("setup_steer g aliases image ->", lib.cvs_setup_steer(h, C.byref(pi), cv.SETUP_BASIS, C.c_float(0.3), C.byref(pi), C.byref(ph)))
print("setup_steer g aliases h ->", lib.cvs_setup_steer(h, C.byref(pi), cv.SETUP_BASIS, C.c_float(0.3), C.byref(pg), C.byref(pg)))
# partial overlap of output and input (a row-shifted view of the same buffer)
big = torch.rand((80, 50), device="cuda")
a, b = big[0:40], big[20:60]
print("setup_steer g overlaps image rows ->", lib.cvs_setup_steer(h, C.byref(_plane(a)), cv.SETUP_BASIS, C.c_float(0.3), C.byref(_plane(b)), C.byref(ph)))
for opt, val in ((999, 1), (L.OPT_STRIP_ROWS, -5), (L.OPT_STRIP_ROWS, 10**9), (L.OPT_BLOCK_ORDER, -7), (L.OPT_XCD_WEIGHTS, 99999), (L.OPT_STORE_POLICY, 9), (L.OPT_G4_SPLIT, 7), (L.OPT_ATAN_MODE, 5)):
    print("set_option", opt, val, "->", lib.cvs_set_option(h, opt, val))
# after all that the handle still works
rc = lib.cvs_setup(h, C.byref(_plane(img)), cv.SETUP_FULL); torch.cuda.synchronize(); print("setup after errors ->", rc)
print("destroy ->", lib.cvs_destroy(h), "destroy NULL ->", lib.cvs_destroy(None))
# ---- overlap rules (round 3)
rc, h = create(2, 4, 0.67)
big = torch.rand((80, 120), device="cuda")
left, right = big[:40, 0:50], big[:40, 60:110]          # side-by-side column ranges of one buffer: share nothing
g2 = torch.empty((40, 50), device="cuda"); h2 = torch.empty((40, 50), device="cuda")
print("ROI in, ROI out side by side ->", lib.cvs_setup_steer(h, C.byref(_plane(left)), cv.SETUP_BASIS, C.c_float(0.3), C.byref(_plane(right)), C.byref(_plane(h2))))
torch.cuda.synchronize()
ref_g, ref_h = cv.SteerableFiltersG2(None).setup_steer(left.contiguous(), 0.3)
print("   values equal the contiguous case:", bool(torch.equal(right, ref_g)))
print("ROI columns overlapping by 5 ->", lib.cvs_setup_steer(h, C.byref(_plane(big[:40, 0:50])), cv.SETUP_BASIS, C.c_float(0.3), C.byref(_plane(big[:40, 45:95])), C.byref(_plane(h2))))
print("rows overlapping ->", lib.cvs_setup_steer(h, C.byref(_plane(big[0:40, 0:50])), cv.SETUP_BASIS, C.c_float(0.3), C.byref(_plane(big[20:60, 0:50])), C.byref(_plane(h2))))
print("rows disjoint ->", lib.cvs_setup_steer(h, C.byref(_plane(big[0:40, 0:50])), cv.SETUP_BASIS, C.c_float(0.3), C.byref(_plane(big[40:80, 0:50])), C.byref(_plane(h2))))
print("g is h ->", lib.cvs_setup_steer(h, C.byref(_plane(left)), cv.SETUP_BASIS, C.c_float(0.3), C.byref(_plane(g2)), C.byref(_plane(g2))))
print("setup flags 0xffff ->", lib.cvs_setup(h, C.byref(_plane(left)), 0xffff))
ang = torch.rand((40, 50), device="cuda") * 9
print("wrap in place ->", lib.cvs_wrap(h, C.byref(_plane(ang)), C.byref(_plane(ang))))
buf = torch.rand((41, 50), device="cuda")
print("wrap shifted by one row ->", lib.cvs_wrap(h, C.byref(_plane(buf[0:40])), C.byref(_plane(buf[1:41]))))
hostbuf = np.random.rand(41, 50).astype(np.float32)
print("wrap host shifted by one row ->", lib.cvs_wrap(h, C.byref(_plane(hostbuf[0:40])), C.byref(_plane(hostbuf[1:41]))))
print("wrap host in place ->", lib.cvs_wrap(h, C.byref(_plane(hostbuf[0:40])), C.byref(_plane(hostbuf[0:40]))))
lvl = torch.empty((20, 25), device="cuda")
print("pyr_down ok ->", lib.cvs_pyr_down(h, C.byref(_plane(g2)), C.byref(_plane(lvl))), " onto itself ->", lib.cvs_pyr_down(h, C.byref(_plane(g2)), C.byref(_plane(g2[:20, :25]))))
torch.cuda.synchronize()
