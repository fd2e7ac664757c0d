#!/usr/bin/env python3
"""tools/alloc_modes.py -- does the speed of the many-plane kernels depend on WHERE the state block was allocated?
Several handles (each with its own state allocation), same image, same pinned block order; interleaved timing.

Measured on MI355X (round 1): on some boxes a few of the allocations run the 9..20-plane variants at 81-84 % of
the HBM roofline while the others stay at 67-71 %, for the same kernel, order and image; the 7-plane pass does not
show it.  Within one big allocation every offset behaves the same; plane-stride padding (256 B .. 192 MiB),
physically contiguous allocations (hipDeviceMallocContiguous: always the slower mode), power-of-two sizes and
row-interleaved planes do not move a handle from one mode to the other.  On other boxes no allocation is fast.
AB_PREALLOC_MB=n occupies the first n MiB of device memory before anything else is allocated."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("CVS_PLACEMENT_SEARCH", "0"); _os.environ.setdefault("CVS_AUTOTUNE", "0")  # A/B runs compare like with like
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

def timeit(fn, steps=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps

n = 4096
nh = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pre = int(os.environ.get("AB_PREALLOC_MB", "0"))
dummy = torch.empty(pre << 20, dtype=torch.uint8, device="cuda") if pre else None   # occupy the first part of device memory
img = torch.rand((n, n), device="cuda")
hs, pads = [], []
for i in range(nh):
    pads.append(torch.empty(((i * 37) % 11 + 1) << 20, dtype=torch.uint8, device="cuda"))  # perturb the allocator
    f = cv.SteerableFiltersG2(None)
    f.set_option(L.OPT_BLOCK_ORDER, 0)
    f.setup(img, flags=cv.SETUP_FULL)
    hs.append(f)
torch.cuda.synchronize()
g, hq = torch.empty_like(img), torch.empty_like(img)
outs = [torch.empty_like(img) for _ in range(8)]
legs = [("M1", lambda f: f.setup(img, flags=cv.SETUP_BASIS), 32), ("M2", lambda f: f.setup_steer(img, 0.3, out=(g, hq)), 40),
        ("M4", lambda f: f.setup(img, flags=cv.SETUP_FULL), 52), ("M5", lambda f: f.pipeline(img, out=outs), 84)]
print("state blocks at: " + " ".join("0x%x" % C.cast(f.basis_view(0)[0], C.c_void_p).value for f in hs))
for name, fn, bpp in legs:
    res = [[] for _ in hs]
    for f in hs: fn(f)
    for r in range(5):
        for i, f in enumerate(hs):
            res[i].append(timeit(lambda: fn(f)))
    print(name + " % of 8 TB/s per handle: " + "  ".join("%.1f" % (bpp * n * n / statistics.median(r) / 1e6 / 80) for r in res), flush=True)
