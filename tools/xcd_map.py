#!/usr/bin/env python3
"""tools/xcd_map.py -- is the workgroup -> XCD placement stable from launch to launch?  (diagnostic build:
every strip stamps HW_REG_XCC_ID.)  Prints, per launch, the XCD of workgroup 0 and how many workgroups break
the pattern  xcd(b) == (xcd(0) + b) % 8."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CVSTEER_HIP_LIB"] = os.path.join(ROOT, "tools", "libcvsteer_hip_diag.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

n, sr = 4096, 19
img = torch.rand((n, n), device="cuda")
small = torch.rand((777, 1333), device="cuda")
f = cv.SteerableFiltersG2(None)
f.set_strip_rows(sr)
f.set_option(L.OPT_BLOCK_ORDER, 0)
lib = cv.lib()
lib.cvs_diag_set_buffer.argtypes = [C.c_void_p, C.c_void_p]
bands = (n + sr - 1) // sr
nwaves = bands * 16 * 4
buf = torch.zeros((nwaves, 4), dtype=torch.int64, device="cuda")
firsts, breaks = [], []
for it in range(40):
    # unrelated work of varying grid size in between
    for _ in range(it % 3):
        (small * 1.5).sum()
    if it % 4 == 1:
        torch.mm(small[:512, :512], small[:512, :512])
    lib.cvs_diag_set_buffer(f._h, C.c_void_p(buf.data_ptr()))
    buf.zero_()
    f.setup(img, flags=cv.SETUP_BASIS)
    torch.cuda.synchronize()
    lib.cvs_diag_set_buffer(f._h, None)
    x = buf.cpu().numpy()[:, 3].reshape(bands, 16, 4)[:, :, 0]     # XCD of workgroup (by, bx)
    wg = (np.arange(bands)[:, None] * 16 + np.arange(16)[None, :])
    first = int(x[0, 0])
    firsts.append(first)
    breaks.append(int((x != (first + wg) % 8).sum()))
print("XCD of workgroup 0 per launch:", firsts)
print("workgroups off the round-robin pattern per launch:", breaks, "of", bands * 16)
