#!/usr/bin/env python3
"""tools/k1_timeline.py -- where does one basis launch spend its time?  Uses the DIAGNOSTIC twin of the
library (make -C cvsteer_amd/csrc diag -> tools/libcvsteer_hip_diag.so) whose kernels stamp, per wave,
{start, first store, end} with the 100 MHz real-time counter.  Never quote its run time; read its shape."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CVSTEER_HIP_LIB"] = os.path.join(ROOT, "tools", "libcvsteer_hip_diag.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

n = 4096
sr = int(sys.argv[1]) if len(sys.argv) > 1 else 19
order = int(sys.argv[2]) if len(sys.argv) > 2 else 0
xw = int(sys.argv[3]) if len(sys.argv) > 3 else 0
img = torch.rand((n, n), device="cuda")
f = cv.SteerableFiltersG2(None)
f.set_strip_rows(sr)
f.set_option(L.OPT_BLOCK_ORDER, order)
lib = cv.lib()
lib.cvs_diag_set_buffer.argtypes = [C.c_void_p, C.c_void_p]
bands = (n + sr - 1) // sr
nwaves = bands * 16 * 4
buf = torch.zeros((nwaves, 4), dtype=torch.int64, device="cuda")
for flags, name in ((cv.SETUP_BASIS, "M1 basis"), (cv.SETUP_FULL, "M4 full")):
    for _ in range(3): f.setup(img, flags=flags)
    torch.cuda.synchronize()
    lib.cvs_diag_set_buffer(f._h, C.c_void_p(buf.data_ptr()))
    buf.zero_()
    f.setup(img, flags=flags)
    torch.cuda.synchronize()
    lib.cvs_diag_set_buffer(f._h, None)
    raw = buf.cpu().numpy()
    used = raw[:, 0] != 0                                # dynamic order: tall bands leave some slots unused
    xcc = raw[used, 3]
    t = raw[used].astype(np.float64) * 0.01              # 100 MHz ticks -> microseconds
    t0 = t[:, 0].min()
    start, first, end = t[:, 0] - t0, t[:, 1] - t0, t[:, 2] - t0
    total = end.max()
    print("%s  strip_rows=%d  waves=%d  kernel span %.1f us" % (name, sr, nwaves, total))
    print("   wave start      : p0 %.1f  p50 %.1f  p99 %.1f  max %.1f us" % tuple(np.percentile(start, [0, 50, 99, 100])))
    print("   first store-start: p0 %.1f  p50 %.1f  p99 %.1f us (window priming incl. first loads)" % tuple(np.percentile(first - start, [0, 50, 99])))
    print("   wave lifetime   : p1 %.1f  p50 %.1f  p99 %.1f us" % tuple(np.percentile(end - start, [1, 50, 99])))
    edges = np.linspace(0, total, 21)
    occ = [(np.minimum(end, edges[i + 1]) - np.maximum(start, edges[i])).clip(0).sum() / (edges[1] - edges[0]) for i in range(20)]
    sto = [(np.minimum(end, edges[i + 1]) - np.maximum(first, edges[i])).clip(0).sum() / (edges[1] - edges[0]) for i in range(20)]
    print("   resident waves per 5%% slice : " + " ".join("%4.0f" % o for o in occ))
    print("   of which past priming       : " + " ".join("%4.0f" % o for o in sto))
    # workgroup w = by * 16 + bx lands on XCD w % 8 (round-robin dispatch): does every XCD finish at the same time?
    xcd = xcc                                            # stamped from HW_REG_XCC_ID
    print("   strips per XCD     : " + " ".join("%5d" % (xcd == k).sum() for k in range(8)))
    colblk = (np.arange(nwaves) % 64 // 4)[used]         # 256-column block of the strip
    life = end - start
    print("   mean life by column block: " + " ".join("%4.1f" % life[colblk == k].mean() for k in range(16)))
    print("   per-XCD   last end : " + " ".join("%5.1f" % end[xcd == k].max() for k in range(8)))
    print("   per-XCD last start : " + " ".join("%5.1f" % start[xcd == k].max() for k in range(8)))
    print("   per-XCD mean life  : " + " ".join("%5.1f" % (end - start)[xcd == k].mean() for k in range(8)))
