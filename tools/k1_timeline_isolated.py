#!/usr/bin/env python3
"""tools/k1_timeline_isolated.py [strip_rows] -- the fused filter + steer launch on a FRESH image, once after an idle
device ("one object per image") and once at the end of a queue of launches, with per-wave stamps (diagnostic twin of the
library, see k1_timeline.py).  Prints how many waves are resident / past priming in each 5 % slice of the launch."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CVSTEER_HIP_LIB"] = os.path.join(ROOT, "tools", "libcvsteer_hip_diag.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

n = 4096
sr = int(sys.argv[1]) if len(sys.argv) > 1 else 10
imgs = [torch.rand((n, n), device="cuda") for _ in range(8)]
f = cv.SteerableFiltersG2(None)
f.set_option(L.OPT_AUTOTUNE, 0)
f.set_strip_rows(sr)
lib = cv.lib()
lib.cvs_diag_set_buffer.argtypes = [C.c_void_p, C.c_void_p]
bands = (n + sr - 1) // sr
nwaves = bands * 16 * 4
buf = torch.zeros((nwaves, 4), dtype=torch.int64, device="cuda")
g, h = torch.empty_like(imgs[0]), torch.empty_like(imgs[0])


def report(name):
    raw = buf.cpu().numpy()
    used = raw[:, 0] != 0
    t = raw[used].astype(np.float64) * 0.01
    t0 = t[:, 0].min()
    start, first, end = t[:, 0] - t0, t[:, 1] - t0, t[:, 2] - t0
    total = end.max()
    print("%s  strip_rows=%d  waves=%d  kernel span %.1f us" % (name, sr, int(used.sum()), total))
    print("   wave start      : p0 %.1f  p25 %.1f p50 %.1f  p99 %.1f  max %.1f us" % tuple(np.percentile(start, [0, 25, 50, 99, 100])))
    print("   priming (start -> first store): p1 %.1f  p50 %.1f  p99 %.1f us" % tuple(np.percentile(first - start, [1, 50, 99])))
    print("   wave lifetime   : p1 %.1f  p50 %.1f  p99 %.1f us" % tuple(np.percentile(end - start, [1, 50, 99])))
    order = np.argsort(start)
    q = len(order) // 4
    life = end - start
    print("   lifetime by start quartile: " + " ".join("%.1f" % life[order[i * q:(i + 1) * q]].mean() for i in range(4)))
    edges = np.arange(0, total + 5, 5.0)
    occ = [(np.minimum(end, edges[i + 1]) - np.maximum(start, edges[i])).clip(0).sum() / 5.0 for i in range(len(edges) - 1)]
    sto = [(np.minimum(end, edges[i + 1]) - np.maximum(first, edges[i])).clip(0).sum() / 5.0 for i in range(len(edges) - 1)]
    done = [((end > edges[i]) & (end <= edges[i + 1])).sum() for i in range(len(edges) - 1)]
    print("   resident waves per 5 us : " + " ".join("%4.0f" % o for o in occ))
    print("   past priming            : " + " ".join("%4.0f" % o for o in sto))
    print("   strips finished         : " + " ".join("%4d" % o for o in done))


k = 0
for _ in range(6):
    k += 1
    f.setup_steer(imgs[k % 8], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
torch.cuda.synchronize()
for rep in range(2):
    time.sleep(0.002)
    lib.cvs_diag_set_buffer(f._h, C.c_void_p(buf.data_ptr()))
    buf.zero_()
    torch.cuda.synchronize()
    time.sleep(0.001)
    k += 1
    f.setup_steer(imgs[k % 8], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
    torch.cuda.synchronize()
    lib.cvs_diag_set_buffer(f._h, None)
    report("ISOLATED (idle device before)")
for rep in range(2):
    buf.zero_()
    torch.cuda.synchronize()
    for _ in range(4):
        k += 1
        f.setup_steer(imgs[k % 8], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
    lib.cvs_diag_set_buffer(f._h, C.c_void_p(buf.data_ptr()))
    k += 1
    f.setup_steer(imgs[k % 8], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
    lib.cvs_diag_set_buffer(f._h, None)
    k += 1
    f.setup_steer(imgs[k % 8], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
    torch.cuda.synchronize()
    report("QUEUED (fifth of six launches)")
