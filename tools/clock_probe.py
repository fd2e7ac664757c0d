#!/usr/bin/env python3
"""tools/clock_probe.py -- which shader clock do the basis kernels run at?  Uses a diagnostic twin built with
-DCVS_DIAG_STAMPS -DCVS_DIAG_CLOCK (tools/libcvsteer_hip_diagclk.so): every wave stamps its lifetime in shader-clock
ticks (s_memtime) and in 100 MHz real-time ticks (s_memrealtime); the ratio is the clock the wave saw.  The launch is
repeated back to back first, so the chip is in its loaded power state.  Never quote the run time of this build."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CVSTEER_HIP_LIB"] = os.path.join(ROOT, "tools", "libcvsteer_hip_diagclk.so")
os.environ.setdefault("CVS_PLACEMENT_SEARCH", "0")
sys.path.insert(0, ROOT)
import numpy as np, torch
import cvsteer_amd as cv

n = 4096
img = torch.rand((n, n), device="cuda")
zero = torch.zeros((n, n), device="cuda")
lib = cv.lib()
lib.cvs_diag_set_buffer.argtypes = [C.c_void_p, C.c_void_p]
g, h = torch.empty_like(img), torch.empty_like(img)
outs = [torch.empty_like(img) for _ in range(8)]
buf = torch.zeros((1 << 17, 4), dtype=torch.int64, device="cuda")
f2, f4 = cv.SteerableFiltersG2(None), cv.SteerableFiltersG4(None)
legs = [("G2 basis+steer", f2, lambda x: f2.setup_steer(x, 0.3, flags=cv.SETUP_BASIS, out=(g, h))),
        ("G2 pipeline", f2, lambda x: f2.pipeline(x, out=outs)),
        ("G4 basis", f4, lambda x: f4.setup(x)),
        ("G4 basis+steer", f4, lambda x: f4.setup_steer(x, 0.3, out=(g, h)))]
for name, f, fn in legs:
    for data, src in (("random", img), ("zeros", zero)):
        for _ in range(40): fn(src)          # load the chip
        lib.cvs_diag_set_buffer(f._h, C.c_void_p(buf.data_ptr()))
        buf.zero_()
        fn(src)
        torch.cuda.synchronize()
        lib.cvs_diag_set_buffer(f._h, None)
        raw = buf.cpu().numpy()
        used = raw[:, 0] != 0
        real = (raw[used, 2] - raw[used, 0]).astype(np.float64) * 10.0   # ns
        ticks = raw[used, 1].astype(np.float64)
        ok = real > 2000
        mhz = ticks[ok] / real[ok] * 1000.0
        span = (raw[used, 2].max() - raw[used, 0].min()) * 0.01
        print("%-16s %-6s waves %6d  span %7.1f us  clock MHz p5 %.0f p50 %.0f p95 %.0f" % (name, data, used.sum(), span, *np.percentile(mhz, [5, 50, 95])), flush=True)
