#!/usr/bin/env python3
"""tools/warm_drift_probe.py -- does the headline launch speed up over the first seconds of a process?  (bench.py: the headline, timed
first, reads 0-2 % lower than the same configuration on the same handle a few seconds later.)  Back-to-back launches from the first call
on; event time of every chunk of 100 launches, printed against the time since the first launch, plus shader clock / power where readable."""
import os, sys, time, glob
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

n = 4096
LEG = os.environ.get("DRIFT_LEG", "M2")   # M1 = the basis pass alone
BPP = 32 if LEG == "M1" else 40
img = torch.rand((n, n), device="cuda")
g, h = cv.alloc_planes(2, n, n, device="cuda")
f = cv.SteerableFiltersG2(None)
f.set_option(L.OPT_AUTOTUNE, 0)
pr = torch.cuda.get_device_properties(0)
want = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
cards = [c for c in glob.glob("/sys/class/drm/card*/device") if want in os.path.realpath(c)]
clk = (sorted(glob.glob(os.path.join(cards[0], "hwmon/hwmon*/freq1_input"))) or [None])[0] if cards else None
pw = (sorted(glob.glob(os.path.join(cards[0], "hwmon/hwmon*/power1_input"))) or [None])[0] if cards else None
rd = lambda p: int(open(p).read().split()[0]) if p else 0
torch.cuda.synchronize()
time.sleep(0.5)
t0 = time.perf_counter()
rows = []
while time.perf_counter() - t0 < 6.0:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100):
        if LEG == "M1":
            f.setup(img, flags=cv.SETUP_BASIS)
        else:
            f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))
    b.record()
    torch.cuda.synchronize()
    rows.append((time.perf_counter() - t0, a.elapsed_time(b) / 100, rd(clk) / 1e6, rd(pw) / 1e6))
last = 0.0
for t, ms, c, p in rows:
    if t - last >= 0.25 or t < 0.1:
        print("t = %5.2f s  %.4f ms  %.3f   sclk %4.0f MHz  %4.0f W" % (t, ms, BPP * n * n / (ms * 1e-3) / 8e12, c, p))
        last = t
