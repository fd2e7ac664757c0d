#!/usr/bin/env python3
"""tools/headline_position_probe.py -- bench.py's headline reads 0-2 % lower than the same configuration on the same handle later in
the run.  Time, or what happens in between?  bench.py's own _time_steps (lead-in, 15 regions of 20 steps) on the headline launch eight
times in a row with nothing in between, then after allocating and filtering other things (what the secondary legs do), then again."""
import importlib.util, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import torch
import cvsteer_amd as cv
n = 4096
gen = torch.Generator(device="cuda").manual_seed(1234)
img = torch.rand((n, n), generator=gen, device="cuda")
f = cv.SteerableFiltersG2(None, 4, 0.67)
g, h = cv.alloc_planes(2, n, n, device="cuda")
step = lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))
t0 = time.perf_counter()


def measure(tag):
    _w, ev = bench._time_steps(torch, step, 20, 5, lambda: None, repeats=15)
    ev = sorted(v / 20 for v in ev)
    print("t = %5.1f s  %-34s median %.5f ms = %.3f   (min %.5f max %.5f)" % (time.perf_counter() - t0, tag, ev[7], 40 * n * n / (ev[7] * 1e-3) / 8e12, ev[0], ev[-1]), flush=True)


for _ in range(60):
    step()
for k in range(4):
    measure("headline, round %d" % k)
host = torch.empty((n, n), dtype=torch.float32).pin_memory()
for k in range(6):       # what a host-plane leg does: copies over the host link in both directions
    host.copy_(img)
    img2 = host.cuda(non_blocking=True)
torch.cuda.synchronize()
measure("after host <-> device copies")
measure("once more")
imgs = [torch.rand((n, n), generator=gen, device="cuda") for _ in range(7)]
for k in range(200):
    f.setup_steer(imgs[k % 7], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
measure("after 200 launches on new images")
f2 = cv.SteerableFiltersG2(None, 4, 0.67)
for k in range(100):
    f2.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))
measure("after another handle ran")
measure("once more")
