#!/usr/bin/env python3
"""tools/thread_stress.py [threads] [iterations] -- the reference's usage model under load: many host threads, each creating
its own object per image (example/steer.cpp:69-124 inside cv::parallel_for_), here all at once on one GPU and over a small
set of shapes so that the process-wide caches (launch-order tuner, parked state blocks, the placement search's lock and its
"no window" memory) are contended.  Every result is compared bit for bit with one computed single-threaded beforehand."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
shapes = [(1080, 1920), (1200, 1100), (257, 300), (64, 64), (2048, 2048), (13, 70), (2304, 4096)]   # the last one: a state of 256 MiB and more, i.e. the opt-in placement search runs
gen = torch.Generator(device="cuda").manual_seed(5)
imgs = {s: torch.rand(s, generator=gen, device="cuda") for s in shapes}
ref = {}
for s, x in imgs.items():
    f = cv.SteerableFiltersG2(None)
    g, h = f.setup_steer(x, 0.3, flags=cv.SETUP_FULL)
    outs = cv.SteerableFiltersG2(None).pipeline(x)
    f4 = cv.SteerableFiltersG4(None)
    g4, h4 = f4.setup_steer(x, -0.4)
    ref[s] = (g.clone(), h.clone(), f.basis(3).clone(), f.getDominantOrientationAngle().clone(), [o.clone() for o in outs], g4.clone(), h4.clone(), f4.basis(7).clone())
torch.cuda.synchronize()
errors = []
lock = threading.Lock()


def worker(tid):
    rng = np.random.default_rng(100 + tid)
    stream = torch.cuda.Stream()
    try:
        with torch.cuda.stream(stream):
            for it in range(iters):
                s = shapes[int(rng.integers(0, len(shapes)))]
                x = imgs[s]
                what = int(rng.integers(0, 3))
                search = int(rng.integers(0, 2))
                if what == 0:
                    f = cv.SteerableFiltersG2(None)
                    f.set_option(L.OPT_PLACEMENT_SEARCH, search)
                    g, h = f.setup_steer(x, 0.3, flags=cv.SETUP_FULL)
                    ok = torch.equal(g, ref[s][0]) and torch.equal(h, ref[s][1]) and torch.equal(f.basis(3), ref[s][2]) and torch.equal(f.getDominantOrientationAngle(), ref[s][3])
                elif what == 1:
                    f = cv.SteerableFiltersG2(None)
                    f.set_option(L.OPT_PLACEMENT_SEARCH, search)
                    outs = f.pipeline(x)
                    ok = all(torch.equal(a, b) for a, b in zip(outs, ref[s][4]))
                else:
                    f = cv.SteerableFiltersG4(None)
                    g4, h4 = f.setup_steer(x, -0.4)
                    ok = torch.equal(g4, ref[s][5]) and torch.equal(h4, ref[s][6]) and torch.equal(f.basis(7), ref[s][7])
                stream.synchronize()
                del f
                if not ok:
                    with lock:
                        errors.append((tid, it, s, what, search))
    except Exception as ex:
        with lock:
            errors.append((tid, "exception", type(ex).__name__, str(ex)))


t0 = time.perf_counter()
ths = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
for t in ths:
    t.start()
for t in ths:
    t.join()
torch.cuda.synchronize()
print("thread stress: %d threads x %d objects in %.1f s, %d mismatches / errors" % (nthreads, iters, time.perf_counter() - t0, len(errors)))
for e in errors[:20]:
    print("  ", e)
sys.exit(1 if errors else 0)
