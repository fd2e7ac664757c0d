import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
n = 4096
imgs = [np.random.default_rng(i).random((n, n), dtype=np.float32) for i in range(4)]
g, h = np.empty_like(imgs[0]), np.empty_like(imgs[0])
def run(tag):
    res = {}
    for overlap in (0, 1):
        f = cv.SteerableFiltersG2(None); f.set_option(L.OPT_HOST_OVERLAP, overlap); f.set_option(L.OPT_PLACEMENT_SEARCH, 0)
        f.setup_steer(imgs[0], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
        best = 1e9
        for r in range(3):
            t0 = time.perf_counter()
            for im in imgs: f.setup_steer(im, 0.3, flags=cv.SETUP_BASIS, out=(g, h))
            best = min(best, (time.perf_counter() - t0) / len(imgs))
        res[overlap] = best * 1e3
    print("%-64s sequential %.3f ms  overlapped %.3f ms" % (tag, res[0], res[1]), flush=True)
run("fresh process (plain states)")
img = torch.rand((n, n), device="cuda")
f0 = cv.SteerableFiltersG2(None); f0.set_option(L.OPT_PLACEMENT_SEARCH, 0); f0.setup(img); torch.cuda.synchronize()
run("while a handle with a PLAIN 4096^2 state is alive")
f1 = cv.SteerableFiltersG2(None); f1.set_option(L.OPT_PLACEMENT_SEARCH, 1); f1.setup(img); torch.cuda.synchronize()
run("while a handle with a searched state is alive")
time.sleep(2.0)
run("... two seconds later")
f1.set_option(L.OPT_PLACEMENT_SEARCH, 0); f1.setup(img); torch.cuda.synchronize()
run("after that handle switched to a plain block (window unmapped + released)")
f2 = cv.SteerableFiltersG2(None); f2.set_option(L.OPT_PLACEMENT_SEARCH, 2); f2.setup(img); torch.cuda.synchronize()
run("while a handle with a forced window (mode 2) is alive")
del f2
run("after it was destroyed (parked in the cache)")
