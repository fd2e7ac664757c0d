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
    print("%-60s sequential %.3f ms  overlapped %.3f ms" % (tag, res[0], res[1]), flush=True)
run("fresh process")
big = torch.rand((8192, 8192), device="cuda")
fb = cv.SteerableFiltersG2(None); fb.set_option(L.OPT_PLACEMENT_SEARCH, 1); fb.setup(big, flags=cv.SETUP_BASIS); torch.cuda.synchronize()
run("while a handle with a searched 8192^2 state is alive")
del fb
run("after that handle was destroyed (block in the cache)")
cv.lib().cvs_release_cached_memory()
run("after the block cache was emptied")
frames = torch.rand((32, 1080, 1920), device="cuda"); fout = torch.empty((32, 8, 1080, 1920), device="cuda")
ff = cv.SteerableFiltersG2(None)
for _ in range(5): ff.pipeline_batch(frames, out=fout)
torch.cuda.synchronize(); del ff, frames, fout
run("after a 32-frame batch (3.2 GB plain state) came and went")
lv = cv.SteerableFiltersG2(None).pyramid(big, 5)
hp = [cv.SteerableFiltersG2(None) for _ in lv]
for hnd, l in zip(hp, lv): hnd.setup(l, flags=cv.SETUP_BASIS)
torch.cuda.synchronize()
run("with five pyramid-level handles alive")
del hp, lv
run("after they were destroyed")
