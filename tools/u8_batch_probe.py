import os, sys, statistics, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cvsteer_amd as cv
from cvsteer_amd import _lib as L
nfr = 32
fs = [torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)]
fu = [(f * 255).to(torch.uint8) for f in fs]
fo3 = torch.empty((nfr, 3, 1080, 1920), device="cuda")
ff = cv.SteerableFiltersG2(None)
ff.set_persist(False)
if os.environ.get("NOTUNE"): ff.set_option(L.OPT_AUTOTUNE, 0)
alt = {"i": 0}
def run(src):
    alt["i"] ^= 1
    ff.pipeline_batch(src[alt["i"]], out=fo3, outputs=(5, 6, 7))
for name, src in (("f32", fs), ("u8", fu), ("f32", fs), ("u8", fu)):
    for _ in range(70): run(src)
    torch.cuda.synchronize()
    ts = []
    for _ in range(60):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(src); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    # back-to-back
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(50): run(src)
    b.record(); torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 50 * 1e3
    print("%s single-call ms: min %.3f median %.3f max %.3f | 50 back to back: %.4f ms per call (events), host wall %.4f ms per call; launch %s" % (name, min(ts), statistics.median(ts), max(ts), a.elapsed_time(b) / 50, wall, {k: ff.launch_info()[k] for k in ("block_order", "strip_rows")}), flush=True)
