import os, sys, statistics
sys.path.insert(0, "/root/repo")
import torch
import cvsteer_amd as cv
nfr = 32
fsets = [torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)]
fout = torch.empty((nfr, 8, 1080, 1920), device="cuda")
ff = cv.SteerableFiltersG2(None, 4, 0.67)
alt = [0]
def step2():
    alt[0] ^= 1
    ff.pipeline_batch(fsets[alt[0]], out=fout)
def step1():
    ff.pipeline_batch(fsets[0], out=fout)
def timeit(fn, steps=10):
    for _ in range(30): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
for _ in range(100): step2()
for r in range(3):
    a, b = timeit(step1), timeit(step2)
    print("C4 32x1080p state kept: one frame set %.4f ms %.3f | two sets alternating %.4f ms %.3f" % (a, 84 * nfr * 1080 * 1920 / a / 8e9, b, 84 * nfr * 1080 * 1920 / b / 8e9), flush=True)
