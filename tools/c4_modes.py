#!/usr/bin/env python3
"""tools/c4_modes.py -- the config-4 launch with state has a fast and a slow mode per process (59-74 %).  Which allocation
decides it?  One process: several handles (= several state blocks, all kept alive) against one set of frames / outputs, then
one handle against several output tensors and several frame sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
nfr = 32
pix = nfr * 1080 * 1920
def run(f, fs, out, steps=12):
    for i in range(3): f.pipeline_batch(fs[i & 1], out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(steps): f.pipeline_batch(fs[i & 1], out=out)
    b.record(); torch.cuda.synchronize()
    return 84 * pix / (a.elapsed_time(b) / steps) / 1e6 / 80
fs = [torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)]
out = torch.empty((nfr, 8, 1080, 1920), device="cuda")
hs = [cv.SteerableFiltersG2(None) for _ in range(6)]
print("six state blocks, one output tensor :", " ".join("%.1f" % run(h, fs, out) for h in hs), flush=True)
print("the same again                      :", " ".join("%.1f" % run(h, fs, out) for h in hs), flush=True)
outs = [out] + [torch.empty((nfr, 8, 1080, 1920), device="cuda") for _ in range(4)]
print("one state block, five output tensors:", " ".join("%.1f" % run(hs[0], fs, o) for o in outs), flush=True)
fss = [fs] + [[torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)] for _ in range(3)]
print("one state block, four frame-set pairs:", " ".join("%.1f" % run(hs[0], f2, out) for f2 in fss), flush=True)
