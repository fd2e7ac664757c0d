#!/usr/bin/env python3
"""tools/c3_alone.py -- is level 0 of the config-3 pyramid slower inside the sequence of five levels than alone?
Same handles, same state blocks: level 0 alone, the five levels in sequence, level 0 alone again."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
big = torch.rand((8192, 8192), device="cuda")
f0 = cv.SteerableFiltersG2(None)
lv = f0.pyramid(big, 5)
print("level 0 is the input tensor itself:", lv[0].data_ptr() == big.data_ptr())
hs = [cv.SteerableFiltersG2(None) for _ in lv]
def t(fn, reps=20):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
def seq():
    for h, l in zip(hs, lv): h.setup(l, flags=cv.SETUP_BASIS)
pix0 = 8192 * 8192
for rnd in range(3):
    ms = t(lambda: hs[0].setup(lv[0], flags=cv.SETUP_BASIS))
    print("level 0 alone        %.4f ms (%.1f%%)" % (ms, 32 * pix0 / ms / 1e6 / 80))
    ms = t(seq)
    print("five levels in a row %.4f ms (%.1f%% over all levels)" % (ms, 32 * sum(l.numel() for l in lv) / ms / 1e6 / 80))
    for k in (1, 2, 3, 4):
        ms = t(lambda: [hs[i].setup(lv[i], flags=cv.SETUP_BASIS) for i in (0, k)])
        print("levels 0 + %d          %.4f ms" % (k, ms))
    ms = t(lambda: [hs[i].setup(lv[i], flags=cv.SETUP_BASIS) for i in (1, 2, 3, 4)])
    print("levels 1-4           %.4f ms" % ms)
