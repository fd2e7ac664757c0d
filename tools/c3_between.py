#!/usr/bin/env python3
"""tools/c3_between.py -- what does a launch BETWEEN two 8192^2 basis launches cost the big launch?  (diagnostic for config 3)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
big = torch.rand((8192, 8192), device="cuda")
big2 = torch.rand((8192, 8192), device="cuda")
h = cv.SteerableFiltersG2(None)
tiny = torch.empty(256, device="cuda")
m8 = torch.empty(2 * 1024 * 1024, device="cuda")
m64 = torch.empty(16 * 1024 * 1024, device="cuda")
m256 = torch.empty(64 * 1024 * 1024, device="cuda")
def t(fn, reps=20):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
base = lambda: h.setup(big, flags=cv.SETUP_BASIS)
cases = [("big alone", lambda: None), ("+ 1 KiB fill", lambda: tiny.zero_()), ("+ 8 MiB fill", lambda: m8.zero_()), ("+ 64 MiB fill", lambda: m64.zero_()),
         ("+ 256 MiB fill", lambda: m256.zero_()), ("+ 8 MiB read (sum)", lambda: m8.sum()), ("+ 64 MiB read (sum)", lambda: m64.sum()), ("+ 256 MiB read (sum)", lambda: m256.sum())]
for rnd in range(2):
    for name, extra in cases:
        alone = t(extra) if name != "big alone" else 0.0
        both = t(lambda: (base(), extra()))
        print("%-22s both %.4f ms, extra alone %.4f ms, big launch net %.4f ms (%.1f%%)" % (name, both, alone, both - alone, 32 * 8192 * 8192 / (both - alone) / 1e6 / 80), flush=True)
    i = [0]
    def rot():
        i[0] ^= 1
        h.setup(big2 if i[0] else big, flags=cv.SETUP_BASIS)
    ms = t(rot)
    print("two alternating 8192^2 inputs: %.4f ms (%.1f%%)" % (ms, 32 * 8192 * 8192 / ms / 1e6 / 80), flush=True)
