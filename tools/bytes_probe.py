#!/usr/bin/env python3
"""tools/bytes_probe.py -- where does the 8-bit end-to-end flow (NativeBatch.run_to_u8: bytes up, maps kept on the GPU,
normalised there, bytes down) spend its time?  32 frames of 1080p."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cvsteer_amd as cv
from cvsteer_amd import batch
n, shape = 32, (1080, 1920)
u8 = np.random.default_rng(1).integers(0, 256, (n,) + shape, dtype=np.uint8)
nb = batch.NativeBatch.local((0,))
nb.set_persist(False)
out = np.zeros((n, 3) + shape, np.uint8)
time.sleep(2)
for rep in range(4):
    t0 = time.perf_counter()
    _, t = nb.run(u8, n, shape, outputs=(5, 6, 7), gather=False)
    t1 = time.perf_counter()
    nb.run_to_u8(u8, out=out)
    t2 = time.perf_counter()
    print("run (bytes up + launches) %.2f ms [upload %.2f]   run_to_u8 total %.2f ms  -> conversion + download %.2f ms" %
          ((t1 - t0) * 1e3, t["scatter"], (t2 - t1) * 1e3, (t2 - t1 - (t1 - t0)) * 1e3), flush=True)
