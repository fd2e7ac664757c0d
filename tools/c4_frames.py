#!/usr/bin/env python3
"""tools/c4_frames.py -- the batched pipeline launch (1080x1920 frames, state kept, 84 B/pix) against the number of
frames per launch, and 32 frames as 2 / 4 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
def timeit(fn, steps=10):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps
frames = torch.rand((64, 1080, 1920), device="cuda")
out = torch.empty((64, 8, 1080, 1920), device="cuda")
for n in (4, 8, 16, 24, 32, 48, 64):
    f = cv.SteerableFiltersG2(None)
    ms = timeit(lambda: f.pipeline_batch(frames[:n], out=out[:n]))
    print("%2d frames in one launch: %.3f ms  %.4f ms/frame  %4.1f %%" % (n, ms, ms / n, 84 * n * 1080 * 1920 / ms / 1e6 / 80), flush=True)
    del f
for parts in (2, 4):
    hs = [cv.SteerableFiltersG2(None) for _ in range(parts)]
    k = 32 // parts
    def go():
        for i, h in enumerate(hs): h.pipeline_batch(frames[i * k:(i + 1) * k], out=out[i * k:(i + 1) * k])
    ms = timeit(go)
    print("32 frames as %d launches of %d (one handle each): %.3f ms  %4.1f %%" % (parts, k, ms, 84 * 32 * 1080 * 1920 / ms / 1e6 / 80), flush=True)
