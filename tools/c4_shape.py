#!/usr/bin/env python3
"""tools/c4_shape.py -- is the config-4 leg slow because of the frame shape or because of the batching?  The same 66 Mpix
as 32 frames of 1080x1920 in one batched launch, as ONE image of 34560x1920, and as one image of 8640x7680; pipeline
with state kept (84 B/pix)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
def timeit(fn, steps=10):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps
pix = 32 * 1080 * 1920
frames = torch.rand((32, 1080, 1920), device="cuda")
out = torch.empty((32, 8, 1080, 1920), device="cuda")
f = cv.SteerableFiltersG2(None)
ms = timeit(lambda: f.pipeline_batch(frames, out=out))
print("32 x 1080x1920, one batched launch : %.3f ms  %4.1f %%" % (ms, 84 * pix / ms / 1e6 / 80), flush=True)
for shape in ((34560, 1920), (8640, 7680), (17280, 3840)):
    img = frames.reshape(shape)
    outs = [out.reshape(-1)[k * pix:(k + 1) * pix].reshape(shape) for k in range(8)]
    for place in (0, 1):
        os.environ["CVS_PLACEMENT_SEARCH"] = str(place)
        f1 = cv.SteerableFiltersG2(None)
        ms = timeit(lambda: f1.pipeline(img, out=outs))
        print("one image %5d x %4d, placement %d    : %.3f ms  %4.1f %%" % (shape[0], shape[1], place, ms, 84 * pix / ms / 1e6 / 80), flush=True)
        del f1
