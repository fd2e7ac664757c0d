#!/usr/bin/env python3
"""tools/two_streams.py -- throughput of the fused filter+steer launch when consecutive images go to two handles on
two HIP streams (the tail of one launch overlaps the start-up of the next) against one handle on one stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv

n = 4096
imgs = [torch.rand((n, n), device="cuda") for _ in range(2)]
outs = [(torch.empty_like(imgs[0]), torch.empty_like(imgs[0])) for _ in range(2)]
hs = [cv.SteerableFiltersG2(None) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]

def one_stream(k):
    for i in range(k):
        hs[0].setup_steer(imgs[i & 1], 0.3, out=outs[0])

def two_streams(k):
    for i in range(k):
        with torch.cuda.stream(streams[i & 1]):
            hs[i & 1].setup_steer(imgs[i & 1], 0.3, out=outs[i & 1])

def timed(fn, k=200):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    main = torch.cuda.current_stream()
    e0.record(main)
    for s in streams: s.wait_event(e0)
    fn(k)
    for s in streams: main.wait_stream(s)
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k

for _ in range(3):
    one_stream(20); two_streams(20)        # tuning / placement search of both handles
for rep in range(3):
    a, b = timed(one_stream), timed(two_streams)
    print("one stream %.4f ms/image (%.0f Mpix/s, %.1f%%) | two streams %.4f ms/image (%.0f Mpix/s, %.1f%%)" %
          (a, n * n / a / 1e3, 40 * n * n / a / 1e6 / 80, b, n * n / b / 1e3, 40 * n * n / b / 1e6 / 80))
