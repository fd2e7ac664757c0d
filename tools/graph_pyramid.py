#!/usr/bin/env python3
"""tools/graph_pyramid.py -- BASELINE config 3 (G2+H2 on a 5-level pyramid of one 8192x8192 image): the five launches
issued one by one from Python vs captured once in a HIP graph (torch.cuda.CUDAGraph on the engine's stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv

def timeit(fn, steps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps

big = torch.rand((8192, 8192), device="cuda")
builder = cv.SteerableFiltersG2(None)
lv = builder.pyramid(big, 5)
ppix = sum(l.shape[0] * l.shape[1] for l in lv)
hs = [cv.SteerableFiltersG2(None) for _ in lv]
def run():
    for f, l in zip(hs, lv):
        f.setup(l, flags=cv.SETUP_BASIS)
for _ in range(12): run()          # allocations, order tuning and placement search happen here, not under capture
torch.cuda.synchronize()
t_plain = timeit(run)
for f, l in zip(hs, lv):
    t = timeit(lambda: f.setup(l, flags=cv.SETUP_BASIS))
    print("   level %5d x %5d alone: %.4f ms  %.1f%%" % (l.shape[0], l.shape[1], t, 32 * l.shape[0] * l.shape[1] / t / 1e6 / 80))
ref = [f.basis(3).clone() for f in hs]
graph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=s):
        run()
torch.cuda.synchronize()
for f in hs: f.basis(3).zero_()
graph.replay(); torch.cuda.synchronize()
assert all(torch.equal(f.basis(3), r) for f, r in zip(hs, ref)), "graph replay must reproduce the planes"
t_graph = timeit(graph.replay)
side = [torch.cuda.Stream() for _ in lv]
def run_streams():
    main = torch.cuda.current_stream()
    ev = main.record_event()
    for f, l, st in zip(hs, lv, side):
        st.wait_event(ev)
        with torch.cuda.stream(st):
            f.setup(l, flags=cv.SETUP_BASIS)
    for st in side:
        main.wait_event(st.record_event())
for _ in range(3): run_streams()
torch.cuda.synchronize()
for f in hs: f.basis(3).zero_()
run_streams(); torch.cuda.synchronize()
assert all(torch.equal(f.basis(3), r) for f, r in zip(hs, ref))
t_streams = timeit(run_streams)
print("one stream per level: %.4f ms (%.0f Mpix/s, %.1f%%)" % (t_streams, ppix / t_streams / 1e3, 32 * ppix / t_streams / 1e6 / 80))
print("pyramid %d px: plain %.4f ms (%.0f Mpix/s, %.1f%% of 8 TB/s at 32 B/pix) | graph %.4f ms (%.0f Mpix/s, %.1f%%)" %
      (ppix, t_plain, ppix / t_plain / 1e3, 32 * ppix / t_plain / 1e6 / 80, t_graph, ppix / t_graph / 1e3, 32 * ppix / t_graph / 1e6 / 80))
