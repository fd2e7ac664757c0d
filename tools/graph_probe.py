#!/usr/bin/env python3
"""tools/graph_probe.py -- config 3 (pyramid of one 8192^2 image, five dependent launches) and the one-object / small-image cases as
plain stream launches against a replay of the same launches captured into a HIP graph (torch.cuda.CUDAGraph on a side stream;
the engine neither tunes nor allocates under capture).
Measured (profiles/r04_graph_probe.txt): the replay is 3 % SLOWER for the pyramid chain and no faster for single small launches --
the stream path is what the library and bench.py use."""
import os, sys, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import cvsteer_amd as cv

side = torch.cuda.Stream()


def timeit(fn, steps, reps=7):
    out = []
    for _ in range(reps):
        for _ in range(max(3, steps // 2)):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / steps)
    return statistics.median(out)


def compare(name, work, steps, settle=70):
    with torch.cuda.stream(side):
        for _ in range(settle):
            work()
        torch.cuda.synchronize()
        t_stream = timeit(work, steps)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            work()
        torch.cuda.synchronize()
        t_graph = timeit(graph.replay, steps)
        t_stream2 = timeit(work, steps)
    print("%-44s stream %.4f ms | graph replay %.4f ms (%+.1f %%) | stream again %.4f ms" % (name, t_stream, t_graph, 100 * (t_graph / t_stream - 1), t_stream2), flush=True)


def main():
    big = torch.rand((8192, 8192), device="cuda")
    with torch.cuda.stream(side):
        hp = [cv.SteerableFiltersG2(None) for _ in range(5)]
        lv = hp[0].pyramid(big, 5)
    compare("C3 pyramid 8192^2, 5 levels (one image)", lambda: cv.pyramid_setup(hp, big, level_images=lv[1:], flags=cv.SETUP_BASIS), 10)
    for n in (512, 1024, 2048):
        img = torch.rand((n, n), device="cuda")
        with torch.cuda.stream(side):
            f = cv.SteerableFiltersG2(None)
            outs = cv.alloc_planes(8, n, n, device="cuda")
        compare("pipeline %d^2 (one launch per call)" % n, lambda: f.pipeline(img, out=outs), 50)


if __name__ == "__main__":
    main()
