#!/usr/bin/env python3
"""tools/burst_probe.py -- does the time of a 20-step timed region depend on what the card did in the milliseconds before it?

bench.py times bursts (K steps of 0.1-0.3 ms between two host synchronisations); its sustained legs (`device_telemetry`) showed the
same launches running faster back to back for a second than inside such a burst on some boxes.  Variants of the lead-in to the
region, same handle, same launches:
  idle30   : synchronize, 30 ms of host sleep (what a generation-2 gc.collect() of the interpreter costs), region
  w5       : 5 warm-up steps, synchronize, region                      (the contract's W = 5)
  w50      : 50 warm-up steps, synchronize, region
  w200     : 200 warm-up steps, synchronize, region
  w5_ns    : 5 warm-up steps, NO synchronize before the first event (the events are on the stream)
Each variant: 9 regions, median / min / max of the event time per step."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import cvsteer_amd as cv

K = int(os.environ.get("K", "20"))


def region(fn, lead, sync=True):
    lead()
    if sync:
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K


def main():
    import gc
    gc.disable()
    img = torch.rand((4096, 4096), device="cuda")
    f = cv.SteerableFiltersG2(img, 4, 0.67)
    f4 = cv.SteerableFiltersG4(img, 6, 0.5)
    g, h = cv.alloc_planes(2, 4096, 4096, device="cuda")
    outs8 = cv.alloc_planes(8, 4096, 4096, device="cuda")
    legs = (("M2", lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h)), 40), ("M1", lambda: f.setup(img, flags=cv.SETUP_BASIS), 32),
            ("M5", lambda: f.pipeline(img, out=outs8), 84), ("G4", lambda: f4.setup(img), 48))
    for name, fn, bpp in legs:
        for _ in range(80):
            fn()
        torch.cuda.synchronize()

        def warm(n):
            def go():
                for _ in range(n):
                    fn()
            return go

        def idle30():
            torch.cuda.synchronize()
            time.sleep(0.03)
        variants = (("idle30", idle30, True), ("w5", warm(5), True), ("w50", warm(50), True), ("w200", warm(200), True), ("w5_ns", warm(5), False),
                    ("idle30", idle30, True), ("w50", warm(50), True))
        line = []
        for vn, lead, sync in variants:
            v = sorted(region(fn, lead, sync) for _ in range(9))
            fr = lambda ms: bpp * 4096 * 4096 / (ms * 1e-3) / 8e12
            line.append("%s %.3f (%.3f-%.3f)" % (vn, fr(v[4]), fr(v[-1]), fr(v[0])))
        print("%-3s K=%d  " % (name, K) + " | ".join(line), flush=True)


if __name__ == "__main__":
    main()
