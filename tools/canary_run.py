#!/usr/bin/env python3
"""tools/canary_run.py <twin.so> [--quick] -- run the strip kernels of a CANARY twin of the library (make -C cvsteer_amd/csrc canary) over
every kind of launch bench.py times, at full size, plus a random mix of shapes / options, and print the twin's counters as one JSON line:
  stale      words read from a ring line that still held the pattern written before the load that refills the line was issued -- behind the
             hand-counted s_waitcnt vmcnt(N) that is supposed to cover the row.  MUST be 0 for libcvsteer_hip_canary.so and > 0 for
             libcvsteer_hip_canary_slack.so (counts 6 too high).
  short_rows output rows whose vector-memory stores were fewer than S_ROW, the compile-time lower bound the counts are built from.  MUST be 0.
  reads / rows  row reads checked / output rows tallied (proof that the checks ran).
Started by tests/test_gpu_canary.py as a child process (the twin is chosen through CVSTEER_HIP_LIB before the package is imported)."""
import ctypes as C, json, os, sys
twin = os.path.abspath(sys.argv[1])
quick = "--quick" in sys.argv
os.environ["CVSTEER_HIP_LIB"] = twin
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

lib = C.CDLL(twin)
lib.cvs_diag_canary.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]


def counters(reset=True):
    out = (C.c_ulonglong * 4)()
    assert lib.cvs_diag_canary(out, 1 if reset else 0) == 0
    return [int(v) for v in out]


counters()
report = {}


def section(name):
    torch.cuda.synchronize()
    c = counters()
    report[name] = c
    return c


gen = torch.Generator(device="cuda").manual_seed(11)
reps = 2 if quick else 6
n = 4096
imgs = [torch.rand((n, n), device="cuda", generator=gen) for _ in range(3)]
g, h = cv.alloc_planes(2, n, n, device="cuda")
outs8 = cv.alloc_planes(8, n, n, device="cuda")
for order in (L.ORDER_PLAIN, L.ORDER_DYNAMIC_TAIL, L.ORDER_XCD_COLUMNS):
    f = cv.SteerableFiltersG2(None, 4, 0.67)
    f.set_option(L.OPT_AUTOTUNE, 0)
    f.set_option(L.OPT_BLOCK_ORDER, order)
    for r in range(reps):
        im = imgs[r % 3]                       # new images: the read-ahead (dma_warm) priming counts
        f.setup(im, flags=cv.SETUP_BASIS)
        f.setup_steer(im, 0.3, flags=cv.SETUP_BASIS, out=(g, h))
        f.setup(im, flags=cv.SETUP_FULL)
        f.setup_steer(im, 0.3, flags=cv.SETUP_FULL, out=(g, h))
        f.pipeline(im, out=outs8)
        f.setup(imgs[0], flags=cv.SETUP_BASIS)   # ... and the same image again: resident
        f.setup(imgs[0], flags=cv.SETUP_FULL)
    del f
section("g2_4096_all_entry_points_three_orders")
f = cv.SteerableFiltersG2(None, 4, 0.67)
for lay in (0, 1, 3):
    f.set_option(L.OPT_STATE_LAYOUT, lay)
    f.setup(imgs[1], flags=cv.SETUP_FULL)
    f.pipeline(imgs[2], out=outs8)
f.set_option(L.OPT_STATE_LAYOUT, 1)
f.set_persist(False)
for mask in ((5, 6, 7), (0, 1), (2, 3, 4), (0, 1, 2, 3, 4, 5, 6, 7), (4,)):
    o = [outs8[k] if k in mask else None for k in range(8)]
    f.pipeline(imgs[0], out=o)
section("g2_4096_layouts_and_outputs_only")
u8 = (imgs[0] * 255).to(torch.uint8)
f = cv.SteerableFiltersG2(None, 4, 0.67)
for r in range(reps):
    f.setup_steer(u8, 0.3, flags=cv.SETUP_BASIS, out=(g, h))
    f.setup(u8, flags=cv.SETUP_FULL)
section("g2_4096_u8")
f4 = cv.SteerableFiltersG4(None, 6, 0.5)
for r in range(reps):
    f4.setup(imgs[r % 3])
    f4.setup_steer(imgs[r % 3], 0.3, out=(g, h))
f4.set_option(L.OPT_BLOCK_ORDER, L.ORDER_DYNAMIC_TAIL)
f4.setup(imgs[0])
f4.setup(u8)
section("g4_4096")
del imgs, outs8, g, h, f, f4, u8
big = [torch.rand((8192, 8192), device="cuda", generator=gen) for _ in range(2)]
gb, hb = torch.empty_like(big[0]), torch.empty_like(big[0])
fb = cv.SteerableFiltersG2(None, 4, 0.67)
for r in range(2 if quick else 4):
    fb.setup(big[r & 1], flags=cv.SETUP_BASIS)
    fb.setup_steer(big[r & 1], 0.3, flags=cv.SETUP_BASIS, out=(gb, hb))
section("g2_8192")
del gb, hb, fb
fp = cv.SteerableFiltersG2(None, 4, 0.67)
lv = fp.pyramid(big[0], 5)
hp = [cv.SteerableFiltersG2(None, 4, 0.67) for _ in lv]
for r in range(2 if quick else 4):
    cv.pyramid_setup(hp, big[r & 1], level_images=lv[1:], flags=cv.SETUP_BASIS)
    cv.pyramid_setup(hp, big[r & 1], level_images=lv[1:], flags=cv.SETUP_FULL)
section("pyramid_8192_5_levels")
del big, lv, hp, fp
frames = [torch.rand((32, 1080, 1920), device="cuda", generator=gen) for _ in range(2)]
fo8 = torch.empty((32, 8, 1080, 1920), device="cuda")
fo3 = torch.empty((32, 3, 1080, 1920), device="cuda")
ff = cv.SteerableFiltersG2(None, 4, 0.67)
for order in (L.ORDER_PLAIN, L.ORDER_DYNAMIC_TAIL):
    ff.set_option(L.OPT_AUTOTUNE, 0)
    ff.set_option(L.OPT_BLOCK_ORDER, order)
    ff.set_persist(True)
    for r in range(reps):
        ff.pipeline_batch(frames[r & 1], out=fo8)
    ff.set_persist(False)
    for r in range(reps):
        ff.pipeline_batch(frames[r & 1], out=fo3, outputs=(5, 6, 7))
        ff.pipeline_batch(frames[r & 1], out=fo8)
    ff.pipeline_batch((frames[0] * 255).to(torch.uint8), out=fo3, outputs=(5, 6, 7))
    lst = [frames[0][i] for i in range(6)]      # unrelated planes: the frame-table form
    ff.pipeline_batch(lst)
section("batch_32x1080p")
# the pipeline variants once more with the taps in scalar registers (the launches above ran the instances with the default taps compiled in)
os.environ["CVS_OPTS"] = "lit=0"
ff.set_option(L.OPT_BLOCK_ORDER, L.ORDER_PLAIN)
ff.set_persist(True)
ff.pipeline_batch(frames[0], out=fo8)
assert ff.launch_info()["literal_taps"] == 0
ff.set_persist(False)
ff.pipeline_batch(frames[1], out=fo3, outputs=(5, 6, 7))
ff.pipeline_batch(frames[1], out=fo8)
one = frames[0][0].contiguous()
fsingle = cv.SteerableFiltersG2(None, 4, 0.67)
fsingle.pipeline(one)
fsingle.set_persist(False)
fsingle.pipeline(one)
os.environ.pop("CVS_OPTS", None)
fsingle.set_persist(True)
fsingle.pipeline(one)
assert fsingle.launch_info()["literal_taps"] == 1
del fsingle, one
section("pipeline_variants_taps_from_arguments")
del frames, fo8, fo3, ff
# random shapes / options / entry points (ragged widths, few rows, strip heights, row ranges)
rng = np.random.default_rng(5)
for it in range(30 if quick else 120):
    rows, cols = int(rng.integers(13, 700)), int(rng.integers(5, 900))
    im = torch.rand((rows, cols), device="cuda", generator=gen)
    if rng.random() < 0.25:
        im = (im * 255).to(torch.uint8)
    kind4 = rng.random() < 0.3
    f = (cv.SteerableFiltersG4 if kind4 else cv.SteerableFiltersG2)(None)
    f.set_option(L.OPT_AUTOTUNE, 0)
    if rng.random() < 0.5:
        f.set_option(L.OPT_STRIP_ROWS, int(rng.integers(1, 60)))
    f.set_option(L.OPT_BLOCK_ORDER, int(rng.choice([L.ORDER_PLAIN, L.ORDER_DYNAMIC_TAIL, L.ORDER_XCD_COLUMNS])))
    os.environ["CVS_OPTS"] = "nt_stores=%d" % int(rng.integers(0, 2))
    if kind4:
        f.setup(im)
        f.setup_steer(im, 0.7)
    else:
        f.setup(im, flags=cv.SETUP_FULL)
        f.setup_steer(im, -0.4, flags=cv.SETUP_BASIS)
        if im.dtype != torch.uint8:
            f.pipeline(im)
            if rows >= 40:
                f.setup_pyr(im, flags=cv.SETUP_BASIS)
    del f
os.environ.pop("CVS_OPTS", None)
section("random_small_shapes")
tot = [sum(v[k] for v in report.values()) for k in range(4)]
print(json.dumps({"twin": os.path.basename(twin), "stale": tot[0], "short_rows": tot[1], "reads": tot[2], "rows": tot[3], "sections": report}))
