#!/usr/bin/env python3
"""tools/r06_probe.py -- round-6 A/B probes, one process = one allocation history (run it in several).  Every comparison is made
on ONE handle / state block with the online tuner off, settings alternating, SUSTAINED regions (a lead-in of a third of the region,
then `steps` launches between two events), three rounds, medians.  Sections (argv): c4 m4 c3 fresh c3lv c3x c3s c4w c4m strips strips2 strips3 strips4 long lit tune
  c4     32 x 1080p caller pipeline with state kept (BASELINE config 4): plain order vs dynamic tail, workgroups per CU, batch ways,
         strip height, planar [n][8][H][W] outputs vs row-interleaved [n][H][8][W] ones
  m4     full setup at 4096^2: two plane groups (layout 1) vs one merged group (layout 2), workgroups per CU
  c3     five-level pyramid of 8192^2: per-level budget and the emission variants (CVS_OPTS pyr_nt / warm_any / pyr_split)
  fresh  new-image regimes: exact priming wait (warm_exact) on M2 rotating, M1 8192^2 alternating, one object per image, G4 / u8 warm
Fractions are of the 8 TB/s HBM roofline on algorithmic bytes."""
import os, sys, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L

sections = sys.argv[1:] or ["c4", "m4", "c3", "fresh"]
PEAK = 8e12


def timeit(fn, steps):
    for _ in range(max(4, steps // 3)):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def ab(title, nbytes, steps, variants, rounds=3):
    """variants: [(label, setup_fn or None, call_fn)]; alternating, medians"""
    res = {lab: [] for lab, _, _ in variants}
    for _ in range(rounds):
        for lab, pre, fn in variants:
            if pre:
                pre()
            fn(); fn()
            res[lab].append(timeit(fn, steps))
    base = statistics.median(res[variants[0][0]])
    print(title)
    for lab, _, _ in variants:
        m = statistics.median(res[lab])
        print("   %-34s %.4f  (%+5.1f %%)   ms %.4f  [%s]" % (lab, nbytes / (m * 1e-3) / PEAK, 100 * (base / m - 1), m,
                                                             " ".join("%.4f" % (nbytes / (v * 1e-3) / PEAK) for v in res[lab])), flush=True)


def opts(s):
    def f():
        os.environ["CVS_OPTS"] = ("autotune=0," + s).rstrip(",")
    return f


os.environ["CVS_OPTS"] = "autotune=0"

if "c4" in sections:
    nfr = 32
    fsets = [torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)]
    fout = torch.empty((nfr, 8, 1080, 1920), device="cuda")
    fout_il = torch.empty((nfr, 1080, 8, 1920), device="cuda").permute(0, 2, 1, 3)   # [n][H][8][W] seen as [n][8][H][W]
    ff = cv.SteerableFiltersG2(None, 4, 0.67)
    alt = [0]
    fp = nfr * 1080 * 1920

    def c4(out=fout):
        alt[0] ^= 1
        ff.pipeline_batch(fsets[alt[0]], out=out)

    def order(o, s=""):
        def f():
            ff.set_option(L.OPT_BLOCK_ORDER, o)
            opts(s)()
        return f

    def strip(r):
        def f():
            ff.set_option(L.OPT_BLOCK_ORDER, 0)
            ff.set_option(L.OPT_STRIP_ROWS, r)
            opts("")()
        return f
    c4()
    torch.cuda.synchronize()
    ab("C4 32x1080p pipeline, state kept (84 B/pix), same state block", 84 * fp, 150, [
        ("plain order", order(0), c4),
        ("dynamic tail", order(L.ORDER_DYNAMIC_TAIL), c4),
        ("plain, wgcap=5", order(0, "wgcap=5"), c4),
        ("plain, wgcap=4", order(0, "wgcap=4"), c4),
        ("plain, wgcap=3", order(0, "wgcap=3"), c4),
        ("plain, batch_ways=1", order(0, "batch_ways=1"), c4),
        ("plain, batch_ways=4", order(0, "batch_ways=4"), c4),
        ("plain, interleaved outputs", order(0), lambda: c4(fout_il)),
        ("dynamic, interleaved outputs", order(L.ORDER_DYNAMIC_TAIL), lambda: c4(fout_il)),
        ("plain, interleaved, wgcap=4", order(0, "wgcap=4"), lambda: c4(fout_il)),
    ])
    ab("C4 strip heights (plain order)", 84 * fp, 100, [("10 rows", strip(10), c4), ("19 rows", strip(19), c4), ("28 rows", strip(28), c4)], rounds=2)
    ff.set_option(L.OPT_STRIP_ROWS, 0)
    ff.set_option(L.OPT_BLOCK_ORDER, -1)
    opts("")()
    del fsets, fout, fout_il, ff

if "m4" in sections:
    n = 4096
    img = torch.rand((n, n), device="cuda")
    outs8 = cv.alloc_planes(8, n, n, device="cuda")
    f = cv.SteerableFiltersG2(None, 4, 0.67)

    def lay(l, s=""):
        def fn():
            f.set_option(L.OPT_STATE_LAYOUT, l)
            opts(s)()
        return fn
    m4 = lambda: f.setup(img, flags=cv.SETUP_FULL)
    m5 = lambda: f.pipeline(img, out=outs8)
    m4()
    ab("M4 full setup 4096^2 (52 B/pix), same handle", 52 * n * n, 150, [
        ("two groups (layout 1), wg 3", lay(1), m4), ("merged (layout 2), wg 3", lay(2), m4),
        ("two groups, wg 4", lay(1, "wgcap=4"), m4), ("merged, wg 4", lay(2, "wgcap=4"), m4),
        ("two groups, wg 5", lay(1, "wgcap=5"), m4), ("merged, wg 5", lay(2, "wgcap=5"), m4),
        ("merged, uncapped", lay(2, "wgcap=0"), m4)])
    ab("M5 caller pipeline 4096^2 (84 B/pix), same handle", 84 * n * n, 100, [
        ("two groups (layout 1), wg 3", lay(1), m5), ("merged (layout 2), wg 3", lay(2), m5),
        ("two groups, wg 5", lay(1, "wgcap=5"), m5), ("merged, wg 5", lay(2, "wgcap=5"), m5),
        ("merged, wg 4", lay(2, "wgcap=4"), m5)])
    opts("")()
    del img, outs8, f

if "c3" in sections:
    bigs = [torch.rand((8192, 8192), device="cuda") for _ in range(2)]
    fp3 = cv.SteerableFiltersG2(None, 4, 0.67)
    lv = fp3.pyramid(bigs[0], 5)
    ppix = sum(l.shape[0] * l.shape[1] for l in lv)
    hp = [cv.SteerableFiltersG2(None, 4, 0.67) for _ in lv]
    fl = [0]

    def pyr():
        fl[0] ^= 1
        cv.pyramid_setup(hp, bigs[fl[0]], level_images=lv[1:], flags=cv.SETUP_BASIS)

    def lvl0_emit():
        fl[0] ^= 1
        hp[0].setup_pyr(bigs[fl[0]], flags=cv.SETUP_BASIS, out=lv[1])

    def lvl0_plain():
        fl[0] ^= 1
        hp[0].setup(bigs[fl[0]], flags=cv.SETUP_BASIS)

    def rest():
        for k in range(1, 5):
            if k < 4:
                hp[k].setup_pyr(lv[k], flags=cv.SETUP_BASIS, out=lv[k + 1])
            else:
                hp[k].setup(lv[k], flags=cv.SETUP_BASIS)
    whole = 32 * ppix + 4 * (ppix - 8192 * 8192)
    pyr()
    torch.cuda.synchronize()
    variants = [("default", ""), ("pyr_nt=1", "pyr_nt=1"), ("warm_any=1", "warm_any=1,warm_exact=1"), ("pyr_nt=1 + warm", "pyr_nt=1,warm_any=1,warm_exact=1"),
                ("pyr_split=1", "pyr_split=1,warm_exact=1"), ("pyr_split=1, wgcap=4", "pyr_split=1,warm_exact=1,wgcap=4"), ("wgcap=4", "wgcap=4"), ("wgcap=0", "wgcap=0")]
    ab("C3 whole: five levels of 8192^2 in one call (two images alternating)", whole, 24, [(lab, opts(s), pyr) for lab, s in variants])
    b0 = 36 * 8192 * 8192
    ab("C3 level 0 alone: 8192^2 new image, basis pass + emission of level 1 (36 B/pix)", b0, 24, [(lab, opts(s), lvl0_emit) for lab, s in variants])
    ab("   level 0 without the emission (32 B/pix)", 32 * 8192 * 8192, 24, [("default", opts(""), lvl0_plain), ("warm_exact=1", opts("warm_exact=1"), lvl0_plain)])
    ab("   levels 1..4 (4096^2 .. 512^2; input just written by the level above)", 32 * (ppix - 8192 * 8192) + 4 * (ppix - 8192 * 8192 - 4096 * 4096), 24,
       [("default", opts(""), rest), ("wgcap=4", opts("wgcap=4"), rest), ("wgcap=0", opts("wgcap=0"), rest)])
    opts("")()
    del bigs, lv, hp, fp3

if "fresh" in sections:
    n = 4096
    imgs = [torch.rand((n, n), device="cuda") for _ in range(8)]
    g, h = cv.alloc_planes(2, n, n, device="cuda")
    f = cv.SteerableFiltersG2(None, 4, 0.67)
    k = [0]

    def nxt():
        k[0] = (k[0] + 1) & 7
        return imgs[k[0]]
    m2 = lambda: f.setup_steer(nxt(), 0.3, flags=cv.SETUP_BASIS, out=(g, h))
    m1 = lambda: f.setup(nxt(), flags=cv.SETUP_BASIS)
    m4 = lambda: f.setup(nxt(), flags=cv.SETUP_FULL)

    def obj():
        fo = cv.SteerableFiltersG2(None, 4, 0.67)
        fo.setup_steer(nxt(), 0.3, flags=cv.SETUP_BASIS, out=(g, h))
        del fo
    wv = [("warm 4, priming wait as before", opts("")), ("warm 4, exact wait", opts("warm_exact=1")), ("warm 6, exact", opts("warm=6,warm_exact=1")),
          ("warm 3, exact", opts("warm=3,warm_exact=1")), ("warm off", opts("warm=0"))]
    m2()
    ab("M2 on 8 rotating 4096^2 images (40 B/pix)", 40 * n * n, 120, [(lab, pre, m2) for lab, pre in wv])
    ab("M1 on 8 rotating images (32 B/pix)", 32 * n * n, 120, [(lab, pre, m1) for lab, pre in wv[:2]])
    ab("M4 on 8 rotating images (52 B/pix)", 52 * n * n, 120, [(lab, pre, m4) for lab, pre in wv[:2]])
    ab("one object per image (40 B/pix)", 40 * n * n, 64, [(lab, pre, obj) for lab, pre in wv[:2]])
    f4 = cv.SteerableFiltersG4(None, 6, 0.5)
    g4 = lambda: f4.setup(nxt())
    ab("G4 basis on 8 rotating images (48 B/pix)", 48 * n * n, 80, [("no warm", opts(""), g4), ("warm 4 exact", opts("warm_any=2,warm_exact=1"), g4),
                                                                   ("warm 2 exact", opts("warm_any=2,warm_exact=1,warm=2"), g4)])
    u8 = [(im * 255).to(torch.uint8) for im in imgs]
    ku = [0]

    def m2u8():
        ku[0] = (ku[0] + 1) & 7
        f.setup_steer(u8[ku[0]], 0.3, flags=cv.SETUP_BASIS, out=(g, h))
    ab("M2 on 8 rotating 8-bit images (37 B/pix)", 37 * n * n, 120, [("no warm", opts(""), m2u8), ("warm 4 exact", opts("warm_any=4,warm_exact=1"), m2u8)])
    del imgs, u8, f4
    big = [torch.rand((8192, 8192), device="cuda") for _ in range(2)]
    fb = cv.SteerableFiltersG2(None, 4, 0.67)
    fl = [0]

    def b1():
        fl[0] ^= 1
        fb.setup(big[fl[0]], flags=cv.SETUP_BASIS)
    ab("M1 on two alternating 8192^2 images (32 B/pix)", 32 * 8192 * 8192, 30, [(lab, pre, b1) for lab, pre in wv])
    opts("")()


if "c3lv" in sections:
    # where the time of the five-level pyramid call goes: events between the levels of the chain (what cvs_pyramid_setup queues), two 8192^2
    # images alternating, under the emission variants
    bigs = [torch.rand((8192, 8192), device="cuda") for _ in range(2)]
    fp3 = cv.SteerableFiltersG2(None, 4, 0.67)
    lv = fp3.pyramid(bigs[0], 5)
    hp = [cv.SteerableFiltersG2(None, 4, 0.67) for _ in lv]
    fl = [0]

    def chain(ev=None):
        fl[0] ^= 1
        src = bigs[fl[0]]
        for k in range(5):
            if ev:
                ev[k].record()
            if k < 4:
                hp[k].setup_pyr(src if k == 0 else lv[k], flags=cv.SETUP_BASIS, out=lv[k + 1])
            else:
                hp[k].setup(lv[k], flags=cv.SETUP_BASIS)
        if ev:
            ev[5].record()
    variants = [("default", ""), ("pyr_nt=1 + warm (old wait)", "pyr_nt=1,warm_any=1"), ("pyr_nt=1 + warm exact", "pyr_nt=1,warm_any=1,warm_exact=1"),
                ("pyr_nt + warm, all levels new", "pyr_nt=1,warm_any=17"), ("pyr_nt + warm, wgcap=4", "pyr_nt=1,warm_any=1,wgcap=4"),
                ("pyr_nt + warm, all new, wgcap=4", "pyr_nt=1,warm_any=17,wgcap=4"), ("warm only, level 0 (plain level stores)", "warm_any=1")]
    res = {lab: [] for lab, _ in variants}
    for rnd in range(3):
        for lab, o in variants:
            opts(o)()
            for _ in range(6):
                chain()
            rows = []
            for _ in range(16):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
                chain(ev)
                torch.cuda.synchronize()
                rows.append([ev[k].elapsed_time(ev[k + 1]) for k in range(5)] + [ev[0].elapsed_time(ev[5])])
            res[lab].append([statistics.median(r[k] for r in rows) for k in range(6)])
    print("C3 chain, ms per level (8192, 4096, 2048, 1024, 512) and whole; median of 16 chains, 3 rounds")
    for lab, _ in variants:
        m = [statistics.median(r[k] for r in res[lab]) for k in range(6)]
        print("   %-40s %s  | whole %.4f" % (lab, " ".join("%.4f" % v for v in m[:5]), m[5]), flush=True)
    opts("")()
    del bigs, lv, hp, fp3

if "c4w" in sections:
    nfr = 32
    fsets = [torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)]
    fout = torch.empty((nfr, 8, 1080, 1920), device="cuda")
    fo3 = torch.empty((nfr, 3, 1080, 1920), device="cuda")
    ff = cv.SteerableFiltersG2(None, 4, 0.67)
    alt = [0]
    fp = nfr * 1080 * 1920

    def c4():
        alt[0] ^= 1
        ff.pipeline_batch(fsets[alt[0]], out=fout)

    def c4same():
        ff.pipeline_batch(fsets[0], out=fout)
    c4()
    ab("C4 32x1080p pipeline, state kept: frames requested ahead inside the launch", 84 * fp, 150, [
        ("no warm (default)", opts(""), c4), ("warm 4", opts("warm_any=8"), c4), ("warm 4 exact", opts("warm_any=8,warm_exact=1"), c4),
        ("warm 2 exact", opts("warm_any=8,warm_exact=1,warm=2"), c4), ("warm 8 exact", opts("warm_any=8,warm_exact=1,warm=8"), c4),
        ("same 32 frames every launch, no warm", opts(""), c4same)])
    ff.set_persist(False)

    def c4f():
        alt[0] ^= 1
        ff.pipeline_batch(fsets[alt[0]], out=fo3, outputs=(5, 6, 7))
    c4f()
    ab("C4 three maps only (16 B/pix)", 16 * fp, 150, [("no warm", opts(""), c4f), ("warm 4 exact", opts("warm_any=8,warm_exact=1"), c4f)])
    opts("")()
    del fsets, fout, fo3, ff

if "tune" in sections:
    # what the online tuner's decision is worth, sustained, on the SAME handle: every leg calls until the tuner has decided, then tuner-on and
    # tuner-off take turns (3 x 150 launches each)
    os.environ["CVS_OPTS"] = "verbose=1"
    n = 4096
    img = torch.rand((n, n), device="cuda")
    outs = cv.alloc_planes(8, n, n, device="cuda")
    frames = [torch.rand((32, 1080, 1920), device="cuda") for _ in range(2)]
    fo8 = torch.empty((32, 8, 1080, 1920), device="cuda")
    fo3 = torch.empty((32, 3, 1080, 1920), device="cuda")
    f = cv.SteerableFiltersG2(None)
    f4 = cv.SteerableFiltersG4(None)
    fb = cv.SteerableFiltersG2(None)
    fs = cv.SteerableFiltersG2(None)
    fs.set_persist(False)
    small = img[:1536, :2048].contiguous()
    alt = [0]

    def batch8():
        alt[0] ^= 1
        fb.pipeline_batch(frames[alt[0]], out=fo8)

    def batch3():
        alt[0] ^= 1
        fs.pipeline_batch(frames[alt[0]], out=fo3, outputs=(5, 6, 7))
    big = torch.rand((4000, 6000), device="cuda")
    uhd = torch.rand((2160, 3840), device="cuda")
    gb, hb = cv.alloc_planes(2, 4000, 6000, device="cuda")
    ouhd = cv.alloc_planes(8, 2160, 3840, device="cuda")
    fbig, fuhd = cv.SteerableFiltersG2(None), cv.SteerableFiltersG2(None)
    extra = (("M4 4000x6000", 52, 24000000, fbig, lambda: fbig.setup(big, flags=cv.SETUP_FULL)),
             ("M2 4000x6000", 40, 24000000, fbig, lambda: fbig.setup_steer(big, 0.3, flags=cv.SETUP_BASIS, out=(gb, hb))),
             ("M5 2160x3840", 84, 2160 * 3840, fuhd, lambda: fuhd.pipeline(uhd, out=ouhd)),
             ("M4 2160x3840", 52, 2160 * 3840, fuhd, lambda: fuhd.setup(uhd, flags=cv.SETUP_FULL)))
    legs = extra + (("M4", 52, n * n, f, lambda: f.setup(img, flags=cv.SETUP_FULL)), ("M5", 84, n * n, f, lambda: f.pipeline(img, out=outs)),
            ("G4", 48, n * n, f4, lambda: f4.setup(img)), ("M4 1536x2048", 52, 1536 * 2048, f, lambda: f.setup(small, flags=cv.SETUP_FULL)),
            ("C4 32x1080p state kept", 84, 32 * 1080 * 1920, fb, batch8), ("C4 32x1080p three maps", 16, 32 * 1080 * 1920, fs, batch3))
    for name, bpp, npx, hd, fn in legs:
        hd.set_option(L.OPT_AUTOTUNE, 1)
        calls = 0
        for _ in range(120):
            for _ in range(25):
                fn()
            calls += 25
            torch.cuda.synchronize()
            if hd.launch_info()["tune_state"] != 1:
                break
        li = hd.launch_info()
        res = {1: [], 0: []}
        for r in range(3):
            for mode in (1, 0):
                hd.set_option(L.OPT_AUTOTUNE, mode)
                fn(); fn()
                res[mode].append(timeit(fn, 150 if npx < (32 << 20) else 60))
        a, b = (bpp * npx / (statistics.median(res[m]) * 1e-3) / PEAK for m in (1, 0))
        print("%-24s tuned %.4f | default %.4f  (%+.1f %%)  decided after %4d calls (state %d): challenger kept %d -- order %d strip %d layout %d wg %d" %
              (name, a, b, 100 * (a / b - 1), calls, li["tune_state"], li["tuned"], li["block_order"], li["strip_rows"], li["state_layout"], li["wg_per_cu"]), flush=True)


if "c3x" in sections:
    # which neighbour costs level 0 its read-ahead gain?  Level 0 (pyr_nt + warm) timed by events inside chains that contain only some levels.
    bigs = [torch.rand((8192, 8192), device="cuda") for _ in range(2)]
    fp3 = cv.SteerableFiltersG2(None, 4, 0.67)
    lv = fp3.pyramid(bigs[0], 5)
    hp = [cv.SteerableFiltersG2(None, 4, 0.67) for _ in lv]
    fl = [0]

    def run(levels, ev=None):
        fl[0] ^= 1
        for k in levels:
            if ev is not None:
                ev.append(torch.cuda.Event(enable_timing=True)); ev[-1].record()
            src = bigs[fl[0]] if k == 0 else lv[k]
            if k < 4:
                hp[k].setup_pyr(src, flags=cv.SETUP_BASIS, out=lv[k + 1])
            else:
                hp[k].setup(src, flags=cv.SETUP_BASIS)
        if ev is not None:
            ev.append(torch.cuda.Event(enable_timing=True)); ev[-1].record()
    subsets = [(0,), (0, 4), (0, 3, 4), (0, 2), (0, 1), (0, 1, 2), (0, 1, 2, 3, 4)]
    for o in ("", "pyr_nt=1,warm_any=1", "pyr_nt=1,warm_any=17"):
        opts(o)()
        print("CVS_OPTS %-28s ms per level inside chains of the given levels (median of 16, 2 rounds)" % (o or "(default)"))
        for sub in subsets:
            meds = []
            for rnd in range(2):
                for _ in range(6):
                    run(sub)
                rows = []
                for _ in range(16):
                    ev = []
                    run(sub, ev)
                    torch.cuda.synchronize()
                    rows.append([ev[i].elapsed_time(ev[i + 1]) for i in range(len(sub))])
                meds.append([statistics.median(r[i] for r in rows) for i in range(len(sub))])
            m = [statistics.median(x[i] for x in meds) for i in range(len(sub))]
            print("   levels %-16s %s" % (sub, " ".join("%.4f" % v for v in m)), flush=True)
    opts("")()


if "c4m" in sections:
    nfr = 32
    fsets = [torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)]
    fout = torch.empty((nfr, 8, 1080, 1920), device="cuda")
    ff = cv.SteerableFiltersG2(None, 4, 0.67)
    alt = [0]
    fp = nfr * 1080 * 1920

    def c4():
        alt[0] ^= 1
        ff.pipeline_batch(fsets[alt[0]], out=fout)

    def pre(o, order=0):
        def f():
            ff.set_option(L.OPT_BLOCK_ORDER, order)
            opts(o)()
        return f
    c4()
    ab("C4 32x1080p pipeline, state kept: one 12-plane group per frame (batch_merge), frames requested ahead (warm_any=8)", 84 * fp, 150, [
        ("two groups per frame (default)", pre(""), c4), ("merged", pre("batch_merge=1"), c4), ("merged + warm 2", pre("batch_merge=1,warm_any=8,warm=2"), c4),
        ("merged + warm 2 + dynamic tail", pre("batch_merge=1,warm_any=8,warm=2", L.ORDER_DYNAMIC_TAIL), c4),
        ("merged, wgcap=4", pre("batch_merge=1,wgcap=4"), c4), ("merged, wgcap=3", pre("batch_merge=1,wgcap=3"), c4),
        ("warm 2 only", pre("warm_any=8,warm=2"), c4)])
    # bit-identical whatever the grouping
    opts("")(); ff.set_option(L.OPT_BLOCK_ORDER, 0)
    ff.pipeline_batch(fsets[0], out=fout); ref = fout.clone(); ff.select_frame(7); rb = [ff.basis(p).clone() for p in range(7)] + [ff.getDominantOrientationAngle().clone()]
    opts("batch_merge=1,warm_any=8,warm=2")()
    fout.zero_(); ff.pipeline_batch(fsets[0], out=fout); ff.select_frame(7)
    same = torch.equal(ref, fout) and all(torch.equal(a_, b_) for a_, b_ in zip(rb, [ff.basis(p) for p in range(7)] + [ff.getDominantOrientationAngle()]))
    print("   merged + warm: outputs and state planes of frame 7 bit-identical to the default: %s (state_layout %d)" % (same, ff.launch_info()["state_layout"]))
    opts("")()
    del fsets, fout, ff


if "strips" in sections:
    # the default strip height of mid-size images (states the Infinity Cache holds: 19 rows since round 2) against 10 rows, tuner off
    for shape in ((1024, 1024), (1080, 1920), (1536, 2048), (2048, 2048), (2160, 3840)):
        r, c = shape
        img = torch.rand(shape, device="cuda")
        g, h = cv.alloc_planes(2, r, c, device="cuda")
        o8 = cv.alloc_planes(8, r, c, device="cuda")
        f = cv.SteerableFiltersG2(None, 4, 0.67)

        def sr(n):
            def fn():
                f.set_option(L.OPT_STRIP_ROWS, n)
                opts("")()
            return fn
        for name, bpp, fn in (("M1", 32, lambda: f.setup(img, flags=cv.SETUP_BASIS)), ("M2", 40, lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))),
                              ("M4", 52, lambda: f.setup(img, flags=cv.SETUP_FULL)), ("M5", 84, lambda: f.pipeline(img, out=o8))):
            fn()
            ab("%s %dx%d" % (name, r, c), bpp * r * c, 200, [("default (0)", sr(0), fn), ("10 rows", sr(10), fn), ("19 rows", sr(19), fn), ("28 rows", sr(28), fn)], rounds=2)
        del f, img, g, h, o8


if "lit" in sections:
    # the pipeline variants with the reference's default taps compiled in as literal operands (k_basis_lit) against the same variants with the
    # taps in scalar registers (CVS_OPTS lit=0): same handle, alternating, sustained; and bit for bit the same outputs
    n = 4096
    img = torch.rand((n, n), device="cuda")
    outs8 = cv.alloc_planes(8, n, n, device="cuda")
    f = cv.SteerableFiltersG2(None, 4, 0.67)
    m5 = lambda: f.pipeline(img, out=outs8)
    m5()
    assert f.launch_info()["literal_taps"] == 1, f.launch_info()
    ab("M5 caller pipeline 4096^2, state kept (84 B/pix)", 84 * n * n, 150, [("taps in scalar registers (lit=0)", opts("lit=0"), m5), ("taps as literals", opts(""), m5)])
    ref = [o.clone() for o in outs8] + [f.basis(p).clone() for p in range(7)] + [f.getDominantOrientationAngle().clone()]
    opts("lit=0")(); m5(); assert f.launch_info()["literal_taps"] == 0
    same = all(torch.equal(a_, b_) for a_, b_ in zip(ref, list(outs8) + [f.basis(p) for p in range(7)] + [f.getDominantOrientationAngle()]))
    print("   outputs and state planes bit-identical: %s" % same)
    f.set_persist(False)
    o3 = [None] * 5 + list(outs8[5:])
    m5o = lambda: f.pipeline(img, out=o3)
    ab("M5 three maps only, single 4096^2 image (16 B/pix)", 16 * n * n, 150, [("lit=0", opts("lit=0"), m5o), ("literals", opts(""), m5o)])
    nfr = 32
    fsets = [torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)]
    fout = torch.empty((nfr, 8, 1080, 1920), device="cuda")
    fo3 = torch.empty((nfr, 3, 1080, 1920), device="cuda")
    ff = cv.SteerableFiltersG2(None, 4, 0.67)
    alt = [0]
    fp = nfr * 1080 * 1920

    def c4():
        alt[0] ^= 1
        ff.pipeline_batch(fsets[alt[0]], out=fout)

    def c4f():
        alt[0] ^= 1
        ff.pipeline_batch(fsets[alt[0]], out=fo3, outputs=(5, 6, 7))
    c4()
    ab("C4 32x1080p pipeline, state kept (84 B/pix)", 84 * fp, 150, [("lit=0", opts("lit=0"), c4), ("literals", opts(""), c4)])
    ff.set_persist(False)
    c4f()
    ab("C4 32x1080p three maps only (16 B/pix)", 16 * fp, 150, [("lit=0", opts("lit=0"), c4f), ("literals", opts(""), c4f)])
    opts("")(); ff.pipeline_batch(fsets[0], out=fo3, outputs=(5, 6, 7)); r3 = fo3.clone(); lit1 = ff.launch_info()["literal_taps"]
    opts("lit=0")(); fo3.zero_(); ff.pipeline_batch(fsets[0], out=fo3, outputs=(5, 6, 7))
    print("   three maps bit-identical: %s (literal_taps %d / %d)" % (torch.equal(r3, fo3), lit1, ff.launch_info()["literal_taps"]))
    opts("")()


if "strips2" in sections:
    # mid-size single images are short launches (15-40 us): how many workgroups there are against how many run at once decides the tail.  A fine
    # sweep of the strip height (tuner off, same handle), with the number of workgroups of each height beside it.
    import math
    heights = (5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 19)
    for shape in ((1080, 1920), (1536, 2048), (2048, 2048), (2160, 3840), (3000, 4000)):
        r, c = shape
        img = torch.rand(shape, device="cuda")
        g, h = cv.alloc_planes(2, r, c, device="cuda")
        o8 = cv.alloc_planes(8, r, c, device="cuda")
        f = cv.SteerableFiltersG2(None, 4, 0.67)
        gx = (math.ceil(c / 64) + 3) // 4
        print("%dx%d: workgroups per height: %s" % (r, c, " ".join("%d:%d" % (hh, gx * math.ceil(r / hh)) for hh in heights)))
        for name, bpp, fn in (("M1", 32, lambda: f.setup(img, flags=cv.SETUP_BASIS)), ("M2", 40, lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))),
                              ("M4", 52, lambda: f.setup(img, flags=cv.SETUP_FULL)), ("M5", 84, lambda: f.pipeline(img, out=o8))):
            res = {}
            for rnd in range(2):
                for hh in heights:
                    f.set_option(L.OPT_STRIP_ROWS, hh)
                    fn(); fn()
                    res.setdefault(hh, []).append(timeit(fn, 150))
            li = f.launch_info()
            best = min(heights, key=lambda hh: statistics.median(res[hh]))
            print("   %s wg_cap %d: %s   best %d" % (name, li["wg_per_cu"], " ".join("%d:%.3f" % (hh, bpp * r * c / (statistics.median(res[hh]) * 1e-3) / PEAK) for hh in heights), best), flush=True)
        del f, img, g, h, o8


if "strips3" in sections:
    # does the 7-row strip of the 3-12 Mpix sweep (strips2) also hold for the launches that write the orientation planes at 4096^2 and above, on new
    # images, and for the 32 x 1080p batch with state?
    heights = (6, 7, 8, 9, 10)

    def sweep(title, bpp, npx, f, fn, steps):
        res = {}
        for rnd in range(3):
            for hh in heights:
                f.set_option(L.OPT_STRIP_ROWS, hh)
                fn(); fn()
                res.setdefault(hh, []).append(timeit(fn, steps))
        print("   %-44s %s" % (title, " ".join("%d:%.4f" % (hh, bpp * npx / (statistics.median(res[hh]) * 1e-3) / PEAK) for hh in heights)), flush=True)
        f.set_option(L.OPT_STRIP_ROWS, 0)
    for shape in ((4096, 4096), (4000, 6000)):
        r, c = shape
        imgs = [torch.rand(shape, device="cuda") for _ in range(4)]
        o8 = cv.alloc_planes(8, r, c, device="cuda")
        g, h = cv.alloc_planes(2, r, c, device="cuda")
        f = cv.SteerableFiltersG2(None, 4, 0.67)
        k = [0]

        def nxt():
            k[0] = (k[0] + 1) & 3
            return imgs[k[0]]
        print("%dx%d" % shape)
        sweep("M4 resident", 52, r * c, f, lambda: f.setup(imgs[0], flags=cv.SETUP_FULL), 120)
        sweep("M5 resident", 84, r * c, f, lambda: f.pipeline(imgs[0], out=o8), 100)
        sweep("M4 on 4 rotating images", 52, r * c, f, lambda: f.setup(nxt(), flags=cv.SETUP_FULL), 120)
        sweep("M2 + orientation (setup_steer FULL) resident", 60, r * c, f, lambda: f.setup_steer(imgs[0], 0.3, flags=cv.SETUP_FULL, out=(g, h)), 120)
        sweep("M1 resident (for reference)", 32, r * c, f, lambda: f.setup(imgs[0], flags=cv.SETUP_BASIS), 120)
        sweep("M2 resident (for reference)", 40, r * c, f, lambda: f.setup_steer(imgs[0], 0.3, flags=cv.SETUP_BASIS, out=(g, h)), 120)
        del f, imgs, o8, g, h
    nfr = 32
    fsets = [torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)]
    fout = torch.empty((nfr, 8, 1080, 1920), device="cuda")
    fo3 = torch.empty((nfr, 3, 1080, 1920), device="cuda")
    ff = cv.SteerableFiltersG2(None, 4, 0.67)
    alt = [0]

    def c4():
        alt[0] ^= 1
        ff.pipeline_batch(fsets[alt[0]], out=fout)

    def c4f():
        alt[0] ^= 1
        ff.pipeline_batch(fsets[alt[0]], out=fo3, outputs=(5, 6, 7))
    print("32 x 1080p")
    sweep("C4 state kept", 84, nfr * 1080 * 1920, ff, c4, 100)
    ff.set_persist(False)
    heights = (7, 10, 13, 19, 28)
    sweep("C4 three maps only", 16, nfr * 1080 * 1920, ff, c4f, 100)


if "c3s" in sections:
    # config 3 once more with what the strip sweeps taught: the height of level 0 (8192^2 new image + emission, 73 % of the call) and of level 1, and their occupancy caps
    bigs = [torch.rand((8192, 8192), device="cuda") for _ in range(2)]
    fp3 = cv.SteerableFiltersG2(None, 4, 0.67)
    lv = fp3.pyramid(bigs[0], 5)
    ppix = sum(l.shape[0] * l.shape[1] for l in lv)
    hp = [cv.SteerableFiltersG2(None, 4, 0.67) for _ in lv]
    fl = [0]

    def pyr():
        fl[0] ^= 1
        cv.pyramid_setup(hp, bigs[fl[0]], level_images=lv[1:], flags=cv.SETUP_BASIS)
    whole = 32 * ppix + 4 * (ppix - 8192 * 8192)

    def heights(h0, h1, o=""):
        def f():
            hp[0].set_option(L.OPT_STRIP_ROWS, h0)
            hp[1].set_option(L.OPT_STRIP_ROWS, h1)
            opts(o)()
        return f
    pyr()
    ab("C3 whole, strip height of level 0 / level 1 (0 = default 10)", whole, 24, [("10 / 10", heights(0, 0), pyr), ("7 / 10", heights(7, 0), pyr), ("8 / 10", heights(8, 0), pyr),
        ("9 / 10", heights(9, 0), pyr), ("12 / 10", heights(12, 0), pyr), ("14 / 10", heights(14, 0), pyr), ("19 / 10", heights(19, 0), pyr), ("28 / 10", heights(28, 0), pyr),
        ("10 / 7", heights(0, 7), pyr), ("10 / 9", heights(0, 9), pyr), ("10 / 19", heights(0, 19), pyr), ("19 / 19", heights(19, 19), pyr)])
    heights(0, 0)()


if "strips4" in sections:
    # 7 against 10 rows for the launches that write the orientation planes, 3-14 Mpix, resident AND new images (four rotating), tuner off
    for shape in ((1536, 2048), (2048, 2048), (2000, 3000), (2160, 3840), (3000, 4000), (3456, 4608)):
        r, c = shape
        imgs = [torch.rand(shape, device="cuda") for _ in range(4)]
        o8 = cv.alloc_planes(8, r, c, device="cuda")
        g, h = cv.alloc_planes(2, r, c, device="cuda")
        f = cv.SteerableFiltersG2(None, 4, 0.67)
        k = [0]

        def nxt():
            k[0] = (k[0] + 1) & 3
            return imgs[k[0]]

        def sr(n):
            def fn():
                f.set_option(L.OPT_STRIP_ROWS, n)
                opts("")()
            return fn
        for name, bpp, fn in (("M4 resident", 52, lambda: f.setup(imgs[0], flags=cv.SETUP_FULL)), ("M4 new images", 52, lambda: f.setup(nxt(), flags=cv.SETUP_FULL)),
                              ("M5 resident", 84, lambda: f.pipeline(imgs[0], out=o8)), ("M5 new images", 84, lambda: f.pipeline(nxt(), out=o8)),
                              ("setup_steer FULL resident", 60, lambda: f.setup_steer(imgs[0], 0.3, flags=cv.SETUP_FULL, out=(g, h)))):
            fn()
            ab("%s %dx%d" % (name, r, c), bpp * r * c, 150, [("10 rows", sr(10), fn), ("7 rows", sr(7), fn), ("8 rows", sr(8), fn)], rounds=2)
        del f, imgs, o8, g, h


if "long" in sections:
    # what the tuner's remaining challengers are worth when they run for SECONDS (the power management answers a configuration over a longer time than a 12 ms turn):
    # regions of ~1.5 s each, alternating, same handle, tuner off; shader clock and power read from sysfs during each region
    import glob, threading, time
    pr = torch.cuda.get_device_properties(0)
    want = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    card = [c_ for c_ in glob.glob("/sys/class/drm/card*/device") if want in os.path.realpath(c_)]
    fclk = (sorted(glob.glob(os.path.join(card[0], "hwmon/hwmon*/freq1_input"))) or [None])[0] if card else None

    def region(fn, seconds=1.5):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        clk, stop = [], [False]

        def sampler():
            while not stop[0]:
                try:
                    clk.append(int(open(fclk).read().split()[0]))
                except Exception:
                    pass
                time.sleep(0.05)
        th = threading.Thread(target=sampler) if fclk else None
        if th:
            th.start()
        n, t0 = 0, time.perf_counter()
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        while time.perf_counter() - t0 < seconds:
            for _ in range(200):
                fn()
            n += 200
            torch.cuda.synchronize()
        b_.record()
        torch.cuda.synchronize()
        stop[0] = True
        if th:
            th.join()
        half = sorted(clk[len(clk) // 2:])
        return a_.elapsed_time(b_) / n, (half[len(half) // 2] / 1e6 if half else 0)
    n_ = 4096
    img = torch.rand((n_, n_), device="cuda")
    big = torch.rand((4000, 6000), device="cuda")
    o8 = cv.alloc_planes(8, n_, n_, device="cuda")
    g, h = cv.alloc_planes(2, n_, n_, device="cuda")
    f = cv.SteerableFiltersG2(None, 4, 0.67)
    fb = cv.SteerableFiltersG2(None, 4, 0.67)

    def cfg(hd, order=0, strip=0, o=""):
        def fn():
            hd.set_option(L.OPT_BLOCK_ORDER, order)
            hd.set_option(L.OPT_STRIP_ROWS, strip)
            opts(o)()
        return fn
    cases = [("M4 4096^2", 52 * n_ * n_, lambda: f.setup(img, flags=cv.SETUP_FULL), [("plain 10", cfg(f)), ("dynamic tail", cfg(f, L.ORDER_DYNAMIC_TAIL)), ("7 rows", cfg(f, 0, 7))]),
             ("M5 4096^2", 84 * n_ * n_, lambda: f.pipeline(img, out=o8), [("plain 10, wg 3", cfg(f)), ("wg 5", cfg(f, 0, 0, "wgcap=5")), ("dynamic tail", cfg(f, L.ORDER_DYNAMIC_TAIL)), ("7 rows", cfg(f, 0, 7))]),
             ("M2 4096^2 (headline)", 40 * n_ * n_, lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h)), [("plain 10", cfg(f)), ("9 rows", cfg(f, 0, 9)), ("8 rows", cfg(f, 0, 8))]),
             ("M4 4000x6000", 52 * 24000000, lambda: fb.setup(big, flags=cv.SETUP_FULL), [("plain 10", cfg(fb)), ("7 rows", cfg(fb, 0, 7)), ("dynamic tail", cfg(fb, L.ORDER_DYNAMIC_TAIL))])]
    imgs4 = [img] + [torch.rand((n_, n_), device="cuda") for _ in range(3)]
    rot = [0]

    def nxt4():
        rot[0] = (rot[0] + 1) & 3
        return imgs4[rot[0]]
    cases += [("M4 4096^2, four rotating images", 52 * n_ * n_, lambda: f.setup(nxt4(), flags=cv.SETUP_FULL), [("plain 10", cfg(f)), ("dynamic tail", cfg(f, L.ORDER_DYNAMIC_TAIL))]),
              ("M5 4096^2, four rotating images", 84 * n_ * n_, lambda: f.pipeline(nxt4(), out=o8), [("plain 10, wg 3", cfg(f)), ("dynamic tail", cfg(f, L.ORDER_DYNAMIC_TAIL))])]
    if "fresh_only" in sections:
        cases = cases[-2:]
    for title, nbytes, fn, variants in cases:
        fn()
        res = {lab: [] for lab, _ in variants}
        for rnd in range(2):
            for lab, pre in variants:
                pre()
                res[lab].append(region(fn))
        print(title + "  (1.5 s regions, two rounds: fraction of the HBM roofline @ shader clock)")
        for lab, _ in variants:
            print("   %-18s %s" % (lab, "   ".join("%.4f @ %.2f GHz" % (nbytes / (ms * 1e-3) / PEAK, ck / 1e3) for ms, ck in res[lab])), flush=True)
    cfg(f)(); cfg(fb)()
