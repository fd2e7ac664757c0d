#!/usr/bin/env python3
"""tools/r3_probe.py -- round-3 A/B probes, one process, interleaved rounds (every figure = median of the rounds).

    python tools/r3_probe.py c4strips      stateless / state-kept 32 x 1080p batch vs strip height
    python tools/r3_probe.py pitch         (run with CVS_STATE_PITCH_PAD=<n> in the environment; prints M1/M2/M4/rot legs)
    python tools/r3_probe.py c3order       config 3: fused chain vs "pyrDown first, then filter the (now cached) level"
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L


def timeit(fn, steps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def c4strips():
    nfr = 32
    sets = [torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)]
    pix = nfr * 1080 * 1920
    fo3 = torch.empty((nfr, 3, 1080, 1920), device="cuda")
    fo8 = torch.empty((nfr, 8, 1080, 1920), device="cuda")
    alt = {"i": 0}
    for persist, out, sel, bpp in ((False, fo3, (5, 6, 7), 16), (True, fo8, None, 84)):
        cands = [0, 10, 19, 28, 37, 46, 64, 91]
        hs = {}
        for sr in cands:
            f = cv.SteerableFiltersG2(None)
            f.set_persist(persist)
            f.set_option(L.OPT_AUTOTUNE, 0)
            if sr:
                f.set_strip_rows(sr)
            hs[sr] = f

        def run(f):
            alt["i"] ^= 1
            f.pipeline_batch(sets[alt["i"]], out=out, outputs=sel)

        res = {sr: [] for sr in cands}
        for rnd in range(5):
            for sr in cands:
                res[sr].append(timeit(lambda: run(hs[sr]), steps=6, warm=2))
        for sr in cands:
            ms = med(res[sr])
            print("persist=%d strip_rows=%3d : %.4f ms  %6.1f Gpix/s  %.3f of HBM (%d B/pix)  [info %s]" %
                  (persist, sr, ms, pix / ms / 1e6, bpp * pix / ms / 1e6 / 8000, bpp, hs[sr].launch_info()["strip_rows"]), flush=True)
        del hs


def pitch():
    n = 4096
    imgs = [torch.rand((n, n), device="cuda") for _ in range(8)]
    g, h = torch.empty_like(imgs[0]), torch.empty_like(imgs[0])
    outs8 = [torch.empty_like(imgs[0]) for _ in range(8)]
    f = cv.SteerableFiltersG2(None)
    f.set_option(L.OPT_AUTOTUNE, int(os.environ.get("PROBE_AUTOTUNE", "0")))
    npix = n * n
    rot = {"i": 0}

    def rotstep():
        rot["i"] = (rot["i"] + 1) & 7
        f.setup_steer(imgs[rot["i"]], 0.3, flags=cv.SETUP_BASIS, out=(g, h))

    legs = (("M2", lambda: f.setup_steer(imgs[0], 0.3, flags=cv.SETUP_BASIS, out=(g, h)), 40),
            ("M1", lambda: f.setup(imgs[0], flags=cv.SETUP_BASIS), 32),
            ("M4", lambda: f.setup(imgs[0], flags=cv.SETUP_FULL), 52),
            ("M5", lambda: f.pipeline(imgs[0], out=outs8), 84),
            ("M2rot", rotstep, 40))
    print("CVS_STATE_PITCH_PAD=%s  step of a state plane: %d B" % (os.environ.get("CVS_STATE_PITCH_PAD", "0"), 0), flush=True)
    for name, fn, bpp in legs:
        r = [timeit(fn, steps=20, warm=4) for _ in range(5)]
        ms = med(r)
        print("%-6s %.4f ms  %.3f of HBM   (min %.4f max %.4f)  pitch %d B" % (name, ms, bpp * npix / ms / 1e6 / 8000, min(r), max(r), f.basis_view(0)[3] if name != "M5x" else 0), flush=True)


def c3order():
    bigs = [torch.rand((8192, 8192), device="cuda") for _ in range(2)]
    f0 = cv.SteerableFiltersG2(None)
    lv = f0.pyramid(bigs[0], 5)
    ppix = sum(l.shape[0] * l.shape[1] for l in lv)
    hp = [cv.SteerableFiltersG2(None) for _ in lv]
    flip = {"i": 0}
    whole_bytes = 32 * ppix + 4 * (ppix - 8192 * 8192)

    def fused():
        flip["i"] ^= 1
        cur = bigs[flip["i"]]
        for k, hnd in enumerate(hp):
            if k + 1 < len(hp):
                hnd.setup_pyr(cur, flags=cv.SETUP_BASIS, out=lv[k + 1])
                cur = lv[k + 1]
            else:
                hnd.setup(cur, flags=cv.SETUP_BASIS)

    def down_first():
        """level 1 by the stand-alone strip-march pyrDown (reads level 0 from HBM, leaves it in the Infinity Cache), then the
        level-0 filter launch right behind it; the smaller levels as in the fused chain"""
        flip["i"] ^= 1
        cur = bigs[flip["i"]]
        pd, ps = cv.api._plane(lv[1]), cv.api._plane(cur)
        import ctypes as C
        f0._bind_stream(cur, lv[1])
        f0._check(cv.lib().cvs_pyr_down(f0._h, C.byref(ps), C.byref(pd)), "cvs_pyr_down")
        hp[0].setup(cur, flags=cv.SETUP_BASIS)
        cur = lv[1]
        for k in range(1, len(hp)):
            if k + 1 < len(hp):
                hp[k].setup_pyr(cur, flags=cv.SETUP_BASIS, out=lv[k + 1])
                cur = lv[k + 1]
            else:
                hp[k].setup(cur, flags=cv.SETUP_BASIS)

    def small_first():
        """all levels built first (four stand-alone pyrDown launches), then the five filter launches, smallest level LAST"""
        flip["i"] ^= 1
        cur = bigs[flip["i"]]
        import ctypes as C
        src = cur
        for k in range(1, 5):
            pd, ps = cv.api._plane(lv[k]), cv.api._plane(src)
            f0._bind_stream(src, lv[k])
            f0._check(cv.lib().cvs_pyr_down(f0._h, C.byref(ps), C.byref(pd)), "cvs_pyr_down")
            src = lv[k]
        hp[0].setup(cur, flags=cv.SETUP_BASIS)
        for k in range(1, 5):
            hp[k].setup(lv[k], flags=cv.SETUP_BASIS)

    res = {"fused": [], "down_first": [], "small_first": []}
    for rnd in range(5):
        for name, fn in (("fused", fused), ("down_first", down_first), ("small_first", small_first)):
            res[name].append(timeit(fn, steps=8, warm=3))
    for name, r in res.items():
        ms = med(r)
        print("%-11s %.4f ms  %.1f Gpix/s  %.3f of HBM (whole config, %d bytes)  min %.4f max %.4f" %
              (name, ms, ppix / ms / 1e6, whole_bytes / ms / 1e6 / 8000, whole_bytes, min(r), max(r)), flush=True)
    # per-level times inside the fused chain: events between the launches
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    acc = [0.0] * 5
    for it in range(10):
        flip["i"] ^= 1
        cur = bigs[flip["i"]]
        evs[0].record()
        for k, hnd in enumerate(hp):
            if k + 1 < len(hp):
                hnd.setup_pyr(cur, flags=cv.SETUP_BASIS, out=lv[k + 1])
                cur = lv[k + 1]
            else:
                hnd.setup(cur, flags=cv.SETUP_BASIS)
            evs[k + 1].record()
        torch.cuda.synchronize()
        for k in range(5):
            acc[k] += evs[k].elapsed_time(evs[k + 1]) / 10
    print("fused chain, per level (ms): " + "  ".join("L%d %.4f" % (k, acc[k]) for k in range(5)), flush=True)


def g4():
    """G4 bank layouts x strip heights on ONE image, interleaved: split 2 = two half banks in one launch (round 2 default),
    0 = one 11-plane kernel (2 waves/SIMD), 3 = one kernel with two window planes in LDS (3 waves/SIMD)"""
    n = 4096
    img = torch.rand((n, n), device="cuda")
    g, h = torch.empty_like(img), torch.empty_like(img)
    npix = n * n
    cands = [(2, 40), (2, 27), (2, 53), (2, 66), (0, 27), (0, 40)] + ([(3, 14), (3, 27), (3, 40), (3, 53), (3, 66)] if os.environ.get("PROBE_G4L") else [])   # split 3: tools/patches/g4_single_kernel_lds_window.patch
    hs = {}
    for sp, sr in cands:
        f = cv.SteerableFiltersG4(None)
        f.set_option(L.OPT_AUTOTUNE, 0)
        f.set_option(L.OPT_G4_SPLIT, sp)
        f.set_strip_rows(sr)
        hs[(sp, sr)] = f
    for name, bpp, fn in (("basis", 48, lambda f: f.setup(img)), ("basis+steer", 56, lambda f: f.setup_steer(img, 0.3, out=(g, h)))):
        res = {c: [] for c in cands}
        for rnd in range(5):
            for c in cands:
                res[c].append(timeit(lambda: fn(hs[c]), steps=10, warm=3))
        for c in cands:
            ms = med(res[c])
            print("G4 %-11s split %d strip %2d : %.4f ms  %.3f of HBM  (min %.4f)" % (name, c[0], c[1], ms, bpp * npix / ms / 1e6 / 8000, min(res[c])), flush=True)
    # bit-identity of the layouts
    ref = [hs[(2, 40)].basis(p).clone() for p in range(11)]
    if (3, 27) in hs:
        hs[(3, 27)].setup(img)
        print("split 3 == split 2:", all(torch.equal(hs[(3, 27)].basis(p), ref[p]) for p in range(11)), flush=True)


def host():
    """(needs tools/patches/host_register_option.patch applied: CVS_OPT_HOST_REGISTER was measured and not adopted)
    M2 with HOST planes (stream of 8 images, g/h back to the host) for CVS_OPT_HOST_REGISTER = 0 / 1 / 2, and the raw cost of
    hipHostRegister + hipHostUnregister of one 64 MiB plane"""
    import numpy as np
    n = 4096
    himgs = [np.random.default_rng(500 + i).random((n, n), dtype=np.float32) for i in range(8)]
    hg, hh = np.empty_like(himgs[0]), np.empty_like(himgs[0])
    rt = torch.cuda.cudart()
    probe = np.random.default_rng(1).random((n, n), dtype=np.float32)
    for rep in range(3):
        t0 = time.perf_counter()
        rc = rt.cudaHostRegister(probe.ctypes.data, probe.nbytes, 0)
        t1 = time.perf_counter()
        rt.cudaHostUnregister(probe.ctypes.data)
        t2 = time.perf_counter()
        print("hipHostRegister 64 MiB: %.3f ms (rc %s), unregister %.3f ms" % ((t1 - t0) * 1e3, rc, (t2 - t1) * 1e3), flush=True)
    fs = {}
    for mode in (0, 1, 2):
        f = cv.SteerableFiltersG2(None)
        f.set_option(14, mode)   # CVS_OPT_HOST_REGISTER of the patch
        f.setup_steer(himgs[0], 0.3, flags=cv.SETUP_BASIS, out=(hg, hh))
        fs[mode] = f
    torch.cuda.synchronize()
    time.sleep(2.5)
    best = {m: 1e9 for m in fs}
    for rnd in range(4):
        for m, f in fs.items():
            t0 = time.perf_counter()
            for im in himgs:
                f.setup_steer(im, 0.3, flags=cv.SETUP_BASIS, out=(hg, hh))
            best[m] = min(best[m], (time.perf_counter() - t0) / len(himgs))
    for m in fs:
        print("host planes, register mode %d: %.3f ms per image  %.2f Gpix/s  (link floor 2.4 ms)" % (m, best[m] * 1e3, n * n / best[m] / 1e9), flush=True)
    # the whole pipeline with 8 host outputs
    outs = [np.empty_like(himgs[0]) for _ in range(8)]
    for m, f in fs.items():
        b = 1e9
        for rnd in range(3):
            t0 = time.perf_counter()
            for im in himgs[:4]:
                f.pipeline(im, out=outs)
            b = min(b, (time.perf_counter() - t0) / 4)
        print("pipeline host in, 8 host planes out, mode %d: %.3f ms per image (floor 9.6 ms)" % (m, b * 1e3), flush=True)


def firstcall():
    """the reference's pattern -- one new object per image, a different image each time, wait after every call -- for a few
    launch configurations: which one suits an ISOLATED launch on a fresh image?"""
    n = 4096
    imgs = [torch.rand((n, n), device="cuda") for _ in range(8)]
    g, h = torch.empty_like(imgs[0]), torch.empty_like(imgs[0])
    npix = n * n
    confs = [("default", None, None), ("order 0, strip 10", 0, 10), ("order 0, strip 19", 0, 19), ("order 1 (4:3), strip 10", 1, 10),
             ("order 1 (4:3), strip 19", 1, 19), ("xcd columns, strip 10", 1000000, 10), ("xcd columns, strip 19", 1000000, 19), ("order 0, strip 28", 0, 28)]

    def one(image, order, strip):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f = cv.SteerableFiltersG2(None)
        if order is not None:
            f.set_option(L.OPT_BLOCK_ORDER, order)
            f.set_strip_rows(strip)
        e0.record()
        f.setup_steer(image, 0.3, flags=cv.SETUP_BASIS, out=(g, h))
        e1.record()
        torch.cuda.synchronize()
        del f
        return e0.elapsed_time(e1)

    res = {c[0]: [] for c in confs}
    i = 0
    for rnd in range(12):
        for name, order, strip in confs:
            i += 1
            res[name].append(one(imgs[i & 7], order, strip))
    for name, _, _ in confs:
        r = sorted(res[name][2:])
        ms = r[len(r) // 2]
        print("%-26s %.4f ms  %.3f of HBM  (min %.4f)" % (name, ms, 40 * npix / ms / 1e6 / 8000, r[0]), flush=True)


def sc1():
    """(probe build only: bst() issuing aux 18 when BasisArgs::nt_stores == 2, switched by CVS_PROBE_SC1; result in
    profiles/r03_nt_sc1_one_allocation_probe.txt: <= +1.5 points where the allocation is in its slow mode, -6 in its fast mode) streaming stores as `nt` vs `nt sc1`, alternating on ONE handle = one
    allocation, for several handles in a row: is sc1 the better policy where the allocation is in its slow mode?"""
    import ctypes as C
    n = 4096
    img = torch.rand((n, n), device="cuda")
    imgs = [img] + [torch.rand((n, n), device="cuda") for _ in range(3)]
    g, h = torch.empty_like(img), torch.empty_like(img)
    outs8 = [torch.empty_like(img) for _ in range(8)]
    npix = n * n
    for hnd in range(5):
        f = cv.SteerableFiltersG2(None)
        f.set_option(L.OPT_AUTOTUNE, 0)
        f.set_option(L.OPT_STORE_POLICY, 2)
        rot = {"i": 0}

        def rotstep():
            rot["i"] = (rot["i"] + 1) & 3
            f.setup_steer(imgs[rot["i"]], 0.3, flags=cv.SETUP_BASIS, out=(g, h))

        legs = (("M1", lambda: f.setup(img, flags=cv.SETUP_BASIS), 32), ("M2", lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h)), 40),
                ("M4", lambda: f.setup(img, flags=cv.SETUP_FULL), 52), ("M5", lambda: f.pipeline(img, out=outs8), 84), ("M2rot", rotstep, 40))
        line = "handle %d:" % hnd
        for name, fn, bpp in legs:
            r = {0: [], 1: []}
            for rnd in range(4):
                for pol in (0, 1):
                    torch.cuda.synchronize()
                    if pol:
                        os.environ["CVS_PROBE_SC1"] = "1"
                    else:
                        os.environ.pop("CVS_PROBE_SC1", None)
                    r[pol].append(timeit(fn, steps=20, warm=3))
            a, b = med(r[0]), med(r[1])
            line += "  %s nt %.3f | nt sc1 %.3f" % (name, bpp * npix / a / 1e6 / 8000, bpp * npix / b / 1e6 / 8000)
        print(line, flush=True)
        del f
        keep = torch.empty((64 << 20,), device="cuda")   # shift the next handle's allocation
        cv.lib().cvs_release_cached_memory()


def planepad():
    """(probe build: CVS_PROBE_PLANE_PAD_KB) M1 / M4 / M5 on five handles in a row (each its own allocation, a spacer kept in
    between) -- does a large pad between the state planes take the allocation's slow mode away?"""
    n = 4096
    img = torch.rand((n, n), device="cuda")
    outs8 = [torch.empty_like(img) for _ in range(8)]
    npix = n * n
    print("CVS_PROBE_PLANE_PAD_KB=%s" % os.environ.get("CVS_PROBE_PLANE_PAD_KB", "0"), flush=True)
    keep = []
    for hnd in range(6):
        f = cv.SteerableFiltersG2(None)
        f.set_option(L.OPT_AUTOTUNE, 0)
        legs = (("M1", lambda: f.setup(img, flags=cv.SETUP_BASIS), 32), ("M4", lambda: f.setup(img, flags=cv.SETUP_FULL), 52), ("M5", lambda: f.pipeline(img, out=outs8), 84))
        line = "handle %d:" % hnd
        for name, fn, bpp in legs:
            r = [timeit(fn, steps=20, warm=3) for _ in range(3)]
            line += "  %s %.3f" % (name, bpp * npix / med(r) / 1e6 / 8000)
        p0 = f.basis_view(0)[0]
        p1 = f.basis_view(1)[0]
        print(line + "   plane stride %.2f MiB" % ((p1 - p0) / 2 ** 20), flush=True)
        del f
        keep = [torch.empty((64 << 20,), device="cuda")]   # 256 MiB spacer: shifts the next handle's block
        cv.lib().cvs_release_cached_memory()


def c4modes():
    """config 4 with state kept on eight handles that are all alive (eight state blocks of 3.2 GB from one process), the same two
    frame sets and the same output tensor: how far apart are the allocations' modes, and how many draws find the fast one?"""
    nfr = 32
    sets = [torch.rand((nfr, 1080, 1920), device="cuda") for _ in range(2)]
    out = torch.empty((nfr, 8, 1080, 1920), device="cuda")
    pix = nfr * 1080 * 1920
    alt = {"i": 0}
    hs = []
    for k in range(8):
        f = cv.SteerableFiltersG2(None)
        f.set_option(L.OPT_AUTOTUNE, 0)
        hs.append(f)

    def run(f):
        alt["i"] ^= 1
        f.pipeline_batch(sets[alt["i"]], out=out)

    res = [[] for _ in hs]
    for rnd in range(4):
        for k, f in enumerate(hs):
            res[k].append(timeit(lambda: run(f), steps=6, warm=2))
    for k, f in enumerate(hs):
        ms = med(res[k])
        print("handle %d (state at %#x): %.4f ms  %.3f of HBM" % (k, f.basis_view(0)[0], ms, 84 * pix / ms / 1e6 / 8000), flush=True)
    print("output tensor at %#x" % out.data_ptr(), flush=True)


def pyrcost():
    """what does the F_PYR emission cost the 8192^2 filter launch?  setup() vs setup_pyr() on two alternating 8192^2 images,
    one handle each, interleaved rounds; and the same at 4096^2"""
    for n in (8192, 4096):
        bigs = [torch.rand((n, n), device="cuda") for _ in range(2 if n == 8192 else 8)]
        nxt = torch.empty((n // 2, n // 2), device="cuda")
        fa, fb = cv.SteerableFiltersG2(None), cv.SteerableFiltersG2(None)
        flip = {"i": 0}

        def plain():
            flip["i"] = (flip["i"] + 1) % len(bigs)
            fa.setup(bigs[flip["i"]], flags=cv.SETUP_BASIS)

        def fused():
            flip["i"] = (flip["i"] + 1) % len(bigs)
            fb.setup_pyr(bigs[flip["i"]], flags=cv.SETUP_BASIS, out=nxt)

        r = {"plain": [], "fused": []}
        for rnd in range(5):
            r["plain"].append(timeit(plain, steps=10, warm=3))
            r["fused"].append(timeit(fused, steps=10, warm=3))
        for k, bpp in (("plain", 32), ("fused", 33)):
            ms = med(r[k])
            print("%d^2 fresh images, %s: %.4f ms  %.3f of HBM (%d B/pix)  strip %d order %d" % (n, k, ms, bpp * n * n / ms / 1e6 / 8000, bpp,
                  (fa if k == "plain" else fb).launch_info()["strip_rows"], (fa if k == "plain" else fb).launch_info()["block_order"]), flush=True)


if __name__ == "__main__":
    {"c4strips": c4strips, "pitch": pitch, "c3order": c3order, "g4": g4, "host": host, "firstcall": firstcall, "sc1": sc1, "planepad": planepad, "c4modes": c4modes, "pyrcost": pyrcost}[sys.argv[1]]()
