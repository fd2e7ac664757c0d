#!/usr/bin/env python3
"""tools/fuzz_campaign.py [iterations] [seed] -- a long randomised differential run of the HIP path against the oracle, beyond
what the test suite can afford every time: random kind, shape (1 x 1 ... 400 x 900, biased towards the fast / generic path
boundaries and multiples of 64), device or host planes, strided (ROI) inputs and outputs, launch options (strip height, block
order, persist; CVS_OPTS store policy, warm, workgroups per CU), entry points (setup flags, fused steer, pipeline, batch with random
frame counts and dispatch parts, row ranges), non-finite pixels.  Prints one line per failure with everything needed to replay
it, and a summary.  Nothing here is timed."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cvsteer_amd as cv
from cvsteer_amd import _lib as L
from oracle import pyoracle as ora

TOL = 1e-5
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)


def angle_diff(a, b, period):
    d = np.abs(a - b) % period
    return np.minimum(d, period - d)


def pick_shape():
    mode = rng.integers(0, 5)
    if rng.integers(0, 25) == 0:      # now and then an image large enough for streaming stores and many strips per wave slot
        return int(rng.integers(1200, 2600)), int(rng.integers(1500, 3300))
    if rng.integers(0, 12) == 0:      # column counts whose 256-column blocks divide evenly among the 8 XCDs (the XCD-column order with uneven shares)
        return int(rng.integers(13, 700)), int(2048 * rng.integers(1, 3) - rng.integers(0, 200))
    if mode == 0:
        return int(rng.integers(1, 24)), int(rng.integers(1, 24))
    if mode == 1:
        return int(rng.choice([12, 13, 14, 18, 19, 20, 27, 28, 29])), int(rng.choice([4, 5, 6, 7, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257]))
    if mode == 2:
        return int(rng.integers(13, 120)), int(64 * rng.integers(1, 8) + rng.integers(-2, 3))
    return int(rng.integers(1, 400)), int(rng.integers(1, 900))


def make_image(rows, cols):
    kind = rng.integers(0, 4)
    if kind == 0:
        img = rng.random((rows, cols), dtype=np.float32)
    elif kind == 1:
        yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
        img = (0.5 + 0.3 * np.sin(0.21 * xx + 0.05 * yy) + 0.2 * np.cos(0.13 * (yy - xx))).astype(np.float32) + 0.05 * rng.random((rows, cols), dtype=np.float32)
    elif kind == 2:
        img = (rng.random((rows, cols), dtype=np.float32) * 255.0).astype(np.float32)
    else:
        img = np.zeros((rows, cols), np.float32)
        for _ in range(int(rng.integers(1, 6))):
            img[rng.integers(0, rows), rng.integers(0, cols)] = float(rng.normal())
    return img


def as_plane(img, device, strided):
    """the same values as a host array or device tensor, optionally as a view into a wider buffer (row pitch > cols)"""
    rows, cols = img.shape
    if strided:
        pad_l, pad_r = int(rng.integers(0, 9)), int(rng.integers(1, 70))
        big = np.full((rows, cols + pad_l + pad_r), 7.5, np.float32)
        big[:, pad_l:pad_l + cols] = img
        if device:
            return torch.from_numpy(big).cuda()[:, pad_l:pad_l + cols]
        return big[:, pad_l:pad_l + cols]
    return torch.from_numpy(img).cuda() if device else img


def to_np(x):
    return x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


fails = 0
counts = {}
for it in range(iters):
    rows, cols = pick_shape()
    kind = int(rng.choice([2, 2, 4]))
    device = bool(rng.integers(0, 2))
    strided = bool(rng.integers(0, 2))
    img = make_image(rows, cols)
    scale = max(1.0, float(np.abs(img).max()))
    nonfinite = entry_nonfinite = bool(rng.integers(0, 7) == 0)
    if nonfinite:   # a few NaN / Inf pixels: the footprint of non-finite results must be the reference's (oracle f32 restatement)
        for _ in range(int(rng.integers(1, 4))):
            img[rng.integers(0, rows), rng.integers(0, cols)] = rng.choice([np.nan, np.inf, -np.inf])
    opts = {}
    if rng.integers(0, 2):
        opts[L.OPT_STRIP_ROWS] = int(rng.choice([1, 5, 10, 19, 28, 37, 64, 131]))
    if rng.integers(0, 2):
        opts[L.OPT_BLOCK_ORDER] = int(rng.choice([0, 1000000, 1000000, 2000000, 2000000]))   # 1000000 = every XCD on its own column blocks, 2000000 = dynamic tail
    env = []                                                   # CVS_OPTS (read at every call): store policy, requesting new images ahead, workgroups per CU
    if rng.integers(0, 2):
        env.append("nt_stores=%d" % int(rng.integers(0, 2)))
    if rng.integers(0, 2):
        env.append("warm=%d" % int(rng.choice([0, 1, 3, 4, 9])))
    if rng.integers(0, 3) == 0:
        env.append("wgcap=%d" % int(rng.integers(0, 6)))
    if rng.integers(0, 3) == 0:
        env.append("lit=0")                                    # round 6: the pipeline variants with their taps from the kernel arguments instead of literals
    os.environ["CVS_OPTS"] = ",".join(env)
    if rng.integers(0, 2):
        opts[L.OPT_STATE_LAYOUT] = int(rng.integers(0, 4))    # planar / the engine's choice / one merged group / two groups pinned (round 6)
    if rng.integers(0, 4) == 0:
        opts[L.OPT_AUTOTUNE] = 1                              # otherwise tools run with what the environment says (the tuner's candidates take turns)
    entry = str(rng.choice(["setup", "setup_steer", "pipeline", "batch", "rows", "pyr"]))
    custom = bool(rng.integers(0, 6) == 0)   # a non-default (width, spacing): the two-pass fallback at every size
    cw, cs = int(rng.integers(1, 9)), float(np.float32(rng.uniform(0.3, 1.0)))
    if custom and entry in ("pipeline", "batch", "pyr"):
        entry = "setup"
    theta = float(rng.uniform(-4, 4))
    as_u8 = bool(rng.integers(0, 5) == 0) and not nonfinite and not custom and entry in ("setup", "setup_steer", "pipeline")
    if as_u8:   # 8-bit image (what the reference's callers hold): read as bytes inside the filter kernel, widened unscaled
        img = np.floor(rng.random((rows, cols)) * 256.0).clip(0, 255).astype(np.float32)
        scale = 255.0
    desc = dict(it=it, seed=seed, nonfinite=nonfinite, kind=kind, rows=rows, cols=cols, device=device, strided=strided, opts=opts, env=os.environ.get("CVS_OPTS", ""), entry=entry, theta=round(theta, 4))
    try:
        w, s = (4, 0.67) if kind == 2 else (6, 0.5)
        if custom:
            w, s = cw, cs
            desc["width"], desc["spacing"] = w, s
        nb = 7 if kind == 2 else 11
        f = cv.SteerableFiltersG2(None, w, s) if kind == 2 else cv.SteerableFiltersG4(None, w, s)
        for o, v in opts.items():
            f.set_option(o, v)
        if opts.get(L.OPT_BLOCK_ORDER) == 2000000:
            counts["dynamic tail"] = counts.get("dynamic tail", 0) + 1
        if L.OPT_STATE_LAYOUT in opts:
            counts["layout %d" % opts[L.OPT_STATE_LAYOUT]] = counts.get("layout %d" % opts[L.OPT_STATE_LAYOUT], 0) + 1
        x = as_plane(img, device, strided)
        if as_u8:
            desc["u8"] = True
            counts["8-bit image"] = counts.get("8-bit image", 0) + 1
            if torch.is_tensor(x):
                xb = torch.zeros((rows, cols + 5), dtype=torch.uint8, device="cuda")[:, 2:2 + cols] if strided else torch.empty((rows, cols), dtype=torch.uint8, device="cuda")
                xb.copy_(x.to(torch.uint8))
                x = xb
            else:
                x = np.ascontiguousarray(x).astype(np.uint8)
        truth = ora.basis(kind, img, w, s, f64=True)
        fin_t = truth[np.isfinite(truth)]
        scale = max(scale, float(np.abs(fin_t).max()) if fin_t.size else 1.0)   # wide / dense tap sets have gains of 30 and more
        if nonfinite:
            entry = "setup"
            f.setup(x, flags=cv.SETUP_BASIS)
            got = np.stack([to_np(f.basis(p)) for p in range(nb)])
            o32 = ora.basis(kind, img, w, s)
            assert np.array_equal(np.isnan(got), np.isnan(o32)), "NaN footprint"
            fin = np.isfinite(o32) & np.isfinite(truth)
            assert np.array_equal(np.isinf(got), np.isinf(o32)), "Inf footprint"
            assert np.abs(got[fin] - truth[fin]).max(initial=0.0) <= TOL * scale, "finite values beside non-finite ones"
            counts["nonfinite"] = counts.get("nonfinite", 0) + 1
            continue
        if entry == "setup" or (kind == 4 and entry in ("pipeline", "batch")):
            flags = cv.SETUP_FULL if (kind == 2 and rng.integers(0, 2)) else cv.SETUP_BASIS
            f.setup(x, flags=flags)
            got = np.stack([to_np(f.basis(p)) for p in range(nb)])
            assert np.abs(got - truth).max() <= TOL * scale, "basis"
            if flags == cv.SETUP_FULL:
                exact = bool(rng.integers(0, 3) == 0)   # CVS_OPT_ATAN_MODE = 1: atan2f instead of OpenCV's polynomial
                desc["atan_exact"] = exact
                if exact:
                    f.set_atan_mode(True)
                    f.setup(x, flags=flags)
                    got = np.stack([to_np(f.basis(p)) for p in range(nb)])
                o1, o2, o3, oth, ost = ora.g2_orientation(got, mode=ora.ATAN_EXACT if exact else ora.ATAN_CV)
                c = [to_np(v) for v in f.coefficients()]
                for a, b in zip(c + [to_np(f.getDominantOrientationStrength())], (o1, o2, o3, ost)):
                    assert np.abs(a - b).max() <= 1e-6 * scale * scale * 30, "orientation planes"
                ok = ost > 1e-3 * scale * scale
                if ok.any():
                    assert angle_diff(to_np(f.getDominantOrientationAngle()), oth, np.pi)[ok].max() <= TOL, "theta"
            g, h = f.steer(theta)
            og, oh = ora.g2_steer_scalar(got, theta) if kind == 2 else ora.g4_steer_scalar(got, theta)
            assert np.abs(to_np(g) - og).max() <= 1e-6 * scale * 4 and np.abs(to_np(h) - oh).max() <= 1e-6 * scale * 4, "steer"
        elif entry == "setup_steer":
            g, h = f.setup_steer(x, theta) if kind == 4 else f.setup_steer(x, theta, flags=cv.SETUP_BASIS if rng.integers(0, 2) else cv.SETUP_FULL)
            got = np.stack([to_np(f.basis(p)) for p in range(nb)])
            assert np.abs(got - truth).max() <= TOL * scale, "basis"
            og, oh = ora.g2_steer_scalar(got, theta) if kind == 2 else ora.g4_steer_scalar(got, theta)
            assert np.abs(to_np(g) - og).max() <= 1e-6 * scale * 4 and np.abs(to_np(h) - oh).max() <= 1e-6 * scale * 4, "fused steer"
        elif entry == "pipeline":
            persist = bool(rng.integers(0, 2))
            f.set_persist(persist)
            sel = None if persist or rng.integers(0, 2) else tuple(sorted(rng.choice(8, size=int(rng.integers(1, 8)), replace=False).tolist()))
            if sel is None:
                outs = f.pipeline(x)
            else:
                like = x if device else img
                buf = [None] * 8
                for k in sel:
                    buf[k] = torch.empty((rows, cols), device="cuda") if device else np.empty((rows, cols), np.float32)
                outs = f.pipeline(x, out=buf)
            ref = cv.SteerableFiltersG2(None)
            ref.setup(x if not strided else as_plane(img, device, False), flags=cv.SETUP_FULL)
            step = list(ref.steer(None, full=True))
            mag, ph = step[3], step[4]
            step += [ref.findEdges(mag, ph), ref.findDarkLines(mag, ph), ref.findBrightLines(mag, ph)]
            for k in range(8):
                if outs[k] is not None:
                    a, b = to_np(outs[k]), to_np(step[k])
                    assert np.array_equal(a, b, equal_nan=True), "pipeline output %d != stepwise" % k
            got = np.stack([to_np(ref.basis(p)) for p in range(7)])
            assert np.abs(got - truth).max() <= TOL * scale, "basis"
        elif entry == "batch":
            n = int(rng.integers(1, 9))
            ways = int(rng.integers(1, 5))
            os.environ["CVS_OPTS"] = ",".join(env + ["batch_ways=%d" % ways])
            frames = np.stack([make_image(rows, cols) for _ in range(n)])
            persist = bool(rng.integers(0, 2))
            f.set_persist(persist)
            fx = torch.from_numpy(frames).cuda() if device else frames
            sel = None if persist else (5, 6, 7)
            out = f.pipeline_batch(fx, outputs=sel)
            os.environ["CVS_OPTS"] = ",".join(env)
            for i in range(n):
                single = cv.SteerableFiltersG2(None).pipeline(fx[i])
                for j, k in enumerate(range(8) if sel is None else sel):
                    assert np.array_equal(to_np(out[i][j]), to_np(single[k]), equal_nan=True), "batch frame %d output %d" % (i, k)
            desc["n"], desc["ways"], desc["persist"] = n, ways, persist
        elif entry == "pyr":
            if kind == 4:
                kind = 2
                f = cv.SteerableFiltersG2(None)
                for o, v in opts.items():
                    f.set_option(o, v)
                nb, truth = 7, ora.basis(2, img, 4, 0.67, f64=True)
            want = ora.pyr_down(img)
            down = f.pyrDown(x)
            assert np.array_equal(to_np(down), want), "pyrDown"
            flags = cv.SETUP_FULL if rng.integers(0, 2) else cv.SETUP_BASIS
            nxt = f.setup_pyr(x, flags=flags)
            assert np.array_equal(to_np(nxt), want), "next level written by the filter launch"
            got = np.stack([to_np(f.basis(p)) for p in range(nb)])
            assert np.abs(got - truth).max() <= TOL * scale, "basis of the fused pyramid launch"
        else:  # rows: a row range of the image (cvs_setup_rows); needs a device image
            if rows < 2:
                continue
            lo = int(rng.integers(0, rows - 1))
            hi = int(rng.integers(lo + 1, rows + 1))
            xd = x if device else torch.from_numpy(np.ascontiguousarray(img)).cuda()
            import ctypes as C
            f._like = xd
            f._bind_stream(xd)
            pl = cv.api._plane(xd)
            f._keep = xd
            f._check(cv.lib().cvs_setup_rows(f._h, C.byref(pl), cv.SETUP_BASIS, lo, hi), "cvs_setup_rows")
            got = np.stack([to_np(f.basis(p))[lo:hi] for p in range(nb)])
            assert np.abs(got - truth[:, lo:hi]).max() <= TOL * scale, "row range"
            desc["range"] = (lo, hi)
        counts[entry] = counts.get(entry, 0) + 1
    except AssertionError as ex:
        fails += 1
        print("FAIL %s: %s" % (ex, desc), flush=True)
    except Exception as ex:
        fails += 1
        print("ERROR %s: %s: %s" % (type(ex).__name__, ex, desc), flush=True)
        if fails <= 3:
            traceback.print_exc()
    if it % 50 == 49:
        print("... %d iterations, %d failures" % (it + 1, fails), flush=True)
print("fuzz campaign: %d iterations (seed %d), %d failures; entries %s" % (iters, seed, fails, counts))
sys.exit(1 if fails else 0)
