#!/usr/bin/env python3
"""tools/fuzz_points.py [iterations] [seed] -- randomised differential run of the PER-PIXEL stages (magnitude / phase, wrap,
phaseWeights, find*, steer at a scalar / at a theta map, G2 and G4) against the oracle: random shapes (1 x 1 ... 300 x 700,
dword and 16-byte column counts), host or device planes, strided views, values drawn from normal / tiny / huge ranges with
zeros, signed zeros, NaN and +-Inf sprinkled in, angles well outside (-pi, pi].  Complements tools/fuzz_campaign.py (the filter
bank).  Prints a line per failure and a summary."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cvsteer_amd as cv
from oracle import pyoracle as ora

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)


def angle_diff(a, b, period):
    d = np.abs(a - b) % period
    return np.minimum(d, period - d)


def values(shape, special=True):
    mode = rng.integers(0, 4)
    if mode == 0:
        a = rng.standard_normal(shape).astype(np.float32)
    elif mode == 1:
        a = (rng.standard_normal(shape) * 1e-6).astype(np.float32)
    elif mode == 2:
        a = (rng.standard_normal(shape) * 1e4).astype(np.float32)
    else:
        a = (rng.random(shape, dtype=np.float32) * 2 - 1).astype(np.float32)
    if special and a.size:
        flat = a.reshape(-1)
        for v in (0.0, -0.0, np.nan, np.inf, -np.inf):
            if rng.integers(0, 2):
                flat[rng.integers(0, flat.size)] = v
    return a


def plane(a, device, strided):
    rows, cols = a.shape
    if strided:
        pl, pr = int(rng.integers(0, 5)), int(rng.integers(1, 40))
        big = np.full((rows, cols + pl + pr), -3.25, np.float32)
        big[:, pl:pl + cols] = a
        return torch.from_numpy(big).cuda()[:, pl:pl + cols] if device else big[:, pl:pl + cols]
    return torch.from_numpy(a).cuda() if device else a


def to_np(x):
    return x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def close(got, want, tol, what, rel=None):
    got, want = to_np(got), np.asarray(want)
    assert np.array_equal(np.isnan(got), np.isnan(want)), what + ": NaN footprint"
    inf = np.isinf(want)
    assert np.array_equal(got[inf], want[inf]), what + ": Inf values"
    ok = np.isfinite(want)
    if ok.any():
        scale = np.maximum(1.0, np.abs(want[ok])) if rel else 1.0
        assert (np.abs(got[ok] - want[ok]) / scale).max() <= tol, what


fails = 0
counts = {}
f2 = cv.SteerableFiltersG2(None)
for it in range(iters):
    rows, cols = int(rng.integers(1, 300)), int(rng.integers(1, 700))
    if rng.integers(0, 3) == 0:
        cols = int(4 * rng.integers(1, 160))
    device, strided = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    stage = str(rng.choice(["magphase", "wrap", "weights", "find", "steer2", "steer4"]))
    desc = dict(it=it, seed=seed, stage=stage, rows=rows, cols=cols, device=device, strided=strided)
    try:
        if stage == "magphase":
            g, h = values((rows, cols)), values((rows, cols))
            m, p = f2.computeMagnitudeAndPhase(plane(g, device, strided), plane(h, device, strided))
            om, op = ora.mag_phase(g, h)
            close(m, om, 2e-7, "magnitude", rel=True)
            assert np.array_equal(to_np(p), op), "phase (same polynomial, same op order: bit for bit)"
        elif stage == "wrap":
            a = (values((rows, cols)) * np.float32(rng.choice([1.0, 3.0, 10.0]))).astype(np.float32)
            got = f2.wrap(plane(a, device, strided))
            assert np.array_equal(to_np(got), ora.wrap(a), equal_nan=True), "wrap"
        elif stage == "weights":
            ph = (rng.random((rows, cols), dtype=np.float32) * np.float32(2 * np.pi) - np.float32(np.pi)).astype(np.float32)
            if rng.integers(0, 3) == 0:
                ph = (ph * 6).astype(np.float32)
            phi, sg = float(rng.choice([0.0, np.pi / 2, np.pi, rng.uniform(-3, 3)])), bool(rng.integers(0, 2))
            desc.update(phi=round(phi, 4), signum=sg)
            lam = f2.phaseWeights(plane(ph, device, strided), phi, sg)
            close(lam, ora.phase_weights(ph, phi, sg), 2e-5 if np.abs(ph).max() > 4 else 1e-6, "phaseWeights")
        elif stage == "find":
            ph = (rng.random((rows, cols), dtype=np.float32) * np.float32(2 * np.pi) - np.float32(np.pi)).astype(np.float32)
            e = np.abs(values((rows, cols), special=False))
            outs = f2.find(plane(e, device, strided), plane(ph, device, strided))
            for k, (got, want) in enumerate(zip(outs, ora.find(e, ph))):
                close(got, want, 1e-6 * max(1.0, float(e.max())), "find %d" % k)
        else:
            kind = 2 if stage == "steer2" else 4
            img = rng.random((max(rows, 20), max(cols, 8)), dtype=np.float32)
            f = cv.SteerableFiltersG2(img) if kind == 2 else cv.SteerableFiltersG4(img)
            nb = 7 if kind == 2 else 11
            b = np.stack([to_np(f.basis(p)) for p in range(nb)])
            if rng.integers(0, 2):
                theta = float(rng.uniform(-20, 20))
                desc["theta"] = round(theta, 4)
                if kind == 2:
                    outs = f.steer(theta, full=True)
                    c = tuple(to_np(v) for v in f.coefficients())
                    want = ora.g2_steer_scalar(b, theta, c)
                    for k in range(4):
                        close(outs[k], want[k], 2e-6 * 10, "steer scalar output %d" % k)
                    ok = want[3] > 1e-3
                    if ok.any():
                        assert angle_diff(to_np(outs[4]), want[4], 2 * np.pi)[ok].max() <= 2e-5, "steer scalar phase"
                else:
                    g, h = f.steer(theta)
                    og, oh = ora.g4_steer_scalar(b, theta)
                    close(g, og, 2e-5, "g4 steer g"); close(h, oh, 2e-5, "g4 steer h")
            else:
                tmap = (rng.random(img.shape, dtype=np.float32) * np.float32(rng.choice([3.0, 12.0, 40.0])) - np.float32(1.5)).astype(np.float32)
                if kind == 2:
                    g, h = f.steer(tmap)
                    og, oh = ora.g2_steer_map(b, tmap)[:2]
                else:
                    g, h = f.steer(tmap)
                    og, oh = ora.g4_steer_map(b, tmap)
                tol = 2e-5 if np.abs(tmap).max() <= 8 else 2e-4
                close(g, og, tol, "steer map g"); close(h, oh, tol, "steer map h")
        counts[stage] = counts.get(stage, 0) + 1
    except AssertionError as ex:
        fails += 1
        print("FAIL %s: %s" % (ex, desc), flush=True)
    except Exception as ex:
        fails += 1
        print("ERROR %s: %s: %s" % (type(ex).__name__, ex, desc), flush=True)
        if fails <= 3:
            traceback.print_exc()
print("fuzz points: %d iterations (seed %d), %d failures; stages %s" % (iters, seed, fails, counts))
sys.exit(1 if fails else 0)
