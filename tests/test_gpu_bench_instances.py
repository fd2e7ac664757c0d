"""GPU parity tests (-m gpu) of the EXACT kernel instances bench.py times, at the sizes BASELINE.json names.

Round 2's verdict: streaming stores switch on above ~2.1 Mpix per launch, so the small-shape parity suite never ran the
template instances the benchmark measures.  Every test here names the instances it covers (template arguments of
cvs::k_basis<Bank, FLAGS, STREAM, BATCH, ONE, 4>, FLAGS = F_ORIENT 1 | F_STEER 2 | F_PIPE 4 | F_NOSTATE 8) -- the same
names profiles/r0x_kernel_stats_all_legs.csv lists -- and compares them with the oracle on row bands (top / middle /
bottom, the image borders included), stage by stage on the same upstream planes, with the tolerances of
tests/test_gpu_parity.py.  Replaces test/test.cpp:84-90 and example/steer.cpp:86-90,169 at the benchmark's sizes.
"""
import ctypes as C

import numpy as np
import pytest

from helpers import angle_diff, rand_image, smooth_image

pytestmark = pytest.mark.gpu

TOL = 1e-5
W = 4


@pytest.fixture(scope="module")
def cv():
    import cvsteer_amd
    return cvsteer_amd


def _oracle_basis_rows(ora, img, lo, hi, f64=True):
    """oracle basis planes of image rows [lo, hi): the band is filtered with W rows of context on each side (or the true
    image border), so every returned row has seen exactly the rows the whole-image filter sees"""
    n = img.shape[0]
    a, b = max(0, lo - W), min(n, hi + W)
    band = ora.basis(2, img[a:b], 4, 0.67, f64=f64)
    return band[:, lo - a:lo - a + (hi - lo)]


def _check_pipeline_rows(ora, basis, coeff, theta, strength, outs, where):
    """one band of the callers' sequence (test/test.cpp:86-90) against the oracle, stage by stage: `basis` (7 planes),
    `coeff` (c1..c3), `theta`, `strength` are the GPU's own state planes of the band, `outs` the GPU's outputs
    {g2,h2,e,magnitude,phase,edges,dark,bright} (entries may be None)"""
    o1, o2, o3, oth, ost = ora.g2_orientation(basis)
    if coeff is not None:
        for got, want in zip(tuple(coeff) + (strength,), (o1, o2, o3, ost)):
            assert np.abs(got - want).max() <= 1e-6, where
        ok = ost > 1e-3
        assert angle_diff(theta, oth, np.pi)[ok].max() <= TOL, where
    else:
        coeff, theta = (o1, o2, o3), oth
    og, oh, oe, om, op = ora.g2_steer_map(basis, theta, coeff)
    for k, want in enumerate((og, oh, oe, om)):
        if outs[k] is not None:
            assert np.abs(outs[k] - want).max() <= TOL, (where, k)
    if outs[4] is not None:
        ok = om > 1e-3
        assert angle_diff(outs[4], op, 2 * np.pi)[ok].max() <= TOL, where
    # the three feature maps from the magnitude / phase the GPU itself produced (or the oracle's, for an outputs-only run)
    mag = outs[3] if outs[3] is not None else om
    ph = outs[4] if outs[4] is not None else op
    for k, want in zip((5, 6, 7), ora.find(mag, ph)):
        if outs[k] is not None:
            assert np.abs(outs[k] - want).max() <= 1e-6 * max(1.0, float(np.abs(mag).max())), (where, k)


BANDS_1080 = ((0, 24), (528, 552), (1056, 1080))


def test_config4_batched_pipeline_32x1080p_state_kept_and_outputs_only(cv, ora):
    """BASELINE config 4 as bench.py times it (legs C4_32x1080p_pipeline_batch / C4_32x1080p_feature_maps_only, C4_e2e):
    cvs::k_basis<BankG2, 5, true, 2, true, 4> (state kept: 12 state planes + 8 outputs per frame, streaming stores, regular
    batch with one output resource per frame, single state resource) and cvs::k_basis<BankG2, 77, true, 2, true, 4> (13 | F_FEAT3: the instance specialised at compile time for
    exactly the three feature maps; outputs only: edges / dark / bright).  32 x 1080 x 1920 in ONE launch each.  Every plane of every frame is compared bit
    for bit with the single-frame pipeline() (cvs::k_basis<BankG2, 5, false, 0, true, 4>: 99.5 MB of state, plain stores),
    and frames 0, 15, 31 with the oracle on top / middle / bottom row bands."""
    import torch
    nfr = 32
    gen = torch.Generator(device="cuda").manual_seed(404)
    frames = torch.rand((nfr, 1080, 1920), generator=gen, device="cuda")
    # a structured part so that the dominant orientation is well conditioned on some of the frames
    frames[15] = 0.5 * frames[15] + torch.from_numpy(smooth_image(1080, 1920)).cuda()
    eng = cv.SteerableFiltersG2(None)
    out = torch.empty((nfr, 8, 1080, 1920), device="cuda")
    for _ in range(3):   # first call and calls on which the online tuner tries other configurations: all must agree
        out.zero_()
        eng.pipeline_batch(frames, out=out)
        torch.cuda.synchronize()
        info = eng.launch_info()
        assert info["nt_stores"] == 1, "the benchmark's instance uses streaming stores"
        if _ == 0:
            first = out.clone()
        else:
            assert torch.equal(out, first)
    del first
    single = cv.SteerableFiltersG2(None)
    for i in range(nfr):
        ref = single.pipeline(frames[i])
        assert single.launch_info()["nt_stores"] == 0       # the single 1080p frame stays under the streaming threshold
        for k in range(8):
            assert torch.equal(out[i, k], ref[k]), (i, k)
    basis_err = {}
    for i in (0, 15, 31):
        eng.select_frame(i)
        img = frames[i].cpu().numpy()
        state = [eng.basis(p) for p in range(7)]
        coeff = eng.coefficients()
        th, st = eng.getDominantOrientationAngle(), eng.getDominantOrientationStrength()
        for lo, hi in BANDS_1080:
            sl = slice(lo, hi)
            b = np.stack([s[sl].cpu().numpy() for s in state])
            assert np.abs(b - _oracle_basis_rows(ora, img, lo, hi)).max() <= TOL, (i, lo)
            d32 = float(np.abs(b - _oracle_basis_rows(ora, img, lo, hi, f64=False)).max())
            assert d32 <= TOL, (i, lo)
            basis_err[(i, lo)] = d32      # how far the GPU's basis planes are from the oracle's on this band (used below)
            _check_pipeline_rows(ora, b, tuple(c[sl].cpu().numpy() for c in coeff), th[sl].cpu().numpy(), st[sl].cpu().numpy(),
                                 [out[i, k][sl].cpu().numpy() for k in range(8)], ("state kept", i, lo))
    # outputs only (what example/steer.cpp keeps): the stateless instance, same values
    eng.set_persist(False)
    fo3 = torch.zeros((nfr, 3, 1080, 1920), device="cuda")
    for _ in range(3):
        fo3.zero_()
        eng.pipeline_batch(frames, out=fo3, outputs=(5, 6, 7))
        torch.cuda.synchronize()
        assert eng.launch_info()["nt_stores"] == 1
        assert torch.equal(fo3, out[:, 5:8])
    for i in (0, 15, 31):   # ... and directly against the oracle chain from the image band (no GPU state involved)
        img = frames[i].cpu().numpy()
        for lo, hi in BANDS_1080:
            b32 = _oracle_basis_rows(ora, img, lo, hi, f64=False)
            o1, o2, o3, oth, ost = ora.g2_orientation(b32)
            og, oh, oe, om, op = ora.g2_steer_map(b32, oth, (o1, o2, o3))
            got = [fo3[i, k][lo:hi].cpu().numpy() for k in range(3)]
            # The chain basis -> C2, C3 -> theta_dom -> (g2, h2) at theta_dom -> magnitude, phase -> magnitude * lambda(phase)
            # amplifies a basis difference d_b per pixel as follows (first order, worst-case coefficients):
            #   |dC2|, |dC3| <= 5.75 * 2 * bmax * d_b            (C3 of G2.cpp:95: coefficient sum 5.75, products of two planes)
            #   |dtheta|     <= 0.5 * sqrt(2) * |dC| / strength   (theta = atan2(C3, C2) / 2; the cut at +-pi/2 flips h2's sign only,
            #                                                      and the three maps are even in the phase)
            #   |dg|, |dh|   <= 4 d_b + 12 * bmax * |dtheta|      (sum of |weights| <= 4; |d weights / d theta| <= 3 per plane, 4 planes)
            #   |d out|      <= sqrt(2) |dg| + mag * |dphase|, mag * |dphase| <= |dg|   (lambda = cos^2, gated continuously: |lambda'| <= 1)
            # i.e. bound(pixel) = 2.5 * d_b * (4 + 98 * bmax^2 / strength) -- inversely proportional to the orientation strength,
            # which is why a single constant only ever held on "well conditioned" pixels.  d_b = the measured distance of the
            # GPU's basis planes from the oracle's on this band (<= 1e-5 asserted above, ~5e-7 in practice) + one f32 ulp of the
            # epilogue's own arithmetic.  Held on EVERY pixel with strength and magnitude above 1e-3.
            bmax = float(np.abs(b32).max())
            d_b = basis_err[(i, lo)] + 2e-7 * max(1.0, bmax)
            bound = 2.5 * d_b * (4.0 + 98.0 * bmax * bmax / np.maximum(ost, 1e-3))
            ok = (ost > 1e-3) & (om > 1e-3)
            for gk, want in zip(got, ora.find(om, op)):
                assert (np.abs(gk - want)[ok] <= bound[ok]).all(), (i, lo, float((np.abs(gk - want) / bound)[ok].max()))
                well = (ost > 1e-2) & (om > 1e-2)      # ... and the constant of rounds 1-3 where the chain is well conditioned
                assert np.abs(gk - want)[well].max() <= 5e-4, (i, lo)


def test_headline_fused_filter_steer_4096_streaming(cv, ora):
    """The headline of bench.py (`value`, M2): cvs_setup_steer(image, 0.3, CVS_SETUP_BASIS) at 4096 x 4096 =
    cvs::k_basis<BankG2, 2, true, 0, true, 4> (F_STEER, streaming stores, single state resource), in the launch
    configurations the benchmark passes through (first call: fresh-image defaults; repeats: whatever the
    launch tuner compares and keeps) and the bare defaults with the tuner off (`M2_untuned`).  g, h and the basis planes are
    compared with the oracle on top / middle / bottom bands; M1 (cvs::k_basis<BankG2, 0, true, 0, true, 4>) and M4
    (cvs::k_basis<BankG2, 1, true, 0, true, 4>) ride along on the same image."""
    import torch
    from cvsteer_amd import _lib as L
    n = 4096
    img = torch.rand((n, n), generator=torch.Generator(device="cuda").manual_seed(1234), device="cuda")
    xh = img.cpu().numpy()
    g, h = torch.empty_like(img), torch.empty_like(img)
    bands = ((0, 40), (2040, 2072), (n - 33, n))
    want = {}
    for lo, hi in bands:
        b64 = _oracle_basis_rows(ora, xh, lo, hi)
        b32 = _oracle_basis_rows(ora, xh, lo, hi, f64=False)
        want[lo] = (b64, b32)

    def check(f, what):
        gq, hq = g.cpu(), h.cpu()
        for lo, hi in bands:
            b64, b32 = want[lo]
            got = np.stack([f.basis(p)[lo:hi].cpu().numpy() for p in range(7)])
            assert np.abs(got - b64).max() <= TOL and np.abs(got - b32).max() <= TOL, (what, lo)
            og, oh = ora.g2_steer_scalar(got, 0.3)                      # same upstream planes: same op order
            assert np.abs(gq[lo:hi].numpy() - og).max() <= 1e-6 and np.abs(hq[lo:hi].numpy() - oh).max() <= 1e-6, (what, lo)
            og, oh = ora.g2_steer_scalar(b32, 0.3)                      # ... and the whole chain from the image
            assert np.abs(gq[lo:hi].numpy() - og).max() <= TOL and np.abs(hq[lo:hi].numpy() - oh).max() <= TOL, (what, lo)

    f = cv.SteerableFiltersG2(None)
    seen = set()
    for call in range(6):                                               # bench.py: 4 initialisation calls, then the timed ones
        g.zero_(); h.zero_()
        f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))
        torch.cuda.synchronize()
        info = f.launch_info()
        assert info["nt_stores"] == 1
        key = (info["block_order"], info["strip_rows"], info["warm"])
        if key not in seen:                                             # every configuration the loop passes through
            seen.add(key)
            check(f, ("call", call) + key)
    fu = cv.SteerableFiltersG2(None)                                    # `M2_untuned`
    fu.set_option(L.OPT_AUTOTUNE, 0)
    g.zero_(); h.zero_()
    fu.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))
    fu.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))
    check(fu, "untuned")
    # M1 / M4 on the same handle: basis planes again, plus the orientation planes of the band
    f.setup(img, flags=cv.SETUP_BASIS)
    f.setup(img, flags=cv.SETUP_FULL)
    coeff = f.coefficients()
    th, st = f.getDominantOrientationAngle(), f.getDominantOrientationStrength()
    for lo, hi in bands:
        got = np.stack([f.basis(p)[lo:hi].cpu().numpy() for p in range(7)])
        assert np.abs(got - want[lo][0]).max() <= TOL
        o1, o2, o3, oth, ost = ora.g2_orientation(got)
        for a, b in zip(coeff + (st,), (o1, o2, o3, ost)):
            assert np.abs(a[lo:hi].cpu().numpy() - b).max() <= 1e-6
        assert angle_diff(th[lo:hi].cpu().numpy(), oth, np.pi)[ost > 1e-3].max() <= TOL


def test_m5_pipeline_4096_and_g4_streaming_instances(cv, ora):
    """bench.py legs M5_pipeline (cvs::k_basis<BankG2, 5, true, 0, true, 4>) and M6 (cvs::k_basis_pair<BankG4G, BankG4H, 0 / 2,
    true, true>) at 4096 x 4096 against oracle bands."""
    import torch
    n = 4096
    img = torch.rand((n, n), generator=torch.Generator(device="cuda").manual_seed(77), device="cuda")
    xh = img.cpu().numpy()
    f = cv.SteerableFiltersG2(None)
    outs = None
    for _ in range(3):
        outs = f.pipeline(img)
    torch.cuda.synchronize()
    assert f.launch_info()["nt_stores"] == 1
    coeff = f.coefficients()
    th, st = f.getDominantOrientationAngle(), f.getDominantOrientationStrength()
    for lo, hi in ((0, 24), (2000, 2024), (n - 24, n)):
        sl = slice(lo, hi)
        b = np.stack([f.basis(p)[sl].cpu().numpy() for p in range(7)])
        assert np.abs(b - _oracle_basis_rows(ora, xh, lo, hi)).max() <= TOL
        _check_pipeline_rows(ora, b, tuple(c[sl].cpu().numpy() for c in coeff), th[sl].cpu().numpy(), st[sl].cpu().numpy(),
                             [o[sl].cpu().numpy() for o in outs], ("M5", lo))
    f4 = cv.SteerableFiltersG4(None)
    g4 = h4 = None
    for _ in range(3):
        g4, h4 = f4.setup_steer(img, 0.3)
    torch.cuda.synchronize()
    assert f4.launch_info()["nt_stores"] == 1
    for lo, hi in ((0, 20), (3000, 3020), (n - 20, n)):
        a, b_ = max(0, lo - 6), min(n, hi + 6)
        band = ora.basis(4, xh[a:b_], 6, 0.5, f64=True)[:, lo - a:lo - a + hi - lo]
        got = np.stack([f4.basis(p)[lo:hi].cpu().numpy() for p in range(11)])
        assert np.abs(got - band).max() <= TOL
        og, oh = ora.g4_steer_scalar(got, 0.3)
        assert np.abs(g4[lo:hi].cpu().numpy() - og).max() <= 1e-6 and np.abs(h4[lo:hi].cpu().numpy() - oh).max() <= 1e-6


def test_m3_steer_kernels_4096_streaming(cv, ora):
    """bench.py legs M3_steer_scalar (cvs::k_point<(PointOp)1, 4, true, true>: scalar theta, 7 planes in, g / h out) and
    M3_steer_map_full (cvs::k_point<(PointOp)2, 4, true, true>: theta_dom per pixel, g / h / e / magnitude / phase out) at
    4096 x 4096 -- four pixels per lane, streaming stores, nontemporal loads -- against the oracle on top / middle / bottom
    bands, from the GPU's own state planes of the band (same op order: 1e-6) and stage by stage."""
    import torch
    n = 4096
    img = torch.rand((n, n), generator=torch.Generator(device="cuda").manual_seed(4321), device="cuda")
    f = cv.SteerableFiltersG2(None)
    f.setup(img, flags=cv.SETUP_FULL)
    g, h = torch.zeros_like(img), torch.zeros_like(img)
    outs5 = [torch.zeros_like(img) for _ in range(5)]
    for _ in range(2):
        f.steer(0.3, out=(g, h))
        f.steer(None, full=True, out=outs5)
    torch.cuda.synchronize()
    coeff = f.coefficients()
    th = f.getDominantOrientationAngle()
    for lo, hi in ((0, 16), (2048, 2064), (n - 16, n)):
        sl = slice(lo, hi)
        b = np.stack([f.basis(p)[sl].cpu().numpy() for p in range(7)])
        og, oh = ora.g2_steer_scalar(b, 0.3)
        assert np.abs(g[sl].cpu().numpy() - og).max() <= 1e-6 and np.abs(h[sl].cpu().numpy() - oh).max() <= 1e-6, lo
        c = tuple(x[sl].cpu().numpy() for x in coeff)
        mg, mh, me, mm, mp = ora.g2_steer_map(b, th[sl].cpu().numpy(), c)
        got = [o[sl].cpu().numpy() for o in outs5]
        for k, want in enumerate((mg, mh, me, mm)):
            assert np.abs(got[k] - want).max() <= TOL, (lo, k)
        assert angle_diff(got[4], mp, 2 * np.pi)[mm > 1e-3].max() <= TOL, lo


def _small_shapes():
    rng = np.random.default_rng(2025)
    shapes = [(13, 5), (14, 64), (19, 65), (28, 128), (38, 191), (57, 257), (120, 200)]
    return shapes + [(int(rng.integers(13, 140)), int(rng.integers(5, 330))) for _ in range(8)]


def test_small_shape_fuzz_with_streaming_stores_forced(cv, ora, monkeypatch):
    """CVS_OPTS nt_stores=1 (always nontemporal) on the small shapes of the parity fuzz, so that the STREAM = true
    instances meet the oracle in every launch form: single image with the single state resource
    (cvs::k_basis<BankG2, 0 / 1 / 2 / 3 / 5 / 77, true, 0, true, 4>), the per-plane form of a row-range launch
    (cvs::k_basis<BankG2, 0 / 1, true, 0, false, 4>), frames from a device table (cvs::k_basis<BankG2, 5 / 13, true, 1, true, 4>)
    and the regular batch with one output resource per frame (cvs::k_basis<BankG2, 5 / 77, true, 2, true, 4>; the generic outputs-only instance 13 through a two-map request); G4 as
    cvs::k_basis_pair<BankG4G, BankG4H, 0 / 2, true, true / false>."""
    import torch
    monkeypatch.setenv("CVS_OPTS", "nt_stores=1")
    for rows, cols in _small_shapes():
        img = rand_image(rows, cols, seed=rows * 1000 + cols) + (smooth_image(rows, cols) if rows > 30 else 0)
        img = img.astype(np.float32)
        dev = torch.from_numpy(img).cuda()
        truth = ora.basis(2, img, 4, 0.67, f64=True)
        f = cv.SteerableFiltersG2(None)
        for flags in (cv.SETUP_BASIS, cv.SETUP_FULL):                      # FLAGS 0, 1
            f.setup(dev, flags=flags)
            assert f.launch_info()["nt_stores"] == 1
            b = np.stack([f.basis(p).cpu().numpy() for p in range(7)])
            assert np.abs(b - truth).max() <= TOL, (rows, cols, flags)
        for flags in (cv.SETUP_BASIS, cv.SETUP_FULL):                      # FLAGS 2, 3 (fused scalar steer)
            gq, hq = f.setup_steer(dev, -0.7, flags=flags)
            b = np.stack([f.basis(p).cpu().numpy() for p in range(7)])
            og, oh = ora.g2_steer_scalar(b, -0.7)
            assert np.abs(gq.cpu().numpy() - og).max() <= 1e-6 and np.abs(hq.cpu().numpy() - oh).max() <= 1e-6, (rows, cols)
        outs = f.pipeline(dev)                                             # FLAGS 5
        b = np.stack([f.basis(p).cpu().numpy() for p in range(7)])
        assert np.abs(b - truth).max() <= TOL
        _check_pipeline_rows(ora, b, tuple(c.cpu().numpy() for c in f.coefficients()), f.getDominantOrientationAngle().cpu().numpy(),
                             f.getDominantOrientationStrength().cpu().numpy(), [o.cpu().numpy() for o in outs], (rows, cols, "pipe"))
        f.set_persist(False)                                               # FLAGS 13
        o3 = f.pipeline(dev, out=[None] * 5 + [torch.empty_like(dev) for _ in range(3)])
        for k in (5, 6, 7):
            assert torch.equal(o3[k], outs[k]), (rows, cols, k)
        f.set_persist(True)
        if rows >= 13 and cols >= 5:                                       # the per-plane form: a row range of the image
            lo, hi = rows // 3, max(rows // 3 + 1, 2 * rows // 3)
            fr = cv.SteerableFiltersG2(None)
            fr._like = dev
            fr._bind_stream(dev)
            pl = cv.api._plane(dev)
            for flags in (cv.SETUP_BASIS, cv.SETUP_FULL):
                fr._check(cv.lib().cvs_setup_rows(fr._h, C.byref(pl), flags, lo, hi), "cvs_setup_rows")
                b = np.stack([fr.basis(p)[lo:hi].cpu().numpy() for p in range(7)])
                assert np.abs(b - truth[:, lo:hi]).max() <= TOL, (rows, cols, "rows")
        # batches: a regular block (BATCH = 2) and unrelated planes from a device table (BATCH = 1)
        nb = 3
        block = torch.stack([dev, dev.flip(0).contiguous(), (dev * 0.5 + 0.1)]).contiguous()
        loose = [block[0].clone(), torch.empty((rows + 3, cols + 5), device="cuda")[1:rows + 1, 2:cols + 2], block[2].clone()]
        loose[1].copy_(block[1])
        for persist in (True, False):
            sel = tuple(range(8)) if persist else (5, 6, 7)
            fb = cv.SteerableFiltersG2(None)
            fb.set_persist(persist)
            ob = fb.pipeline_batch(block, outputs=sel)
            fl = cv.SteerableFiltersG2(None)
            fl.set_persist(persist)
            ol = fl.pipeline_batch(loose, outputs=sel)
            torch.cuda.synchronize()
            assert torch.equal(ob, ol), (rows, cols, persist)
            for i in range(nb):
                fi = block[i].cpu().numpy()
                bi = ora.basis(2, fi, 4, 0.67)
                if persist:
                    fb.select_frame(i)
                    gb = np.stack([fb.basis(p).cpu().numpy() for p in range(7)])
                    assert np.abs(gb - bi).max() <= TOL
                    _check_pipeline_rows(ora, gb, tuple(c.cpu().numpy() for c in fb.coefficients()),
                                         fb.getDominantOrientationAngle().cpu().numpy(), fb.getDominantOrientationStrength().cpu().numpy(),
                                         [ob[i, k].cpu().numpy() for k in range(8)], (rows, cols, "batch", i))
                else:
                    ref = cv.SteerableFiltersG2(None).pipeline(block[i])
                    for j, k in enumerate(sel):
                        assert torch.equal(ob[i, j], ref[k]), (rows, cols, i, k)
        # the GENERIC outputs-only instance (FLAGS 13: any other subset than the three feature maps): phase + bright lines
        fg = cv.SteerableFiltersG2(None)
        fg.set_persist(False)
        og = fg.pipeline_batch(block, outputs=(4, 7))
        for i in range(nb):
            ref = cv.SteerableFiltersG2(None).pipeline(block[i])
            assert torch.equal(og[i, 0], ref[4]) and torch.equal(og[i, 1], ref[7]), (rows, cols, i)
        # G4: both halves in one launch, streaming, with and without the fused steer
        f4 = cv.SteerableFiltersG4(None)
        f4.setup(dev)
        t4 = ora.basis(4, img, 6, 0.5, f64=True)
        b4 = np.stack([f4.basis(p).cpu().numpy() for p in range(11)])
        assert np.abs(b4 - t4).max() <= TOL, (rows, cols, "g4")
        g4, h4 = f4.setup_steer(dev, 0.9)
        b4 = np.stack([f4.basis(p).cpu().numpy() for p in range(11)])
        og, oh = ora.g4_steer_scalar(b4, 0.9)
        assert np.abs(g4.cpu().numpy() - og).max() <= 1e-6 and np.abs(h4.cpu().numpy() - oh).max() <= 1e-6


def test_8bit_inputs_with_streaming_stores_forced_equal_widened(cv, monkeypatch):
    """The 8-bit legs of bench.py (M2 from 8-bit images; 32 x 1080p three maps from 8-bit frames) run the byte-reading instances
    with streaming stores -- cvs::k_basis<BankG2, 2, true, 0, true, 4, true> and cvs::k_basis<BankG2, 77, true, 2, true, 4, true>.
    CVS_OPTS nt_stores=1 on small and medium shapes: bit-identical to the same pixels widened to f32 first (whose instances meet
    the oracle in the test above)."""
    import torch
    monkeypatch.setenv("CVS_OPTS", "nt_stores=1")
    gen = torch.Generator(device="cuda").manual_seed(77)
    for rows, cols in ((185, 256), (131, 1021), (540, 960)):
        u8 = (torch.rand((rows, cols), device="cuda", generator=gen) * 256).to(torch.uint8)
        f32 = u8.to(torch.float32)
        a, b = cv.SteerableFiltersG2(None), cv.SteerableFiltersG2(None)
        ga, ha = a.setup_steer(u8, 0.3, flags=cv.SETUP_BASIS)
        assert a.launch_info()["nt_stores"] == 1
        gb, hb = b.setup_steer(f32, 0.3, flags=cv.SETUP_BASIS)
        assert torch.equal(ga, gb) and torch.equal(ha, hb)
        for p in range(7):
            assert torch.equal(a._state(p), b._state(p)), ("basis", p)
        block = (torch.rand((5, rows, cols), device="cuda", generator=gen) * 256).to(torch.uint8)
        fa_, fb_ = cv.SteerableFiltersG2(None), cv.SteerableFiltersG2(None)
        fa_.set_persist(False)
        fb_.set_persist(False)
        oa = fa_.pipeline_batch(block, outputs=(5, 6, 7))
        ob = fb_.pipeline_batch(block.to(torch.float32), outputs=(5, 6, 7))
        assert torch.equal(oa, ob), (rows, cols)


def test_overlapped_host_path_against_the_oracle(cv, ora):
    """the band-wise upload / filter / download path (host_pipeline in cvs_host.cpp; a 1-Mpix-and-more host image with host
    outputs: cvs::k_basis<BankG2, 2 / 3, STREAM, 0, false, 4> per band) directly against the oracle, not only against the
    device path: g, h and basis planes on a band that straddles two upload bands, and on the image borders."""
    rng = np.random.default_rng(32)
    img = rng.random((1531, 1100), dtype=np.float32)
    f = cv.SteerableFiltersG2(None)
    g, h = f.setup_steer(img, 0.3, flags=cv.SETUP_FULL)
    assert isinstance(g, np.ndarray)
    per = -(-1531 // 8)
    per = -(-per // 10) * 10          # bands are whole strips: the seam between band 0 and band 1 lies near here
    for lo, hi in ((0, 30), (max(0, per - 25), per + 25), (1531 - 30, 1531)):
        b32 = _oracle_basis_rows(ora, img, lo, hi, f64=False)
        b64 = _oracle_basis_rows(ora, img, lo, hi)
        got = np.stack([f.basis(p)[lo:hi] for p in range(7)])
        assert np.abs(got - b64).max() <= TOL and np.abs(got - b32).max() <= TOL
        og, oh = ora.g2_steer_scalar(got, 0.3)
        assert np.abs(g[lo:hi] - og).max() <= 1e-6 and np.abs(h[lo:hi] - oh).max() <= 1e-6
        o1, o2, o3, oth, ost = ora.g2_orientation(got)
        assert angle_diff(f.getDominantOrientationAngle()[lo:hi], oth, np.pi)[ost > 1e-3].max() <= TOL
