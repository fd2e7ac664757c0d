"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle.

Tolerances (float32 path; BASELINE.json north_star: <= 1e-5 max-abs vs the CPU path):
  * basis planes on [0,1) inputs ............ <= 1e-5 abs vs f64-accumulated truth AND vs the
                                               f32 OpenCV-order restatement
  * basis planes on the 0..255 fish ......... <= 1e-5 * max|plane| (hard part 2)
  * stages fed the SAME upstream planes ..... <= 1e-6 (same op order; usually bit-identical)
  * angles (theta, phase) ................... <= 1e-5 rad modulo the branch cut, where the
                                               vector length is > 1e-3
"""
import ctypes as C
import io
import os

import numpy as np
import pytest

from helpers import EDGE_SHAPES, angle_diff, rand_image, smooth_image
from cvsteer_amd import _lib as L_

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def cv():
    import cvsteer_amd
    return cvsteer_amd


def _basis_stack(f, n):
    return np.stack([f.basis(p) for p in range(n)])


# ----------------------------------------------------------------------------- basis (K1)
@pytest.mark.parametrize("shape", EDGE_SHAPES + [(200, 300), (129, 513)])
def test_g2_basis_matches_oracle(cv, ora, shape):
    img = rand_image(*shape, seed=100 + shape[0] * 7 + shape[1])
    f = cv.SteerableFiltersG2(img, 4, 0.67)
    got = _basis_stack(f, 7)
    assert np.abs(got - ora.basis(2, img, 4, 0.67, f64=True)).max() <= TOL
    assert np.abs(got - ora.basis(2, img, 4, 0.67)).max() <= TOL


@pytest.mark.parametrize("shape", EDGE_SHAPES + [(200, 300)])
def test_g4_basis_matches_oracle(cv, ora, shape):
    img = rand_image(*shape, seed=200 + shape[0] * 7 + shape[1])
    f = cv.SteerableFiltersG4(img, 6, 0.5)
    got = _basis_stack(f, 11)
    assert np.abs(got - ora.basis(4, img, 6, 0.5, f64=True)).max() <= TOL
    assert np.abs(got - ora.basis(4, img, 6, 0.5)).max() <= TOL


@pytest.mark.parametrize("kind,w,s", [(2, 6, 0.5), (2, 3, 0.9), (4, 8, 0.4), (4, 4, 0.75), (2, 1, 1.0)])
def test_generic_width_path(cv, ora, kind, w, s):
    img = rand_image(37, 91, seed=5)
    cls = cv.SteerableFiltersG2 if kind == 2 else cv.SteerableFiltersG4
    f = cls(img, w, s)
    n = 7 if kind == 2 else 11
    assert np.abs(_basis_stack(f, n) - ora.basis(kind, img, w, s, f64=True)).max() <= TOL
    for i in range(n):
        assert np.array_equal(f.taps(i), ora.make_taps(kind, i, w, s))


def test_basis_on_fish_relative_tolerance(cv, ora, fish):
    f = cv.SteerableFiltersG2(fish, 4, 0.67)
    got = _basis_stack(f, 7)
    truth = ora.basis(2, fish, 4, 0.67, f64=True)
    for p in range(7):
        assert np.abs(got[p] - truth[p]).max() <= TOL * np.abs(truth[p]).max()


def test_strip_rows_do_not_change_results(cv):
    img = rand_image(150, 200, seed=9)
    ref = None
    for sr in (0, 1, 10, 37, 64, 1000):
        f = cv.SteerableFiltersG2(None, 4, 0.67)
        f.set_strip_rows(sr)
        f.setup(img)
        got = _basis_stack(f, 7)
        if ref is None:
            ref = got
        assert np.array_equal(got, ref), sr


def test_non_contiguous_input_step(cv, ora):
    big = rand_image(80, 120, seed=21)
    roi = big[5:70, 9:100]  # a cv::Mat ROI: step > cols*4
    f = cv.SteerableFiltersG2(roi, 4, 0.67)
    assert np.abs(_basis_stack(f, 7) - ora.basis(2, np.ascontiguousarray(roi), 4, 0.67, f64=True)).max() <= TOL


# ----------------------------------------------------------------------------- orientation
def test_orientation_same_basis_in(cv, ora):
    """C1..C3 / strength / theta from the GPU's own basis planes: same op order as the oracle"""
    for img in (rand_image(64, 96, seed=3), smooth_image(90, 130)):
        f = cv.SteerableFiltersG2(img, 4, 0.67)
        b = _basis_stack(f, 7)
        c1, c2, c3 = f.coefficients()
        th, st = f.getDominantOrientationAngle(), f.getDominantOrientationStrength()
        o1, o2, o3, oth, ost = ora.g2_orientation(b)
        for got, want in ((c1, o1), (c2, o2), (c3, o3), (st, ost)):
            assert np.abs(got - want).max() <= 1e-6
        ok = ost > 1e-3
        assert angle_diff(th, oth, np.pi)[ok].max() <= TOL
        assert th.max() <= np.float32(np.pi / 2) + 1e-6 and th.min() > -np.pi / 2 - 1e-6


def test_orientation_end_to_end_smooth(cv, ora):
    img = smooth_image(128, 160)
    f = cv.SteerableFiltersG2(img, 4, 0.67)
    o = ora.g2_orientation(ora.basis(2, img, 4, 0.67))
    c = f.coefficients()
    for got, want in zip(c, o[:3]):
        assert np.abs(got - want).max() <= TOL
    assert np.abs(f.getDominantOrientationStrength() - o[4]).max() <= TOL
    ok = o[4] > 1e-2
    # theta is ill-conditioned where strength is small: d(theta) ~ d(C)/(2*strength)
    assert angle_diff(f.getDominantOrientationAngle(), o[3], np.pi)[ok].max() <= 2e-4


def test_exact_atan_mode(cv, ora):
    img = smooth_image(64, 80)
    f = cv.SteerableFiltersG2(None, 4, 0.67)
    f.set_atan_mode(True)
    f.setup(img)
    b = _basis_stack(f, 7)
    oth = ora.g2_orientation(b, ora.ATAN_EXACT)[3]
    ost = ora.g2_orientation(b, ora.ATAN_EXACT)[4]
    assert angle_diff(f.getDominantOrientationAngle(), oth, np.pi)[ost > 1e-3].max() <= TOL


# ----------------------------------------------------------------------------- steer
@pytest.mark.parametrize("theta", [0.0, 0.3, -1.2, 1.5707964, 3.0])
def test_g2_steer_scalar(cv, ora, theta):
    img = rand_image(70, 100, seed=31)
    f = cv.SteerableFiltersG2(img, 4, 0.67)
    b = _basis_stack(f, 7)
    c = f.coefficients()
    g, h = f.steer(theta)
    og, oh = ora.g2_steer_scalar(b, theta)
    assert np.abs(g - og).max() <= 1e-6 and np.abs(h - oh).max() <= 1e-6
    g, h, e, m, p = f.steer(theta, full=True)
    og, oh, oe, om, op = ora.g2_steer_scalar(b, theta, c)
    for got, want in ((g, og), (h, oh), (e, oe), (m, om)):
        assert np.abs(got - want).max() <= 1e-6
    assert angle_diff(p, op, 2 * np.pi)[om > 1e-3].max() <= TOL


def test_g2_steer_map_given_theta(cv, ora):
    img = rand_image(61, 77, seed=32)
    f = cv.SteerableFiltersG2(img, 4, 0.67)
    b = _basis_stack(f, 7)
    c = f.coefficients()
    theta = (np.random.default_rng(1).random(img.shape, dtype=np.float32) - 0.5) * np.float32(np.pi)
    g, h, e, m, p = f.steer(theta, full=True)
    og, oh, oe, om, op = ora.g2_steer_map(b, theta, c)
    for got, want in ((g, og), (h, oh), (e, oe), (m, om)):
        assert np.abs(got - want).max() <= TOL
    assert angle_diff(p, op, 2 * np.pi)[om > 1e-3].max() <= 2e-5
    g2, h2 = f.steer(theta)
    assert np.array_equal(g2, g) and np.array_equal(h2, h)


def test_g2_steer_at_dominant_orientation(cv, ora):
    img = smooth_image(96, 112)
    f = cv.SteerableFiltersG2(img, 4, 0.67)
    th = f.getDominantOrientationAngle()
    a = f.steer(None, full=True)
    b = f.steer(th, full=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_g2_steer_point(cv, ora):
    img = rand_image(40, 56, seed=33)
    f = cv.SteerableFiltersG2(img, 4, 0.67)
    b = _basis_stack(f, 7)
    c = f.coefficients()
    for (x, y, th) in ((0, 0, 0.3), (55, 39, -0.7), (17, 23, 1.1)):
        got = np.array(f.steer_point((x, y), th, full=True), np.float32)
        want = ora.g2_steer_point(b, c, y, x, th)
        assert np.abs(got - want).max() <= 1e-6
        assert f.steer_point((x, y), th) == tuple(float(v) for v in got[:2])


@pytest.mark.parametrize("theta", [0.0, 0.3, -2.0])
def test_g4_steer(cv, ora, theta):
    img = rand_image(50, 70, seed=41)
    f = cv.SteerableFiltersG4(img, 6, 0.5)
    b = _basis_stack(f, 11)
    g, h = f.steer(theta)
    og, oh = ora.g4_steer_scalar(b, theta)
    assert np.abs(g - og).max() <= 1e-6 and np.abs(h - oh).max() <= 1e-6
    tmap = (np.random.default_rng(2).random(img.shape, dtype=np.float32) - 0.5) * np.float32(2 * np.pi)
    g, h = f.steer(tmap)
    og, oh = ora.g4_steer_map(b, tmap)
    assert np.abs(g - og).max() <= TOL and np.abs(h - oh).max() <= TOL


def test_setup_steer_fused_equals_two_step(cv):
    img = rand_image(130, 190, seed=51)
    for cls, n in ((cv.SteerableFiltersG2, 7), (cv.SteerableFiltersG4, 11)):
        f = cls(img)
        g0, h0 = f.steer(0.3)
        b0 = _basis_stack(f, n)
        f2 = cls(None)
        g1, h1 = f2.setup_steer(img, 0.3)
        assert np.array_equal(_basis_stack(f2, n), b0)
        assert np.array_equal(g0, g1) and np.array_equal(h0, h1)


# ----------------------------------------------------------------------------- mag/phase, weights, find
def test_mag_phase_weights_find(cv, ora):
    rng = np.random.default_rng(61)
    g = rng.standard_normal((45, 67)).astype(np.float32)
    h = rng.standard_normal((45, 67)).astype(np.float32)
    g[0, 0] = h[0, 0] = 0.0
    g[1, 1] = np.nan
    f = cv.SteerableFiltersG2(None)
    m, p = f.computeMagnitudeAndPhase(g, h)
    om, op = ora.mag_phase(g, h)
    assert np.allclose(m, om, atol=1e-6, equal_nan=True)
    assert np.array_equal(p, op)           # same polynomial, same op order
    assert p[1, 1] == 0.0                  # patchNaNs
    for phi, sg in ((np.pi / 2, False), (0.0, True), (np.pi, True), (0.7, True), (-2.0, False)):
        lam = f.phaseWeights(op, phi, sg)
        assert np.abs(lam - ora.phase_weights(op, phi, sg)).max() <= 1e-6
    # find* evaluate ONE cos / sin pair of |phase| for the three maps (phase_lambda3, cvs_device_math.h): against the
    # reference's three separate phaseWeights the weights differ by what its float steps (|p| - float(pi/2), p - float(pi))
    # round away -- measured here on a unit energy plane, stage tolerance 1e-6 -- and the products scale with the energy
    ones = np.ones_like(g)
    for got, want in zip(f.find(ones, op), ora.find(ones, op)):
        assert np.abs(got - want).max() <= 1e-6
    e = np.abs(g)
    e[1, 1] = 1.0
    outs = f.find(e, op)
    for got, want in zip(outs, ora.find(e, op)):
        assert np.abs(got - want).max() <= 1e-6 * max(1.0, float(e.max()))
    # phases beyond (-pi, pi] (a caller's own plane): the identities behind the shared evaluation hold for every angle
    wide = (rng.random((45, 67), dtype=np.float32) * 40 - 20).astype(np.float32)
    for got, want in zip(f.find(ones, wide), ora.find(ones, wide)):
        assert np.abs(got - want).max() <= 2e-5      # the reference's own float steps on |p| ~ 20 are good to ~1 ulp(20) = 2e-6 rad
    assert np.array_equal(f.findEdges(e, op), outs[0])
    assert np.array_equal(f.findDarkLines(e, op), outs[1])
    assert np.array_equal(f.findBrightLines(e, op), outs[2])


def test_wrap_standalone_entry(cv, ora):
    """SteerableFilters::wrap (SteerableFilters.cpp:46-51) through its own entry point cvs_wrap / OP_WRAP:
    out = angle > pi ? angle - 2 pi : angle with both constants narrowed to f32 -- bit for bit against the
    oracle, on host planes, device planes (dword and 16-byte paths) and in place, around pi, 2 pi and NaN."""
    import torch
    pi32 = np.float32(np.pi)
    special = np.array([0.0, -0.0, np.pi, np.nextafter(pi32, np.float32(4)), np.nextafter(pi32, np.float32(0)), -np.pi,
                        2 * np.pi, np.nextafter(np.float32(2 * np.pi), np.float32(7)), 3 * np.pi, -3 * np.pi, 6.2831855, 1e30,
                        -1e30, np.inf, -np.inf, np.nan, 3.1415925, 3.1415927, 3.1415930], np.float32)
    rng = np.random.default_rng(77)
    for shape in ((1, special.size), (37, 91), (64, 256)):           # 91 columns: dword path; 256: dwordx4 path
        a = (rng.random(shape, dtype=np.float32) * np.float32(4 * np.pi) - np.float32(2 * np.pi)).astype(np.float32)
        a.flat[:special.size] = special[:a.size]
        want = ora.wrap(a)
        assert want.shape == a.shape
        # the definition itself (not only the oracle): values above pi (as f32) move down by 2 pi (as f32)
        chk = np.where(a > pi32, a + np.float32(-2.0 * np.pi), a)
        assert np.array_equal(want, chk, equal_nan=True)
        f = cv.SteerableFiltersG2(None)
        got_host = f.wrap(a)
        assert np.array_equal(got_host, want, equal_nan=True), shape
        ta = torch.from_numpy(a).cuda()
        got_dev = f.wrap(ta)
        assert np.array_equal(got_dev.cpu().numpy(), want, equal_nan=True), shape
        # in place, as the reference calls it (wrap(m_theta, m_theta), G2.cpp:98,110)
        pa = cv.api._plane(ta)
        rc = cv.lib().cvs_wrap(f._h, C.byref(pa), C.byref(pa))
        assert rc == 0
        torch.cuda.synchronize()
        assert np.array_equal(ta.cpu().numpy(), want, equal_nan=True), shape
    # range: everything in [0, 2 pi) lands in (-pi, pi]
    a = np.linspace(0, 2 * np.pi, 100001, dtype=np.float32)[:-1].reshape(100, 1000)
    w = cv.SteerableFiltersG2(None).wrap(a)
    assert w.max() <= pi32 and w.min() > -pi32
    # size mismatch is an error, like every other plane pair
    with pytest.raises(cv.CvsError):
        f = cv.SteerableFiltersG2(None)
        pa, po = cv.api._plane(np.zeros((3, 4), np.float32)), cv.api._plane(np.zeros((3, 5), np.float32))
        f._check(cv.lib().cvs_wrap(f._h, C.byref(pa), C.byref(po)), "cvs_wrap")


# ----------------------------------------------------------------------------- the reference's own test
def _recode(u8):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(u8).save(buf, format="JPEG", quality=95)
    return np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("L"))


def test_reference_gtest_basic_on_gpu(cv, ora, fish, golden_dir):
    """reference test/test.cpp:70-108 run through the HIP path, same bar (mean-L1 <= 1.0)."""
    f = cv.SteerableFiltersG2(fish, 4, 0.67)
    g2, h2, e, mag, phase = f.steer(f.getDominantOrientationAngle(), full=True)
    outs = (f.findEdges(mag, phase), f.findDarkLines(mag, phase), f.findBrightLines(mag, phase))
    fused = f.pipeline(fish)
    for name, plane, fz in zip(("edges", "linesDark", "linesBright"), outs, fused[5:]):
        gt = np.load(os.path.join(golden_dir, name + "_u8.npy")).astype(np.float64)
        for pl in (plane, fz):
            u8 = f.normalize_u8(pl)
            assert np.abs(u8.astype(np.int32) - ora.normalize_minmax_u8(pl).astype(np.int32)).max() <= 1
            err = np.abs(_recode(u8).astype(np.float64) - gt).mean()
            assert err <= 1.0, (name, err)
            assert err <= 0.05, (name, err)


def test_pipeline_equals_stepwise(cv):
    img = smooth_image(100, 140) + 0.05 * rand_image(100, 140, seed=71)
    f = cv.SteerableFiltersG2(img)
    g2, h2, e, mag, phase = f.steer(None, full=True)
    ed, dk, br = f.find(mag, phase)
    fused = cv.SteerableFiltersG2(None).pipeline(img)
    for a, b in zip((g2, h2, e, mag, phase, ed, dk, br), fused):
        assert np.array_equal(a, b)


# ----------------------------------------------------------------------------- device planes
def test_device_planes_equal_host_planes(cv):
    import torch
    img = rand_image(75, 133, seed=81)
    fh = cv.SteerableFiltersG2(img)
    host = fh.steer(0.4, full=True) + fh.steer(None, full=True)
    timg = torch.from_numpy(img).cuda()
    fd = cv.SteerableFiltersG2(timg)
    dev = fd.steer(0.4, full=True) + fd.steer(None, full=True)
    torch.cuda.synchronize()
    for a, b in zip(host, dev):
        assert b.is_cuda and np.array_equal(a, b.cpu().numpy())
    assert np.array_equal(fh.getDominantOrientationAngle(), fd.getDominantOrientationAngle().cpu().numpy())
    # strided device views (rows of a larger tensor)
    big = torch.zeros((75, 256), device="cuda")
    big[:, 3:136] = timg
    fv = cv.SteerableFiltersG2(big[:, 3:136])
    assert np.array_equal(fv.basis(2).cpu().numpy(), fh.basis(2))


# ----------------------------------------------------------------------------- errors
def test_error_behaviour(cv):
    f = cv.SteerableFiltersG2(None)
    with pytest.raises(cv.CvsError) as ei:
        f.steer(0.1)
    assert ei.value.status == -5  # CVS_E_STATE: no setup yet
    with pytest.raises(cv.CvsError) as ei:
        f.setup(np.empty((0, 5), np.float32))
    assert ei.value.status == -2  # CVS_E_SIZE: the reference would throw from inside OpenCV
    f.setup(rand_image(8, 9))
    with pytest.raises(cv.CvsError) as ei:
        f.steer(np.zeros((8, 10), np.float32))
    assert ei.value.status == -2
    f.setup(rand_image(8, 9), flags=cv.SETUP_BASIS)
    with pytest.raises(cv.CvsError) as ei:
        f.getDominantOrientationAngle()
    assert ei.value.status == -5
    g4 = cv.SteerableFiltersG4(rand_image(8, 9))
    with pytest.raises(cv.CvsError) as ei:
        g4.steer(0.2, full=True)
    assert ei.value.status == -6  # the reference has no G4 magnitude/phase
    assert g4.getDominantOrientationAngle().size == 0


# ----------------------------------------------------------------------------- full size (BASELINE configs)
def test_full_size_4096_properties(cv, ora):
    """config 2 (4096x4096): size-independent properties + oracle on bands with halo"""
    import torch
    n = 4096
    gen = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.rand((n, n), generator=gen, device="cuda")
    y = torch.rand((n, n), generator=gen, device="cuda")
    f = cv.SteerableFiltersG2(x, 4, 0.67)
    bx = [f.basis(p) for p in range(7)]
    th = f.getDominantOrientationAngle()
    assert float(th.max()) <= np.pi / 2 + 1e-6 and float(th.min()) > -np.pi / 2 - 1e-6
    # (1) oracle on horizontal bands, top / middle / bottom (borders included)
    xh = x.cpu().numpy()
    for lo, hi in ((0, 44), (1996, 2079), (n - 37, n)):
        band = ora.basis(2, xh[lo:hi], 4, 0.67, f64=True)
        # rows whose 4-row halo lies inside the band (or at the true image border)
        a = 0 if lo == 0 else 4
        b = band.shape[1] if hi == n else band.shape[1] - 4
        for p in range(7):
            assert np.abs(bx[p][lo + a:lo + b].cpu().numpy() - band[p][a:b]).max() <= TOL
    # (2) linearity: B(a*x + b*y) == a*B(x) + b*B(y)
    fy = cv.SteerableFiltersG2(y, 4, 0.67, setup_flags=cv.SETUP_BASIS)
    fz = cv.SteerableFiltersG2(0.75 * x - 0.5 * y, 4, 0.67, setup_flags=cv.SETUP_BASIS)
    for p in range(7):
        lin = 0.75 * bx[p] - 0.5 * fy.basis(p)
        assert float((fz.basis(p) - lin).abs().max()) <= TOL
    # (3) a constant image stays constant through REFLECT_101: B_p == c * sum(kx) * sum(ky)
    fc = cv.SteerableFiltersG2(torch.full((n, n), 0.5, device="cuda"), 4, 0.67, setup_flags=cv.SETUP_BASIS)
    for p in range(7):
        ix, iy = cv.basis_taps(2, p)
        want = 0.5 * float(cv.make_taps(2, ix, 4, 0.67).astype(np.float64).sum()) * float(cv.make_taps(2, iy, 4, 0.67).astype(np.float64).sum())
        bp = fc.basis(p)
        assert float((bp - want).abs().max()) <= 2e-6
    # (4) steering identity: steer(theta) is the weighted sum of the persisted bases
    g, h = f.steer(0.3)
    w = cv.steer_weights(2, 0.3)
    assert float((g - (w[0] * bx[0] + w[1] * bx[1] + w[2] * bx[2])).abs().max()) <= 1e-6
    assert float((h - (w[3] * bx[3] + w[4] * bx[4] + w[5] * bx[5] + w[6] * bx[6])).abs().max()) <= 1e-6


@pytest.mark.parametrize("kind,shape", [(2, (4096, 4096)), (4, (4096, 4096)), (2, (1531, 2049)), (4, (1080, 1920))])
def test_mirror_and_transpose_properties_full_size(cv, kind, shape):
    """oracle-independent, size-independent properties of the separable banks at BASELINE sizes (configs 2, 4, 5):
    (1) mirroring the image mirrors every basis plane, with a sign flip where the 1-D kernel along that axis is odd --
        BIT-exact: the folded sums s[+i] + s[-i] commute, the differences change sign, REFLECT_101 is symmetric;
    (2) transposing the image swaps the roles of the row and column kernels: plane (kx, ky) of the transposed image is the
        transpose of plane (ky, kx) -- g2a <-> g2c, h2a <-> h2d, h2b <-> h2c; g4a <-> g4e, g4b <-> g4d, h4a <-> h4f, h4b <-> h4e,
        h4c <-> h4d -- to rounding (the two passes run in the other order), which pins the pairing of kernels and planes of
        G4 / H4, to which the reference's golden images are blind."""
    import torch
    rows, cols = shape
    nb = 7 if kind == 2 else 11
    w, sp = (4, 0.67) if kind == 2 else (6, 0.5)
    cls = cv.SteerableFiltersG2 if kind == 2 else cv.SteerableFiltersG4
    x = torch.rand(shape, generator=torch.Generator(device="cuda").manual_seed(77 + kind), device="cuda")
    flags = cv.SETUP_BASIS
    f = cls(x, w, sp, setup_flags=flags)
    base = [f.basis(p).clone() for p in range(nb)]
    taps = [cv.make_taps(kind, i, w, sp) for i in range(7 if kind == 2 else 11)]
    odd = lambda t: bool(t[0] == -t[-1] and t[len(t) // 2] == 0.0)
    pair = [cv.basis_taps(kind, p) for p in range(nb)]
    # (1) mirrors
    fl = cls(torch.flip(x, dims=(1,)).contiguous(), w, sp, setup_flags=flags)
    fu = cls(torch.flip(x, dims=(0,)).contiguous(), w, sp, setup_flags=flags)
    for p in range(nb):
        ix, iy = pair[p]
        sx = -1.0 if odd(taps[ix]) else 1.0
        sy = -1.0 if odd(taps[iy]) else 1.0
        assert torch.equal(fl.basis(p), sx * torch.flip(base[p], dims=(1,))), (kind, p, "left-right")
        assert torch.equal(fu.basis(p), sy * torch.flip(base[p], dims=(0,))), (kind, p, "up-down")
    # (2) transpose
    ft = cls(x.t().contiguous(), w, sp, setup_flags=flags)
    for p in range(nb):
        ix, iy = pair[p]
        q = [k for k in range(nb) if np.array_equal(taps[pair[k][0]], taps[iy]) and np.array_equal(taps[pair[k][1]], taps[ix])]
        assert len(q) == 1, (kind, p, q)      # every plane has exactly one partner (itself for g2b / g4c)
        assert float((ft.basis(p) - base[q[0]].t()).abs().max()) <= TOL, (kind, p, q[0])


@pytest.mark.parametrize("kind", [2, 4])
@pytest.mark.parametrize("theta", [0.3, -1.1, 2.0])
def test_steering_under_transpose_and_mirror(cv, kind, theta):
    """the steering property itself, oracle-independent (1080 x 1920, config 4's frame): for the transposed image
    g'(theta) = [g(pi/2 - theta)]^T and h'(theta) = -[h(pi/2 - theta)]^T; for the left-right mirrored image g'(theta) =
    mirror(g(-theta)) and h'(theta) = -mirror(h(-theta)).  Both follow from the steering polynomials of G2.cpp:137-145 /
    G4.cpp:114-122 and the parities of the kernels, and both pin the pairing and the RELATIVE signs of the planes of the odd
    (H) bank, to which the reference's golden images are blind (a global sign of the whole H bank would still pass)."""
    import torch
    w, sp = (4, 0.67) if kind == 2 else (6, 0.5)
    cls = cv.SteerableFiltersG2 if kind == 2 else cv.SteerableFiltersG4
    x = torch.rand((1080, 1920), generator=torch.Generator(device="cuda").manual_seed(5 + kind), device="cuda")
    f = cls(x, w, sp, setup_flags=cv.SETUP_BASIS)
    ft = cls(x.t().contiguous(), w, sp, setup_flags=cv.SETUP_BASIS)
    fm = cls(torch.flip(x, dims=(1,)).contiguous(), w, sp, setup_flags=cv.SETUP_BASIS)
    gt, ht = ft.steer(theta)
    g1, h1 = f.steer(float(np.float32(np.pi / 2) - np.float32(theta)))
    scale = max(float(g1.abs().max()), float(h1.abs().max()), 1.0)
    assert float((gt - g1.t()).abs().max()) <= 2 * TOL * scale
    assert float((ht + h1.t()).abs().max()) <= 2 * TOL * scale
    gm, hm = fm.steer(theta)
    g2_, h2_ = f.steer(-theta)
    assert float((gm - torch.flip(g2_, dims=(1,))).abs().max()) <= 1e-6 * scale
    assert float((hm + torch.flip(h2_, dims=(1,))).abs().max()) <= 1e-6 * scale


def test_full_size_g4_4096_band(cv, ora):
    """config 5 (G4+H4 at 4096x4096): oracle on bands + constant-image property"""
    import torch
    n = 4096
    x = torch.rand((n, n), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda")
    f = cv.SteerableFiltersG4(x, 6, 0.5)
    xh = x.cpu().numpy()
    for lo, hi in ((0, 30), (3000, 3040), (n - 30, n)):
        band = ora.basis(4, xh[lo:hi], 6, 0.5, f64=True)
        a = 0 if lo == 0 else 6
        b = band.shape[1] if hi == n else band.shape[1] - 6
        for p in range(11):
            assert np.abs(f.basis(p)[lo + a:lo + b].cpu().numpy() - band[p][a:b]).max() <= TOL


def test_frame_1080p_pipeline_band(cv, ora):
    """config 4 frame shape (1080 rows x 1920 cols): full pipeline, oracle on a band"""
    img = smooth_image(1080, 1920) + 0.02 * rand_image(1080, 1920, seed=4)
    f = cv.SteerableFiltersG2(img)
    outs = f.pipeline(img)
    b = _basis_stack(f, 7)[:, 500:540]
    c = [x[500:540] for x in f.coefficients()]
    th = f.getDominantOrientationAngle()[500:540]
    og, oh, oe, om, op = ora.g2_steer_map(b, th, c)
    for got, want in zip(outs[:4], (og, oh, oe, om)):
        assert np.abs(got[500:540] - want).max() <= TOL
    oed, odk, obr = ora.find(outs[3][500:540], outs[4][500:540])
    for got, want in zip(outs[5:], (oed, odk, obr)):
        assert np.abs(got[500:540] - want).max() <= 1e-6


# ----------------------------------------------------------------------------- pyramid (config 3)
@pytest.mark.parametrize("shape", [(1, 1), (2, 3), (5, 5), (37, 51), (64, 64), (185, 256), (1080, 1920)])
def test_pyr_down_matches_oracle(cv, ora, shape):
    img = rand_image(*shape, seed=17)
    f = cv.SteerableFiltersG2(None)
    got = f.pyrDown(img)
    assert np.array_equal(got, ora.pyr_down(img))  # same op order, contraction off: bit-identical


@pytest.mark.parametrize("shape", [(1, 1), (5, 5), (13, 5), (14, 6), (37, 51), (64, 64), (185, 256), (200, 301), (1080, 1920)])
@pytest.mark.parametrize("flags", ["basis", "full"])
def test_setup_pyr_equals_setup_plus_pyr_down(cv, ora, shape, flags):
    """cvs_setup_pyr: filter this pyramid level and emit the next one in ONE pass (config 3).  The emitted level is
    bit-identical to cvs_pyr_down / the oracle, the state bit-identical to a plain setup; shapes under the fast
    path's minimum (13 x 5) and G4 handles take the two launches internally."""
    img = rand_image(*shape, seed=23)
    fl = cv.SETUP_BASIS if flags == "basis" else cv.SETUP_FULL
    want_next = ora.pyr_down(img)
    for strips in (0, 1, 10, 19, 37):
        f = cv.SteerableFiltersG2(None)
        if strips:
            f.set_strip_rows(strips)
        got = f.setup_pyr(img, flags=fl)
        assert got.shape == want_next.shape
        assert np.array_equal(got, want_next), (shape, strips)
        ref = cv.SteerableFiltersG2(None)
        if strips:
            ref.set_strip_rows(strips)
        ref.setup(img, flags=fl)
        for p_ in range(7):
            assert np.array_equal(f.basis(p_), ref.basis(p_))
        if flags == "full":
            assert np.array_equal(f.getDominantOrientationAngle(), ref.getDominantOrientationAngle())
            assert np.array_equal(f.getDominantOrientationStrength(), ref.getDominantOrientationStrength())
    f4 = cv.SteerableFiltersG4(None)
    got4 = f4.setup_pyr(img, flags=cv.SETUP_BASIS)
    assert np.array_equal(got4, want_next)
    assert np.abs(np.stack([f4.basis(p_) for p_ in range(11)]) - ora.basis(4, img, 6, 0.5, f64=True)).max() <= TOL


def test_setup_pyr_device_planes_and_chain(cv, ora):
    """device planes, a padded destination pitch, the second call on the same handle, and a chain of fused levels
    equal to pyramid() level by level"""
    import torch
    x = torch.rand((1111, 1503), generator=torch.Generator(device="cuda").manual_seed(5), device="cuda")
    f = cv.SteerableFiltersG2(None)
    want = f.pyramid(x, 5)
    hs = [cv.SteerableFiltersG2(None) for _ in range(5)]
    lv = [x]
    for k in range(4):
        lv.append(hs[k].setup_pyr(lv[k], flags=cv.SETUP_BASIS))
    hs[4].setup(lv[4], flags=cv.SETUP_BASIS)
    for a_, b_ in zip(lv, want):
        assert torch.equal(a_, b_)
    big = torch.full((556, 800), -7.0, device="cuda")
    view = big[:, 3:755]   # a padded (and 4-byte-aligned only) destination
    hs[0].setup_pyr(x, flags=cv.SETUP_BASIS, out=view)
    assert torch.equal(view, want[1])
    assert float(big[:, :3].max()) == -7.0 and float(big[:, 755:].max()) == -7.0
    ref = cv.SteerableFiltersG2(None)
    ref.setup(x, flags=cv.SETUP_BASIS)
    for p_ in (0, 4, 6):
        assert torch.equal(hs[0].basis(p_), ref.basis(p_))


@pytest.mark.parametrize("n", [4096, 7000])
def test_setup_pyr_streaming_and_per_plane_variants(cv, n):
    """the fused level on images large enough for nontemporal stores (4096^2) and for one buffer resource per state
    plane (7000^2: the 12-plane state block passes 2 GiB): next level == cvs_pyr_down, state == plain setup, bit for bit"""
    import torch
    x = torch.rand((n, n), generator=torch.Generator(device="cuda").manual_seed(n), device="cuda")
    f = cv.SteerableFiltersG2(None)
    want = f.pyrDown(x)
    g = cv.SteerableFiltersG2(None)
    got = g.setup_pyr(x, flags=cv.SETUP_FULL)
    assert torch.equal(got, want)
    ref = cv.SteerableFiltersG2(None)
    ref.setup(x, flags=cv.SETUP_FULL)
    for p_ in (0, 3, 6):
        assert torch.equal(g.basis(p_), ref.basis(p_))
    assert torch.equal(g.getDominantOrientationAngle(), ref.getDominantOrientationAngle())
    got2 = g.setup_pyr(x, flags=cv.SETUP_BASIS)       # the same image again: the taller strips of a resident image
    assert torch.equal(got2, want)


def test_pyramid_5_levels_8192(cv, ora):
    """BASELINE config 3: G2+H2 over a 5-level Gaussian pyramid of one 8192x8192 image"""
    import torch
    n = 8192
    x = torch.rand((n, n), generator=torch.Generator(device="cuda").manual_seed(3), device="cuda")
    f = cv.SteerableFiltersG2(None)
    levels = f.pyramid(x, 5)
    assert [tuple(l.shape) for l in levels] == [(8192, 8192), (4096, 4096), (2048, 2048), (1024, 1024), (512, 512)]
    # pyramid construction: bands of level 1 and the whole of level 4 against the oracle
    x0 = x.cpu().numpy()
    l1 = ora.pyr_down(x0[:132])  # rows 0..65 of level 1 need rows 0..131+2 of level 0; keep the safe part
    assert np.array_equal(levels[1][:64].cpu().numpy(), l1[:64])
    assert np.array_equal(levels[4].cpu().numpy(), ora.pyr_down(levels[3].cpu().numpy()))
    # filter bank on every level: oracle on a band (top border included)
    for lv in levels:
        f.setup(lv, flags=cv.SETUP_BASIS)
        h = lv[:48].cpu().numpy()
        band = ora.basis(2, h, 4, 0.67, f64=True)
        for p in (0, 3, 6):
            assert np.abs(f.basis(p)[:44].cpu().numpy() - band[p][:44]).max() <= TOL


@pytest.mark.parametrize("shape,levels", [((8192, 8192), 5), ((4100, 3001), 4), ((1000, 1500), 3), ((640, 480), 1)])
def test_pyramid_setup_one_call_equals_the_chain(cv, shape, levels):
    """cvs_pyramid_setup (config 3 in one call): every level image and every basis / orientation plane of every level equals,
    bit for bit, the chain of cvs_setup_pyr calls -- repeated on alternating images, so that a stale plane would show"""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(shape[0] + levels)
    imgs = [torch.rand(shape, device="cuda", generator=gen) for _ in range(2)]
    hs = [cv.SteerableFiltersG2(None) for _ in range(levels)]
    ref_h = [cv.SteerableFiltersG2(None) for _ in range(levels)]
    for rnd in range(3):
        x = imgs[rnd & 1]
        lv = cv.pyramid_setup(hs, x, flags=cv.SETUP_FULL)
        assert len(lv) == levels
        cur = x
        for l in range(levels):
            if l + 1 < levels:
                nxt = ref_h[l].setup_pyr(cur, flags=cv.SETUP_FULL)
                assert torch.equal(lv[l + 1], nxt), (rnd, l)
            else:
                ref_h[l].setup(cur, flags=cv.SETUP_FULL)
                nxt = None
            for p in range(7):
                assert torch.equal(hs[l].basis(p), ref_h[l].basis(p)), (rnd, l, p)
            assert torch.equal(hs[l].getDominantOrientationAngle(), ref_h[l].getDominantOrientationAngle()), (rnd, l)
            cur = nxt
    with pytest.raises(cv.CvsError):
        cv.pyramid_setup([hs[0], hs[0]], imgs[0])          # one handle per level


# ----------------------------------------------------------------------------- batch axis + CLI driver
def test_batch_process_frames_single_rank(cv, ora):
    """the per-frame loop of cvsteer_amd.batch with the HIP engine as the frame function (world = 1)"""
    import torch
    from cvsteer_amd import batch
    frames = torch.from_numpy(np.stack([smooth_image(96, 160) + 0.05 * rand_image(96, 160, seed=s) for s in range(3)])).cuda()
    eng = cv.SteerableFiltersG2(None)
    local, gathered = batch.run_sharded(frames, 3, (96, 160), frames.device, lambda img, outs: eng.pipeline(img, out=outs), 8)
    torch.cuda.synchronize()
    assert gathered is local and tuple(local.shape) == (3, 8, 96, 160)
    for i in range(3):
        single = cv.SteerableFiltersG2(None).pipeline(frames[i])
        for k in range(8):
            assert torch.equal(local[i, k], single[k])
    # against the oracle, decoupled at theta: steer the oracle at the GPU's theta map
    eng.setup(frames[1])
    b = np.stack([eng.basis(p).cpu().numpy() for p in range(7)])
    c = [x.cpu().numpy() for x in eng.coefficients()]
    og, oh, oe, om, op = ora.g2_steer_map(b, eng.getDominantOrientationAngle().cpu().numpy(), c)
    assert np.abs(local[1, 0].cpu().numpy() - og).max() <= TOL
    assert np.abs(local[1, 3].cpu().numpy() - om).max() <= TOL


def test_cli_driver_matches_reference_goldens(ora, golden_dir, tmp_path):
    """python -m cvsteer_amd.run on the reference's fish -> the three golden images (example/steer.cpp flow)"""
    import subprocess, sys
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lst = tmp_path / "files.txt"
    lst.write_text(os.path.join(golden_dir, "fish.jpg") + "\n")
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, "-m", "cvsteer_amd.run", "--input", str(lst), "--output", str(out), "--verbose"],
                       cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    for name, suffix in (("edges", "_edges"), ("linesDark", "_lines_dark"), ("linesBright", "_lines_bright")):
        got = np.asarray(Image.open(str(out / ("fish" + suffix + ".png")))).astype(np.float64)
        gt = np.load(os.path.join(golden_dir, name + "_u8.npy")).astype(np.float64)
        assert np.abs(_recode(got.astype(np.uint8)).astype(np.float64) - gt).mean() <= 1.0
    # --gain branch (steer.cpp:92-97): convertTo(CV_8UC1, gain)
    r = subprocess.run([sys.executable, "-m", "cvsteer_amd.run", "--input", os.path.join(golden_dir, "fish.jpg"),
                        "--output", str(out), "--gain", "2.0", "--ext", ".npy"], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    u8 = np.load(str(out / "fish_edges.npy"))
    assert u8.dtype == np.uint8 and u8.max() == 255


def test_pipeline_batch_equals_per_frame(cv):
    import torch
    frames = torch.from_numpy(np.stack([smooth_image(120, 200) + 0.1 * rand_image(120, 200, seed=s) for s in range(5)])).cuda()
    eng = cv.SteerableFiltersG2(None)
    out = eng.pipeline_batch(frames)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (5, 8, 120, 200)
    for i in range(5):
        single = cv.SteerableFiltersG2(None)
        ref = single.pipeline(frames[i])
        for k in range(8):
            assert torch.equal(out[i, k], ref[k]), (i, k)
        # every frame's state is kept and addressable
        eng.select_frame(i)
        assert torch.equal(eng.getDominantOrientationAngle(), single.getDominantOrientationAngle())
        assert torch.equal(eng.basis(4), single.basis(4))
        g, h = eng.steer(0.3)
        g1, h1 = single.steer(0.3)
        assert torch.equal(g, g1) and torch.equal(h, h1)
    with pytest.raises(cv.CvsError):
        eng.select_frame(5)
    # a later single-image setup drops back to one frame
    eng.setup(frames[0])
    with pytest.raises(cv.CvsError):
        eng.select_frame(1)


@pytest.mark.parametrize("kind", [2, 4])
def test_impulse_response_is_the_outer_product_of_the_reference_taps(cv, golden_dir, kind):
    """the HIP filter bank against a known answer that comes from the reference, not from the oracle (tests/known_answers.py):
    the basis planes of an impulse image are f32(ky * kx) of the reference's own tap tables, bit for bit -- in the interior
    in two corners (REFLECT_101 adds nothing at the first and last index), at strip borders (columns 63 / 64) -- on the strip kernel
    (96 x 200), and for two impulses at once"""
    from known_answers import impulse_planes
    rows, cols = 96, 200
    f = cv.SteerableFiltersG2(None) if kind == 2 else cv.SteerableFiltersG4(None)
    n = 7 if kind == 2 else 11
    for r, c in ((40, 100), (0, 0), (rows - 1, cols - 1), (17, 63), (50, 64), (60, 191)):
        img = np.zeros((rows, cols), np.float32)
        img[r, c] = 1.0
        f.setup(img)
        got = _basis_stack(f, n)
        want = impulse_planes(golden_dir, kind, rows, cols, r, c)
        assert np.array_equal(got, want), (kind, r, c)
    # linearity, exactly: two impulses far enough apart not to meet are the sum of two responses with no value added twice
    img = np.zeros((rows, cols), np.float32)
    img[20, 30] = 1.0
    img[70, 150] = 1.0
    f.setup(img)
    want = impulse_planes(golden_dir, kind, rows, cols, 20, 30) + impulse_planes(golden_dir, kind, rows, cols, 70, 150)
    assert np.array_equal(_basis_stack(f, n), want)


def test_pipeline_batch_dispatch_order_of_frames_does_not_show(cv, monkeypatch):
    """State-keeping batches dispatch their frames dealt from two halves of the batch (k_basis, z_ways; CVS_OPTS batch_ways is the
    tuning aid): any number of parts, batches that do not divide, the frame-table form -- every plane of every frame as in order."""
    import torch
    n = 7
    block = torch.from_numpy(np.stack([smooth_image(61, 150) + 0.1 * rand_image(61, 150, seed=40 + s) for s in range(n)])).cuda()
    loose = [block[i].clone() for i in range(n)]                       # separately allocated frames: per-frame pointer table
    monkeypatch.setenv("CVS_OPTS", "batch_ways=1")
    want = cv.SteerableFiltersG2(None).pipeline_batch(block).clone()
    for ways in ("2", "3", "4", "7", "50"):
        monkeypatch.setenv("CVS_OPTS", "batch_ways=" + ways)
        eng = cv.SteerableFiltersG2(None)
        got = eng.pipeline_batch(block)
        assert torch.equal(got, want), ways
        for i in (0, 3, 6):                                            # the state of frame i is frame i's
            eng.select_frame(i)
            single = cv.SteerableFiltersG2(None)
            single.setup(block[i])
            assert torch.equal(eng.basis(5), single.basis(5)) and torch.equal(eng.getDominantOrientationAngle(), single.getDominantOrientationAngle())
        got_l = cv.SteerableFiltersG2(None).pipeline_batch(loose)
        assert torch.equal(got_l, want), ways
    monkeypatch.delenv("CVS_OPTS")
    assert torch.equal(cv.SteerableFiltersG2(None).pipeline_batch(block), want)   # the default (two parts)


def test_pipeline_batch_fallback_paths(cv):
    # host planes (staged frame by frame) and frames too small for the fused kernel: same results
    host = np.stack([rand_image(40, 64, seed=s) for s in range(3)])
    eng = cv.SteerableFiltersG2(None)
    out = eng.pipeline_batch(host)
    for i in range(3):
        ref = cv.SteerableFiltersG2(None).pipeline(host[i])
        for k in range(8):
            assert np.array_equal(out[i, k], ref[k])
    import torch
    tiny = torch.rand((4, 9, 70), device="cuda")
    out = cv.SteerableFiltersG2(None).pipeline_batch(tiny)
    for i in range(4):
        ref = cv.SteerableFiltersG2(None).pipeline(tiny[i])
        for k in range(8):
            assert torch.equal(out[i, k], ref[k])


def test_generic_width_full_surface(cv, ora):
    """non-default taps take the two-pass generic kernels for every entry point: same results as stepwise"""
    img = smooth_image(60, 90) + 0.1 * rand_image(60, 90, seed=91)
    f = cv.SteerableFiltersG2(img, 3, 0.9)
    g0, h0, e0, m0, p0 = f.steer(None, full=True)
    ed, dk, br = f.find(m0, p0)
    fused = cv.SteerableFiltersG2(None, 3, 0.9).pipeline(img)
    for a, b in zip((g0, h0, e0, m0, p0, ed, dk, br), fused):
        assert np.array_equal(a, b)
    b = np.stack([f.basis(p) for p in range(7)])
    assert np.abs(b - ora.basis(2, img, 3, 0.9, f64=True)).max() <= TOL
    g1, h1 = cv.SteerableFiltersG2(None, 3, 0.9).setup_steer(img, 0.3)
    g2, h2 = f.steer(0.3)
    assert np.array_equal(g1, g2) and np.array_equal(h1, h2)


def test_point_steer_follows_selected_frame_and_alias_is_rejected(cv):
    import torch
    frames = torch.rand((3, 40, 70), device="cuda")
    eng = cv.SteerableFiltersG2(None)
    eng.pipeline_batch(frames)
    for i in range(3):
        eng.select_frame(i)
        single = cv.SteerableFiltersG2(frames[i])
        assert eng.steer_point((5, 7), 0.4, full=True) == single.steer_point((5, 7), 0.4, full=True)
    x = torch.rand((64, 64), device="cuda")
    with pytest.raises(cv.CvsError) as ei:
        cv.SteerableFiltersG2(None).setup_steer(x, 0.3, out=(x, torch.empty_like(x)))
    assert ei.value.status == -1


def test_planes_that_share_memory(cv):
    """include/cvsteer_hip.h, "Planes that share memory": filter-bank entries refuse an output that shares a byte with the input
    image or with another output; per-pixel entries allow an output that IS an input and nothing else; two column ranges of one
    buffer side by side (ROI views: interleaved addresses, nothing shared) are fine and give the contiguous case's values;
    host planes follow the same rules; unknown setup flags and misaligned f32 planes are refused; the handle survives it all."""
    import ctypes as C
    import torch
    from cvsteer_amd.api import _plane
    lib = cv.lib()
    f = cv.SteerableFiltersG2(None)
    h = f._h

    def steer(img, g, hq):
        return lib.cvs_setup_steer(h, C.byref(_plane(img)), cv.SETUP_BASIS, C.c_float(0.3), C.byref(_plane(g)), C.byref(_plane(hq)))

    big = torch.rand((80, 120), device="cuda")
    left, right = big[:40, 0:50], big[:40, 60:110]
    h2 = torch.empty((40, 50), device="cuda")
    assert steer(left, right, h2) == 0
    torch.cuda.synchronize()
    ref_g, ref_h = cv.SteerableFiltersG2(None).setup_steer(left.contiguous(), 0.3)
    assert torch.equal(right, ref_g) and torch.equal(h2, ref_h)
    assert steer(big[:40, 0:50], big[:40, 45:95], h2) == -1          # columns overlap by five
    assert b"overlaps the input" in lib.cvs_last_error(h)
    assert steer(big[0:40, 0:50], big[20:60, 0:50], h2) == -1        # rows overlap
    assert steer(big[0:40, 0:50], big[40:80, 0:50], h2) == 0         # rows disjoint
    g2 = torch.empty((40, 50), device="cuda")
    assert steer(left, g2, g2) == -1 and b"two output planes" in lib.cvs_last_error(h)
    assert steer(left, left, h2) == -1
    from cvsteer_amd.api import Plane
    p5, p6 = _plane(g2), _plane(g2[:, :])                            # two of the eight pipeline outputs on one plane
    outs = (C.POINTER(Plane) * 8)(*([None] * 5 + [C.pointer(p5), C.pointer(p6), None]))
    assert lib.cvs_pipeline(h, C.byref(_plane(left)), outs) == -1
    assert lib.cvs_setup(h, C.byref(_plane(left)), 0xffff) == -1 and b"unknown setup flags" in lib.cvs_last_error(h)
    crooked = _plane(g2)
    crooked.data += 2
    assert lib.cvs_setup(h, C.byref(crooked), cv.SETUP_FULL) == -1
    lvl = torch.empty((20, 25), device="cuda")
    assert lib.cvs_pyr_down(h, C.byref(_plane(g2)), C.byref(_plane(lvl))) == 0
    assert lib.cvs_pyr_down(h, C.byref(_plane(g2)), C.byref(_plane(g2[:20, :25]))) == -1
    # per-pixel entries: in place yes, shifted no -- device and host
    ang = torch.rand((40, 50), device="cuda") * 9
    want = f.wrap(ang.clone())
    assert lib.cvs_wrap(h, C.byref(_plane(ang)), C.byref(_plane(ang))) == 0
    torch.cuda.synchronize()
    assert torch.equal(ang, want)
    buf = torch.rand((41, 50), device="cuda")
    assert lib.cvs_wrap(h, C.byref(_plane(buf[0:40])), C.byref(_plane(buf[1:41]))) == -1
    hb = np.random.default_rng(3).random((41, 50), dtype=np.float32) * 9
    hwant = f.wrap(hb[0:40].copy())
    assert lib.cvs_wrap(h, C.byref(_plane(hb[0:40])), C.byref(_plane(hb[1:41]))) == -1
    assert lib.cvs_wrap(h, C.byref(_plane(hb[0:40])), C.byref(_plane(hb[0:40]))) == 0
    assert np.array_equal(hb[0:40], hwant)
    e, ph = torch.rand((40, 50), device="cuda"), torch.rand((40, 50), device="cuda")
    o = torch.empty((40, 50), device="cuda")
    assert lib.cvs_find(h, C.byref(_plane(e)), C.byref(_plane(ph)), C.byref(_plane(e)), C.byref(_plane(o)), None) == 0     # edges onto e
    assert lib.cvs_find(h, C.byref(_plane(e)), C.byref(_plane(ph)), C.byref(_plane(o)), C.byref(_plane(o)), None) == -1   # two outputs, one plane
    assert lib.cvs_mag_phase(h, C.byref(_plane(e)), C.byref(_plane(ph)), C.byref(_plane(e)), C.byref(_plane(ph))) == 0    # both in place
    # after all that the handle still works
    g, hq = f.setup_steer(left.contiguous(), 0.3)
    assert torch.equal(g, ref_g)


def test_non_finite_inputs_propagate_like_the_reference(cv, ora):
    """NaN / Inf pixels: basis planes carry them exactly where the 9x9 support touches them; phase is patched to 0"""
    img = rand_image(48, 64, seed=5)
    img[20, 30] = np.nan
    img[5, 5] = np.inf
    f = cv.SteerableFiltersG2(img)
    got = f.basis(1)
    want = ora.basis(2, img, 4, 0.67)[1]
    assert np.array_equal(np.isfinite(got), np.isfinite(want))
    ok = np.isfinite(want)
    assert np.abs(got[ok] - want[ok]).max() <= TOL
    g2, h2, e, m, p = f.steer(0.3, full=True)
    assert np.isfinite(p).all()          # patchNaNs (G2.cpp:111)
    assert not np.isfinite(m[20, 30])
    # every plane of both banks, on the strip kernels and on the two-pass fallback (images too small for the strips, and a
    # non-default width): NaN and Inf sit exactly where the reference's filters put them -- the row filter multiplies the
    # zero centre tap of an odd kernel in (0 * Inf = NaN), its folded column filter never touches the centre row.
    # (tools/fuzz_campaign.py found the fallback's column pass summing all taps: a wider footprint for odd column kernels.)
    rng = np.random.default_rng(99)
    for kind, w, s_, rows, cols in ((2, 4, 0.67, 48, 64), (4, 6, 0.5, 48, 64), (2, 4, 0.67, 12, 40), (2, 4, 0.67, 30, 4), (4, 6, 0.5, 14, 127),
                                    (4, 6, 0.5, 5, 90), (2, 4, 0.67, 6, 21), (2, 6, 0.5, 40, 50), (4, 8, 0.4, 60, 70), (2, 4, 0.67, 1, 1)):
        img = rng.random((rows, cols), dtype=np.float32)
        for v in (np.nan, np.inf, -np.inf):
            img[rng.integers(0, rows), rng.integers(0, cols)] = v
        f = cv.SteerableFiltersG2(img, w, s_) if kind == 2 else cv.SteerableFiltersG4(img, w, s_)
        n = 7 if kind == 2 else 11
        got = _basis_stack(f, n)
        want = ora.basis(kind, img, w, s_)
        assert np.array_equal(np.isnan(got), np.isnan(want)), (kind, w, rows, cols)
        assert np.array_equal(np.isposinf(got), np.isposinf(want)) and np.array_equal(np.isneginf(got), np.isneginf(want)), (kind, w, rows, cols)
        ok = np.isfinite(want)
        assert np.abs(got[ok] - want[ok]).max(initial=0.0) <= TOL, (kind, w, rows, cols)


def test_stateless_pipeline_outputs_only(cv):
    """CVS_OPT_PERSIST_STATE = 0: same requested outputs, no state planes afterwards"""
    import torch
    frames = torch.from_numpy(np.stack([smooth_image(90, 150) + 0.1 * rand_image(90, 150, seed=s) for s in range(4)])).cuda()
    ref = cv.SteerableFiltersG2(None).pipeline_batch(frames)
    eng = cv.SteerableFiltersG2(None)
    eng.set_persist(False)
    got = eng.pipeline_batch(frames, outputs=(5, 6, 7))
    torch.cuda.synchronize()
    assert tuple(got.shape) == (4, 3, 90, 150)
    assert torch.equal(got, ref[:, 5:8])
    with pytest.raises(cv.CvsError) as ei:
        eng.getDominantOrientationAngle()
    assert ei.value.status == -5
    single = eng.pipeline(frames[2], out=[None] * 5 + [torch.empty_like(frames[2]) for _ in range(3)])
    assert single[0] is None and torch.equal(single[7], ref[2, 7])
    eng.set_persist(True)
    eng.pipeline(frames[1])
    assert torch.equal(eng.getDominantOrientationAngle(), cv.SteerableFiltersG2(frames[1]).getDominantOrientationAngle())


def test_8bit_input_images(cv, ora, fish, golden_dir):
    """CVS_DEPTH_U8: 8-bit images are widened on the device, unscaled, like cv::Mat1f(const Mat&) (test.cpp:85)"""
    import torch
    u8 = np.load(os.path.join(golden_dir, "fish_u8.npy"))
    ref = cv.SteerableFiltersG2(fish).pipeline(fish)
    for img in (u8, torch.from_numpy(u8).cuda(), u8[3:150, 10:200]):
        f = cv.SteerableFiltersG2(None)
        got = f.pipeline(img)
        want = ref if img.shape == u8.shape else cv.SteerableFiltersG2(None).pipeline(np.ascontiguousarray(fish[3:150, 10:200]))
        for a, b in zip(got, want):
            a = a.cpu().numpy() if hasattr(a, "cpu") else a
            assert a.dtype == np.float32 and np.array_equal(a, b)
    f = cv.SteerableFiltersG2(u8)                    # constructor path, as test.cpp:85 does
    assert np.array_equal(f.basis(0), cv.SteerableFiltersG2(fish).basis(0))
    g1, h1 = cv.SteerableFiltersG4(u8).steer(0.3)
    g2, h2 = cv.SteerableFiltersG4(fish).steer(0.3)
    assert np.array_equal(g1, g2) and np.array_equal(h1, h2)
    out = cv.SteerableFiltersG2(None).pipeline_batch(np.stack([u8, u8[::-1].copy()]))
    assert np.array_equal(out[0, 5], ref[5])
    with pytest.raises(cv.CvsError):                 # 8-bit planes are inputs only
        f.computeMagnitudeAndPhase(u8, u8)


@pytest.mark.parametrize("shape", [(185, 256), (1080, 1920), (131, 1021), (64, 67)])
def test_8bit_images_read_in_kernel_equal_widened(cv, shape):
    """8-bit images are read by the strip kernels themselves (buffer_load_ubyte + cvt, BasisArgs::in_u8; test/test.cpp:73,85 hands
    an 8-bit Mat to the constructor): every entry point gives bit-identical results to the same image widened to f32 first --
    dense planes, odd widths, ROI views (step > cols, odd byte offsets), device and host, G2 and G4, single image and batch"""
    import torch
    rows, cols = shape
    gen = torch.Generator(device="cuda").manual_seed(rows * 31 + cols)
    big = (torch.rand((rows + 6, cols + 13), device="cuda", generator=gen) * 256).to(torch.uint8)
    views = [big[:rows, :cols].contiguous(), big[3:3 + rows, 5:5 + cols], big[1:1 + rows, :cols]]
    for u8 in views:
        f32 = u8.to(torch.float32).contiguous()
        a, b = cv.SteerableFiltersG2(None), cv.SteerableFiltersG2(None)
        a.setup(u8, flags=cv.SETUP_FULL)
        b.setup(f32, flags=cv.SETUP_FULL)
        for p in range(7):
            assert torch.equal(a._state(p), b._state(p)), ("basis", p)
        assert torch.equal(a.getDominantOrientationAngle(), b.getDominantOrientationAngle())
        ga, ha = a.setup_steer(u8, 0.3)
        gb, hb = b.setup_steer(f32, 0.3)
        assert torch.equal(ga, gb) and torch.equal(ha, hb)
        for x, y in zip(a.pipeline(u8), b.pipeline(f32)):
            assert torch.equal(x, y)
        a4, b4 = cv.SteerableFiltersG4(None), cv.SteerableFiltersG4(None)
        g4a, h4a = a4.setup_steer(u8, -0.7)
        g4b, h4b = b4.setup_steer(f32, -0.7)
        assert torch.equal(g4a, g4b) and torch.equal(h4a, h4b)
        for p in range(11):
            assert torch.equal(a4._state(p), b4._state(p)), ("g4 basis", p)
    # host bytes (pageable numpy, padded rows): the bytes are uploaded and read as bytes
    hu8 = big.cpu().numpy()[3:3 + rows, 5:5 + cols]
    assert hu8.strides[0] > cols
    hw = cv.SteerableFiltersG2(None).pipeline(hu8)
    hr = cv.SteerableFiltersG2(None).pipeline(np.ascontiguousarray(hu8).astype(np.float32))
    for x, y in zip(hw, hr):
        assert np.array_equal(x, y)
    # a block of frames: one launch over grid.z reading bytes; state kept and outputs only
    n = 5
    blk = (torch.rand((n, rows, cols), device="cuda", generator=gen) * 256).to(torch.uint8)
    for persist in (True, False):
        e8, e32 = cv.SteerableFiltersG2(None), cv.SteerableFiltersG2(None)
        e8.set_persist(persist)
        e32.set_persist(persist)
        assert torch.equal(e8.pipeline_batch(blk), e32.pipeline_batch(blk.to(torch.float32)))
        assert torch.equal(e8.pipeline_batch(blk, outputs=(5, 6, 7)), e32.pipeline_batch(blk.to(torch.float32), outputs=(5, 6, 7)))
    if True:
        e8.set_persist(True)
        e8.pipeline_batch(blk)
        e8.select_frame(3)
        single = cv.SteerableFiltersG2(blk[3].to(torch.float32))
        assert torch.equal(e8._state(2), single._state(2))


def test_random_shapes_fuzz(cv, ora):
    """seeded shape fuzz across the fast / generic path boundaries (rows around 3W+1, cols around W+1 and
    around multiples of 64, strip boundaries): basis planes and the fused pipeline vs the oracle"""
    rng = np.random.default_rng(2024)
    shapes = [(13, 5), (12, 70), (14, 64), (19, 65), (27, 63), (28, 128), (20, 129), (38, 191), (57, 257)]
    shapes += [(int(rng.integers(1, 140)), int(rng.integers(1, 330))) for _ in range(16)]
    for rows, cols in shapes:
        img = rng.random((rows, cols), dtype=np.float32)
        f2 = cv.SteerableFiltersG2(img)
        got = np.stack([f2.basis(p) for p in range(7)])
        assert np.abs(got - ora.basis(2, img, 4, 0.67, f64=True)).max() <= TOL, (rows, cols)
        f4 = cv.SteerableFiltersG4(img)
        got4 = np.stack([f4.basis(p) for p in range(11)])
        assert np.abs(got4 - ora.basis(4, img, 6, 0.5, f64=True)).max() <= TOL, (rows, cols)
        # fused pipeline == stepwise on the same handle state
        outs = cv.SteerableFiltersG2(None).pipeline(img)
        step = f2.steer(None, full=True)
        for a, b in zip(outs[:5], step):
            assert np.array_equal(a, b), (rows, cols)
        g4, h4 = cv.SteerableFiltersG4(None).setup_steer(img, -0.7)
        og, oh = ora.g4_steer_scalar(got4, -0.7)
        assert np.abs(g4 - og).max() <= 1e-6 and np.abs(h4 - oh).max() <= 1e-6, (rows, cols)


def test_g4_extension_orientation_and_full_steer(cv, ora):
    """opt-in extension beyond the reference: G4 C1..C3 / theta / strength, steer(full), mag/phase"""
    img = smooth_image(70, 110) + 0.05 * rand_image(70, 110, seed=77)
    f = cv.SteerableFiltersG4(img, extensions=True)
    b = np.stack([f.basis(p) for p in range(11)])
    o1, o2, o3, oth, ost = ora.g4_orientation(b)
    c1, c2, c3 = f.coefficients()
    scale = max(1.0, np.abs(o1).max())
    for got, want in ((c1, o1), (c2, o2), (c3, o3), (f.getDominantOrientationStrength(), ost)):
        assert np.abs(got - want).max() <= 1e-6 * scale
    ok = ost > 1e-3 * scale
    assert angle_diff(f.getDominantOrientationAngle(), oth, np.pi)[ok].max() <= TOL
    g, h, e, m, p = f.steer(0.3, full=True)
    og, oh = ora.g4_steer_scalar(b, 0.3)
    assert np.abs(g - og).max() <= 1e-6 and np.abs(h - oh).max() <= 1e-6
    want_e = o1 + np.float32(np.cos(0.6)) * o2 + np.float32(np.sin(0.6)) * o3
    assert np.abs(e - want_e).max() <= 1e-5 * scale
    om, op = ora.mag_phase(og, oh)
    assert np.abs(m - om).max() <= 1e-5 and angle_diff(p, op, 2 * np.pi)[om > 1e-3].max() <= 2e-5
    g2, h2, e2, m2, p2 = f.steer(None, full=True)          # at the G4 dominant orientation
    g3, h3 = f.steer(f.getDominantOrientationAngle())
    assert np.array_equal(g2, g3) and np.array_equal(h2, h3)
    mm, pp = f.computeMagnitudeAndPhase(g2, h2)
    assert np.array_equal(mm, m2) and np.array_equal(pp, p2)
    # default objects stay exactly like the reference
    ref = cv.SteerableFiltersG4(img)
    assert ref.getDominantOrientationAngle().size == 0
    with pytest.raises(cv.CvsError):
        ref.steer(0.3, full=True)


def test_find_on_energy_option(cv):
    """CVS_OPT_FIND_ON = 1: the fused pipeline weights the oriented energy e instead of the magnitude
    (the API's findEdges(e, phase) semantics; the reference's callers happen to pass the magnitude)"""
    from cvsteer_amd import _lib as L
    img = smooth_image(80, 120) + 0.1 * rand_image(80, 120, seed=15)
    f = cv.SteerableFiltersG2(None)
    f.set_option(L.OPT_FIND_ON, 1)
    outs = f.pipeline(img)
    ed, dk, br = f.find(outs[2], outs[4])
    assert np.array_equal(outs[5], ed) and np.array_equal(outs[6], dk) and np.array_equal(outs[7], br)
    f.set_option(L.OPT_FIND_ON, 0)
    outs0 = f.pipeline(img)
    ed0, _, _ = f.find(outs0[3], outs0[4])
    assert np.array_equal(outs0[5], ed0) and not np.array_equal(outs0[5], outs[5])


def test_large_ragged_image_bands(cv, ora):
    """a big image whose width is no multiple of 64 and whose height is no multiple of the strip: bands vs oracle"""
    import torch
    rows, cols = 3001, 5003
    x = torch.rand((rows, cols), generator=torch.Generator(device="cuda").manual_seed(7), device="cuda")
    xh = x.cpu().numpy()
    for cls, kind, w, s, n, halo in ((cv.SteerableFiltersG2, 2, 4, 0.67, 7, 4), (cv.SteerableFiltersG4, 4, 6, 0.5, 11, 6)):
        f = cls(x, w, s)
        for lo, hi in ((0, 40), (1490, 1535), (rows - 41, rows)):
            band = ora.basis(kind, xh[lo:hi], w, s, f64=True)
            a = 0 if lo == 0 else halo
            b = band.shape[1] if hi == rows else band.shape[1] - halo
            for p in (0, n // 2, n - 1):
                got = f.basis(p)[lo + a:lo + b].cpu().numpy()
                assert np.abs(got - band[p][a:b]).max() <= TOL, (kind, lo, p)
                assert np.abs(got[:, -70:] - band[p][a:b, -70:]).max() <= TOL   # right border columns


def test_g4_8192_band(cv, ora):
    """G4+H4 at 8192x8192 (state block 16 x 256 MiB = 4 GiB: one buffer resource per plane instead of one per block)"""
    import torch
    n = 8192
    x = torch.rand((n, n), generator=torch.Generator(device="cuda").manual_seed(11), device="cuda")
    f = cv.SteerableFiltersG4(x)
    lo, hi = 4000, 4040
    band = ora.basis(4, x[lo:hi].cpu().numpy(), 6, 0.5, f64=True)
    for p in (0, 5, 10):
        assert np.abs(f.basis(p)[lo + 6:hi - 6].cpu().numpy() - band[p][6:-6]).max() <= TOL
    g, h = f.steer(0.3)
    w = cv.steer_weights(4, 0.3)
    acc = sum(float(w[p]) * f.basis(p)[lo:hi] for p in range(5))
    assert float((g[lo:hi] - acc).abs().max()) <= 2e-6


def test_block_order_never_changes_results(cv):
    """CVS_OPT_BLOCK_ORDER (row-major / every XCD on its own column range / dynamic tail / the engine's choice): identical outputs
    for every variant, ragged widths and heights too"""
    import torch
    from cvsteer_amd import _lib as L
    for shape in ((1100, 1500), (257, 449), (300, 2048 + 64), (1030, 64)):
        img = torch.rand(shape, device="cuda")
        ref = None
        for order in (L.ORDER_PLAIN, L.ORDER_XCD_COLUMNS, L.ORDER_DYNAMIC_TAIL, -1):
            f = cv.SteerableFiltersG2(None)
            f.set_option(L.OPT_BLOCK_ORDER, order)
            outs = []
            for _ in range(3):                              # the online tuner tries its candidates on these calls
                outs = f.pipeline(img)
            theta = f.getDominantOrientationAngle().clone()
            g, h = f.setup_steer(img, 0.3)
            cur = [o.clone() for o in outs] + [g, h] + [f.basis(p) for p in range(7)] + [theta]
            f.set_persist(False)
            feat = [torch.empty_like(img) for _ in range(3)]
            f.pipeline(img, out=[None] * 5 + feat)
            cur += feat
            frames = torch.stack([img, img.flip(0)])
            cur += [o.clone() for o in f.pipeline_batch(frames, outputs=(5, 6, 7))]
            if ref is None:
                ref = cur
            for a_, b_ in zip(cur, ref):
                assert torch.equal(a_, b_), (shape, order)
    for bad in (-2, 1, 7, 1000001):    # the weighted and band-group orders of rounds 2-4 are gone
        with pytest.raises(cv.CvsError):
            cv.SteerableFiltersG2(None).set_option(L.OPT_BLOCK_ORDER, bad)


def test_workgroups_per_cu_cap_never_changes_results(cv, monkeypatch):
    """the occupancy cap (BasisArgs::wg_per_cu: dynamic LDS no kernel touches, on by rule for single G2 images of 2 Mpix and more, any value
    through CVS_OPTS wgcap=0..8): every value launches -- one or two workgroups per CU would ask for more LDS than a workgroup may have and
    are clamped -- and every G2 variant returns the bits of the uncapped launch; a cap forced through CVS_OPTS is what launch_info reports"""
    import torch
    from cvsteer_amd import _lib as L
    img = torch.rand((1536, 2048), device="cuda", generator=torch.Generator(device="cuda").manual_seed(21))
    frames = torch.rand((4, 600, 1000), device="cuda", generator=torch.Generator(device="cuda").manual_seed(22))
    ref = None
    for cap in (0, 1, 2, 3, 5, 8, None):
        if cap is None:
            monkeypatch.delenv("CVS_OPTS", raising=False)    # the engine's own rule
        else:
            monkeypatch.setenv("CVS_OPTS", "wgcap=%d" % cap)
        f = cv.SteerableFiltersG2(None)
        f.set_option(L.OPT_AUTOTUNE, 0)
        cur = []
        f.setup(img, flags=cv.SETUP_BASIS)
        if cap is not None:
            assert f.launch_info()["wg_per_cu"] == cap
        cur += [f.basis(p).clone() for p in range(7)]
        g, h = f.setup_steer(img, 0.3, flags=cv.SETUP_FULL)
        cur += [g, h, f.getDominantOrientationAngle().clone(), f.getDominantOrientationStrength().clone()]
        cur += [o.clone() for o in f.pipeline(img)]
        cur += [f.setup_pyr(img, flags=cv.SETUP_BASIS), f.basis(6).clone()]
        cur += [f.pipeline_batch(frames).clone()]
        f.set_persist(False)
        cur += [f.pipeline_batch(frames, outputs=(5, 6, 7)).clone()]
        torch.cuda.synchronize()
        if ref is None:
            ref = cur
        for k, (a_, b_) in enumerate(zip(cur, ref)):
            assert torch.equal(a_, b_), (cap, k)


def test_a_recycled_tile_queue_slot_is_reset_before_its_first_dynamic_launch(cv):
    """the per-XCD tile queues of the dynamic launch order live in a slot that travels with the state block; a slot given back with a freed
    block and taken by a new one holds an exhausted set.  The new owner's first dynamic launch resets it with a kernel on its own stream
    (not a memset: memory-side atomics would not see it) -- every tail tile is written"""
    import torch
    from cvsteer_amd import _lib as L
    img = torch.rand((1100, 1500), device="cuda", generator=torch.Generator(device="cuda").manual_seed(23))
    ref = cv.SteerableFiltersG2(None)
    ref.set_option(L.OPT_BLOCK_ORDER, L.ORDER_PLAIN)
    want = [o.clone() for o in ref.pipeline(img)] + [ref.basis(p).clone() for p in range(7)]
    for rnd in range(4):
        f = cv.SteerableFiltersG2(None)
        f.set_option(L.OPT_BLOCK_ORDER, L.ORDER_DYNAMIC_TAIL)
        for _ in range(1 + (rnd & 1)):      # an odd and an even number of launches: either set is the exhausted one
            outs = f.pipeline(img)
        got = [o.clone() for o in outs] + [f.basis(p).clone() for p in range(7)]
        for k, (a_, b_) in enumerate(zip(got, want)):
            assert torch.equal(a_, b_), (rnd, k)
        del f, outs
        torch.cuda.synchronize()
        cv.lib().cvs_release_cached_memory() # the parked block (and its queue slot) really goes back; the next handle recycles the slot


def test_literal_tap_instances_equal_the_argument_tap_instances(cv, ora, monkeypatch):
    """a handle made with the reference's defaults (width 4, spacing 0.67f) runs the caller-pipeline variants as instances with the taps compiled
    in as literal operands (cvs_launch_info.literal_taps = 1; k_basis_lit); CVS_OPTS lit=0 -- and any handle with other taps -- runs the
    instances that take them from the kernel arguments.  Same operations on the same values: every output and state plane bit for bit, single
    images (state kept, outputs only with several masks, streaming and plain stores) and frame batches (regular, with and without state)."""
    import torch
    from cvsteer_amd import _lib as L
    gen = torch.Generator(device="cuda").manual_seed(41)
    for shape, nt in (((1100, 1500), 1), ((257, 449), 0), ((1536, 2048), 1)):
        img = torch.rand(shape, device="cuda", generator=gen)
        frames = torch.rand((3,) + shape, device="cuda", generator=gen)
        got = {}
        for lit in (1, 0):
            monkeypatch.setenv("CVS_OPTS", "nt_stores=%d,lit=%d" % (nt, lit))
            f = cv.SteerableFiltersG2(None)
            f.set_option(L.OPT_AUTOTUNE, 0)
            cur = [o.clone() for o in f.pipeline(img)]
            assert f.launch_info()["literal_taps"] == lit, (shape, lit, f.launch_info())
            cur += [f.basis(p).clone() for p in range(7)] + [f.getDominantOrientationAngle().clone(), f.getDominantOrientationStrength().clone()]
            cur += [f.pipeline_batch(frames).clone()]
            assert f.launch_info()["literal_taps"] == lit
            f.select_frame(2)
            cur += [f.basis(3).clone(), f.coefficients()[0].clone()]
            f.set_persist(False)
            for mask in ((5, 6, 7), (0, 1), (2, 3, 4), (4,)):
                o = [torch.empty_like(img) if k in mask else None for k in range(8)]
                f.pipeline(img, out=o)
                assert f.launch_info()["literal_taps"] == lit
                cur += [o[k] for k in mask]
                cur += [f.pipeline_batch(frames, outputs=mask).clone()]
            f.setup(img, flags=cv.SETUP_FULL)
            assert f.launch_info()["literal_taps"] == 0     # only the pipeline variants have such instances
            got[lit] = cur
        for k, (a_, b_) in enumerate(zip(got[1], got[0])):
            assert torch.equal(a_, b_), (shape, k)
    monkeypatch.delenv("CVS_OPTS")
    # other taps (same width, another spacing): the argument instances, whatever lit says -- and right against the oracle
    img = torch.rand((300, 420), device="cuda", generator=gen)
    f = cv.SteerableFiltersG2(None, 4, 0.5)
    outs = f.pipeline(img)
    assert f.launch_info()["literal_taps"] == 0
    truth = ora.basis(ora.KIND_G2, img.cpu().numpy(), 4, 0.5, f64=True)
    for p in range(7):
        assert np.abs(f.basis(p).cpu().numpy() - truth[p]).max() <= 1e-5, p
    assert all(torch.isfinite(o).all() for o in outs)


def test_xcd_column_order_and_g4(cv):
    """block order 1000000 (every XCD on its own range of column blocks) on widths whose 256-column blocks divide among the 8
    XCDs and on widths where they do not, short and tall images, G2 and G4: identical outputs.  (G4 with this order pinned used
    to take another grid and leave tiles unwritten: tools/fuzz_campaign.py, seed 43.)"""
    import torch
    from cvsteer_amd import _lib as L
    for shape in ((300, 2048), (61, 4096), (1200, 2048 - 70), (95, 449), (220, 309), (29, 255)):
        img = torch.rand(shape, device="cuda")
        for cls, n in ((cv.SteerableFiltersG2, 7), (cv.SteerableFiltersG4, 11)):
            ref = None
            for order in (L.ORDER_PLAIN, L.ORDER_XCD_COLUMNS, L.ORDER_DYNAMIC_TAIL):
                f = cls(None)
                f.set_option(L.OPT_BLOCK_ORDER, order)
                for sr in (0, 10):
                    if sr:
                        f.set_strip_rows(sr)
                    g, h = f.setup_steer(img, 0.3)
                    cur = [g.clone(), h.clone()] + [f.basis(p).clone() for p in range(n)]
                    if cls is cv.SteerableFiltersG2:
                        cur += [o.clone() for o in f.pipeline(img)]
                    if ref is None:
                        ref = cur
                    for a_, b_ in zip(cur, ref):
                        assert torch.equal(a_, b_), (shape, cls.__name__, order, sr)


def test_planes_beyond_2gib_row_banded(cv):
    """Planes of 2 GiB and more (here 30000 x 40000 f32 = 4.8 GB, far end beyond 4 GiB) run through the same
    kernel in row bands.  Size-independent check: any window of rows, cropped out with enough halo and filtered on
    its own, gives bit-identical values -- at the image top/bottom (reflection), across the band seams, anywhere."""
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 120 * 2**30:
        pytest.skip("needs > 100 GB of device memory")
    rows, cols, W = 30000, 40000, 4
    gen = torch.Generator(device="cuda").manual_seed(5)
    img = torch.rand((rows, cols), device="cuda", generator=gen)
    f = cv.SteerableFiltersG2(None)
    g, h = f.setup_steer(img, 0.3)
    band = (0x7ffffff0 // (cols * 4) - 2 * W)          # rows one launch can address (before strip rounding)
    seams = [band // 19 * 19, 2 * (band // 19 * 19)]
    small = cv.SteerableFiltersG2(None)

    def check(lo, hi, keep_lo, keep_hi):
        gs, hs = small.setup_steer(img[lo:hi], 0.3)
        assert torch.equal(gs[keep_lo - lo:keep_hi - lo], g[keep_lo:keep_hi]), (lo, hi)
        assert torch.equal(hs[keep_lo - lo:keep_hi - lo], h[keep_lo:keep_hi]), (lo, hi)
        for p in (0, 3, 6):
            assert torch.equal(small.basis(p)[keep_lo - lo:keep_hi - lo], f.basis(p)[keep_lo:keep_hi]), (lo, hi, p)

    check(0, 120, 0, 120 - W)                                   # top border
    check(rows - 120, rows, rows - 120 + W, rows)               # bottom border (beyond 4 GiB)
    for sm in seams:
        assert 0 < sm < rows
        check(sm - 60, sm + 60, sm - 60 + W, sm + 60 - W)       # band seams
    check(17000, 17100, 17000 + W, 17100 - W)
    del g, h, f
    # the callers' pipeline without persisted state, and the G4 pair kernel, on a 2.3 GB plane (two bands)
    rows2 = 14400
    sub = img[:rows2]
    f = cv.SteerableFiltersG2(None)
    f.set_persist(False)
    feat = [torch.empty_like(sub) for _ in range(3)]
    f.pipeline(sub, out=[None] * 5 + feat)
    seam = (0x7ffffff0 // (cols * 4) - 2 * W) // 19 * 19
    lo, hi = seam - 50, seam + 50
    feat_s = [torch.empty_like(sub[lo:hi]) for _ in range(3)]
    small.set_persist(False)
    small.pipeline(sub[lo:hi], out=[None] * 5 + feat_s)
    for a, b in zip(feat, feat_s):
        assert torch.equal(a[lo + W:hi - W], b[W:-W])
    del feat, f
    f4 = cv.SteerableFiltersG4(None)
    g4, h4 = f4.setup_steer(sub, -0.7)
    s4 = cv.SteerableFiltersG4(None)
    seam4 = (0x7ffffff0 // (cols * 4) - 12) // 27 * 27
    lo, hi = seam4 - 60, seam4 + 60
    g4s, h4s = s4.setup_steer(sub[lo:hi], -0.7)
    assert torch.equal(g4[lo + 6:hi - 6], g4s[6:-6]) and torch.equal(h4[lo + 6:hi - 6], h4s[6:-6])
    g4s, h4s = s4.setup_steer(sub[rows2 - 100:], -0.7)
    assert torch.equal(g4[rows2 - 94:], g4s[6:]) and torch.equal(h4[rows2 - 94:], h4s[6:])


def test_narrow_view_of_a_2gib_plane_small_state(cv, ora):
    """A narrow column view of a plane of 2 GiB and more: the state block is small (single-resource form
    eligible) but the caller's input / output pitch forces row bands.  The banded launches must honour the band
    (the single-resource form ignored row_lo / row_base and rewrote rows at the seam).  Checked
    against the oracle at the top, at the band seam and at the bottom rows, for G2 (+ steer into a strided
    output) and G4."""
    import torch
    rows, big_cols, cols, W = 16400, 32768, 1024, 4        # 16400 x 32768 f32 = 2.15 GB: two bands
    gen = torch.Generator(device="cuda").manual_seed(11)
    big = torch.rand((rows, big_cols), device="cuda", generator=gen)
    out_big = torch.empty((2, rows, big_cols), device="cuda")
    view = big[:, 4096:4096 + cols]
    f = cv.SteerableFiltersG2(None)
    g, h = f.setup_steer(view, 0.3, out=(out_big[0][:, :cols], out_big[1][:, 64:64 + cols]))
    whole = cv.SteerableFiltersG2(None)                    # same pixels from a compact copy: one launch, no bands
    gw, hw = whole.setup_steer(view.contiguous(), 0.3)
    assert torch.equal(g, gw) and torch.equal(h, hw)
    for p in range(7):
        assert torch.equal(f.basis(p), whole.basis(p)), p
    band = 0x7ffffff0 // (big_cols * 4) - 2 * W
    seam = band // 19 * 19
    assert 0 < seam < rows
    host = view.cpu().numpy()
    for lo, hi in ((0, 64), (seam - 40, seam + 40), (rows - 64, rows)):
        lo_h, hi_h = max(0, lo - 2 * W), min(rows, hi + 2 * W)  # halo rows so the crop filters like the whole image
        truth = ora.basis(2, np.ascontiguousarray(host[lo_h:hi_h]), 4, 0.67, f64=True)
        keep_lo = 0 if lo_h == 0 else W
        keep_hi = (hi_h - lo_h) if hi_h == rows else (hi_h - lo_h) - W
        for p in (0, 3, 6):
            got = f.basis(p)[lo_h + keep_lo:lo_h + keep_hi].cpu().numpy()
            assert np.abs(got - truth[p][keep_lo:keep_hi]).max() <= TOL, (lo, hi, p)
    f4 = cv.SteerableFiltersG4(None)
    g4, h4 = f4.setup_steer(view, -0.7)
    w4 = cv.SteerableFiltersG4(None)
    g4w, h4w = w4.setup_steer(view.contiguous(), -0.7)
    assert torch.equal(g4, g4w) and torch.equal(h4, h4w)
    for p in (0, 5, 10):
        assert torch.equal(f4.basis(p), w4.basis(p)), p
    # ... and a plane tall enough that the image's END lies beyond 4 GiB from the upper bands' shifted plane pointers (the
    # G4 kernels keep running 32-bit row offsets: the reflection limit must saturate there, not wrap)
    del big, out_big, view
    rows2 = 40000                                          # 40000 x 32768 f32 = 5.2 GB, three bands
    big2 = torch.rand((rows2, big_cols), device="cuda", generator=gen)
    view2 = big2[:, 8192:8192 + 512]
    g4, h4 = f4.setup_steer(view2, 0.4)
    g4w, h4w = w4.setup_steer(view2.contiguous(), 0.4)
    assert torch.equal(g4, g4w) and torch.equal(h4, h4w)
    for p in (0, 5, 10):
        assert torch.equal(f4.basis(p), w4.basis(p)), p


def test_overlapped_host_path_matches_device_path(cv):
    """host planes of 1 Mpix and more travel in row bands, upload / filtering / download
    overlapped (two copy streams + a download thread).  Results must be bit-identical to the same call on device
    planes and to the non-overlapped host path, for f32 and 8-bit host images, fused steer and the whole pipeline."""
    import torch
    from cvsteer_amd import _lib as L
    rng = np.random.default_rng(31)
    img = rng.random((1531, 1100), dtype=np.float32)          # odd sizes: ragged last band, ragged last strip
    dev = torch.from_numpy(img).cuda()
    ref = cv.SteerableFiltersG2(None)
    g_d, h_d = ref.setup_steer(dev, 0.3, flags=cv.SETUP_FULL)
    want_basis = [ref.basis(p).cpu().numpy() for p in range(7)]
    want_theta = ref.getDominantOrientationAngle().cpu().numpy()
    for overlap in (1, 0):
        f = cv.SteerableFiltersG2(None)
        f.set_option(L.OPT_HOST_OVERLAP, overlap)
        g_h, h_h = f.setup_steer(img, 0.3, flags=cv.SETUP_FULL)       # numpy in, numpy out
        assert isinstance(g_h, np.ndarray)
        assert np.array_equal(g_h, g_d.cpu().numpy()) and np.array_equal(h_h, h_d.cpu().numpy()), overlap
        for p in range(7):
            assert np.array_equal(f.basis(p), want_basis[p]), (overlap, p)
        assert np.array_equal(f.getDominantOrientationAngle(), want_theta, equal_nan=True)
        # host image, device outputs / device image, host outputs
        g_m, h_m = torch.empty_like(dev), torch.empty_like(dev)
        f.setup_steer(img, 0.3, out=(g_m, h_m))
        assert torch.equal(g_m, g_d) and torch.equal(h_m, h_d)
        g_n, h_n = np.empty_like(img), np.empty_like(img)
        f.setup_steer(dev, 0.3, out=(g_n, h_n))
        assert np.array_equal(g_n, g_d.cpu().numpy()) and np.array_equal(h_n, h_d.cpu().numpy())
        # strided host planes (a column window of a wider array), second call on the same handle
        wide_in = rng.random((1531, 1500), dtype=np.float32)
        wide_out = np.zeros((2, 1531, 1400), np.float32)
        f.setup_steer(wide_in[:, 100:1200], -0.9, out=(wide_out[0][:, 7:1107], wide_out[1][:, 300:1400]))
        gw, hw = ref.setup_steer(torch.from_numpy(np.ascontiguousarray(wide_in[:, 100:1200])).cuda(), -0.9)
        assert np.array_equal(wide_out[0][:, 7:1107], gw.cpu().numpy()) and np.array_equal(wide_out[1][:, 300:1400], hw.cpu().numpy())
        assert not wide_out[0][:, :7].any() and not wide_out[0][:, 1107:].any()          # nothing written beside the planes
        # the callers' whole sequence with host planes in and out
        outs_h = f.pipeline(img)
        outs_d = ref.pipeline(dev)
        for a, b in zip(outs_h, outs_d):
            assert np.array_equal(a, b.cpu().numpy(), equal_nan=True)
        # 8-bit host image: bytes travel, widening on the device (test/test.cpp:73,85)
        u8 = (rng.random((1531, 1100)) * 255).astype(np.uint8)
        f8 = cv.SteerableFiltersG2(None)
        f8.set_option(L.OPT_HOST_OVERLAP, overlap)
        g8, h8 = f8.setup_steer(u8, 0.3)
        gr, hr = ref.setup_steer(torch.from_numpy(u8.astype(np.float32)).cuda(), 0.3)
        assert np.array_equal(g8, gr.cpu().numpy()) and np.array_equal(h8, hr.cpu().numpy())
    # G4 through the same path
    f4, r4 = cv.SteerableFiltersG4(None), cv.SteerableFiltersG4(None)
    g4, h4 = f4.setup_steer(img, 0.3)
    g4d, h4d = r4.setup_steer(dev, 0.3)
    assert np.array_equal(g4, g4d.cpu().numpy()) and np.array_equal(h4, h4d.cpu().numpy())
    for p in (0, 10):
        assert np.array_equal(f4.basis(p), r4.basis(p).cpu().numpy())


def test_hip_graph_capture_and_replay(cv):
    """the engine's launches can be captured into a HIP graph on the caller's stream (no tuning and no allocation happen
    under capture) and replayed on new data in the same buffers"""
    import torch
    img = torch.rand((1200, 1600), device="cuda")
    f = cv.SteerableFiltersG2(None)
    g, h = torch.empty_like(img), torch.empty_like(img)
    outs = [torch.empty_like(img) for _ in range(8)]

    def work():
        f.setup_steer(img, 0.3, out=(g, h))
        f.pipeline(img, out=outs)

    for _ in range(3):
        work()                                  # state allocation and order tuning happen here
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        work()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            work()
    torch.cuda.synchronize()
    img.copy_(torch.rand((1200, 1600), device="cuda"))      # new frame, same buffer
    graph.replay()
    torch.cuda.synchronize()
    got = [g.clone(), h.clone()] + [o.clone() for o in outs] + [f.basis(p).clone() for p in range(7)]
    work()
    torch.cuda.synchronize()
    want = [g, h] + outs + [f.basis(p) for p in range(7)]
    for a, b in zip(got, want):
        assert torch.equal(a, b)


@pytest.mark.parametrize("kind", [2, 4])
def test_dynamic_tail_replays_and_eager_launches_mixed(cv, kind):
    """the dynamic tail (block order 2000000) hands the last tiles out from per-handle queues in device memory, two sets used
    alternately.  A captured launch bakes its set into the graph; ONE such launch in a graph, then (replay, eager) twice -- the
    sequence in which the second eager launch used to find its set exhausted by the replay and left the tail tiles unwritten
    (round-4 advisor) -- and eager, eager, replay, replay for good measure: every launch must write every tile."""
    import torch
    from cvsteer_amd import _lib as L
    shape = (1400, 1800)
    imgs = [torch.rand(shape, device="cuda") for _ in range(2)]
    cls, nb = (cv.SteerableFiltersG2, 7) if kind == 2 else (cv.SteerableFiltersG4, 11)
    ref = cls(None)
    ref.set_option(L.OPT_BLOCK_ORDER, L.ORDER_PLAIN)
    want = []
    for im in imgs:
        gq, hq = ref.setup_steer(im, 0.3)
        want.append([gq.clone(), hq.clone()] + [ref.basis(p).clone() for p in range(nb)])
    f = cls(None)
    f.set_option(L.OPT_BLOCK_ORDER, L.ORDER_DYNAMIC_TAIL)
    f.set_option(L.OPT_AUTOTUNE, 0)
    buf = torch.empty(shape, device="cuda")
    g, h = torch.empty_like(buf), torch.empty_like(buf)

    def run():
        f.setup_steer(buf, 0.3, out=(g, h))

    def poison():     # a tile that is not written keeps this
        g.fill_(-7.0); h.fill_(-7.0)
        for p in range(nb):
            f.basis(p).fill_(-7.0)

    def check(which, tag):
        torch.cuda.synchronize()
        cur = [g, h] + [f.basis(p) for p in range(nb)]
        for a_, b_ in zip(cur, want[which]):
            assert torch.equal(a_, b_), tag

    buf.copy_(imgs[0]); run(); check(0, "first eager")
    assert f.launch_info()["block_order"] == L.ORDER_DYNAMIC_TAIL
    graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(side):
        run()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            run()                                # exactly ONE dynamic launch in the graph
    torch.cuda.synchronize()
    seq = ["replay", "eager", "replay", "eager", "eager", "eager", "replay", "replay", "eager"]
    for n, what in enumerate(seq):
        which = n & 1
        buf.copy_(imgs[which])
        poison()
        torch.cuda.synchronize()
        if what == "replay":
            graph.replay()
        else:
            run()
        check(which, (n, what))


def test_one_object_per_image_per_worker_thread(cv):
    """The reference's usage model (example/steer.cpp:69-71,86: cv::parallel_for_ over files, one SteerableFiltersG2
    per image inside the body): eight host threads, each constructing short-lived objects on its own images --
    device planes and host planes, two sizes (one large enough for the launch tuner) -- while the others do the same.
    Everything process-wide (state-block cache, tuner memory, tile-queue slots) is shared; results must equal the serial
    ones bit for bit."""
    import threading
    import torch
    shapes = [(200, 333), (2112, 4096)]
    rng = np.random.default_rng(77)
    images = [[torch.from_numpy(rng.random(s, dtype=np.float32)).cuda() for _ in range(3)] for s in shapes]
    serial = [[[o.clone() for o in cv.SteerableFiltersG2(im).pipeline(im)] for im in group] for group in images]
    host_in = images[0][1].cpu().numpy()
    errors = []

    def worker(tid):
        try:
            for it in range(6):
                g = (tid + it) % 2
                k = (tid * 7 + it) % 3
                im = images[g][k]
                f = cv.SteerableFiltersG2(im)                      # ctor = setup (G2.cpp:57)
                outs = f.pipeline(im)
                for a_, b_ in zip(outs, serial[g][k]):
                    if not torch.equal(a_, b_):
                        errors.append(("device", tid, it, g, k))
                        return
                if it % 3 == 0:                                      # the cv::Mat way: host in, host out
                    fh = cv.SteerableFiltersG2(None)
                    houts = fh.pipeline(host_in)
                    for a_, b_ in zip(houts, serial[0][1]):
                        if not np.array_equal(np.asarray(a_), b_.cpu().numpy()):
                            errors.append(("host", tid, it))
                            return
                del f
        except Exception as ex:   # noqa: BLE001 -- reported below
            errors.append(("exception", tid, repr(ex)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a worker thread hangs"
    assert not errors, errors[:3]


@pytest.mark.parametrize("shape", [(64, 130), (185, 256), (540, 960)])
def test_packed_and_plain_arithmetic_agree_bit_for_bit(cv, shape):
    """The basis, fused-steer and full-setup launches filter with packed f32 pairs (v_pk_fma_f32: a mirror and an anti-mirror kernel in
    the halves of a register pair); the pipeline variants keep plain instructions (cvs_kernels_basis.hip, "Packed f32").  Same
    operations in the same order per result, so the two must agree BIT FOR BIT: the twelve state planes a full setup writes (packed)
    against the ones the caller pipeline leaves behind (plain), and the feature maps of the three pipeline forms among themselves,
    on random frames with a few non-finite pixels."""
    import torch
    rows, cols = shape
    gen = torch.Generator(device="cuda").manual_seed(rows * 7 + cols)
    frames = torch.rand((3, rows, cols), device="cuda", generator=gen) * 255.0
    frames[1, rows // 2, cols // 3] = float("inf")
    frames[2, 5, 7] = float("nan")

    def same(a, b):
        return torch.equal(a.isnan(), b.isnan()) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))

    three_h, eight_h = cv.SteerableFiltersG2(None), cv.SteerableFiltersG2(None)
    three_h.set_persist(False)
    eight_h.set_persist(False)
    three = three_h.pipeline_batch(frames, outputs=(5, 6, 7))
    eight = eight_h.pipeline_batch(frames)
    packed, plain = cv.SteerableFiltersG2(None), cv.SteerableFiltersG2(None)
    for i in range(3):
        img = frames[i].contiguous()
        packed.setup(img, flags=cv.SETUP_FULL)                    # FLAGS 1: packed arithmetic
        one = plain.pipeline(img)                                  # FLAGS 5: plain arithmetic, state kept
        for p in list(range(7)) + [L_.PLANE_C1, L_.PLANE_C2, L_.PLANE_C3, L_.PLANE_THETA, L_.PLANE_STRENGTH]:
            assert same(packed._state(p), plain._state(p)), (i, "state plane", p)
        g, h = packed.setup_steer(img, 0.3, flags=cv.SETUP_BASIS)  # FLAGS 2: packed, fused scalar steer
        for p in range(7):
            assert same(packed._state(p), plain._state(p)), (i, "state plane after the fused steer", p)
        for k in range(3):
            assert same(three[i, k], eight[i, 5 + k]) and same(three[i, k], one[5 + k]), (i, k)
