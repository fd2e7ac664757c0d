"""GPU tests (-m gpu) against REAL OpenCV, where it exists ("vs real cv::sepFilter2D iff OpenCV is found on the
box").  No image this build has run on has had cv2 -- the module then skips as a whole -- but the day one
does, this is the literal reference call sequence (cvsteer/SteerableFiltersG2.cpp:62-99, 107-112, 137-177;
SteerableFiltersG4.cpp:69-80, 114-122) next to the HIP planes AND next to the oracle, so that every `[recalled]` item of
SURVEY.md 8(c) (fastAtan32f constants, SymmColumnFilter order, scalar narrowing) is finally checked against the real thing.
Nothing of OpenCV is vendored; nothing here runs in the product path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

cv2 = pytest.importorskip("cv2", reason="OpenCV is not installed on this image (every box so far): the reference-vs-HIP leg skips")

TOL = 1e-5


@pytest.fixture(scope="module")
def cv():
    import cvsteer_amd
    return cvsteer_amd


def _sep(img, kx, ky):
    # cv::sepFilter2D(image, dst, CV_32FC1, kx, ky.t()): kx along rows, ky along columns, BORDER_DEFAULT (G2.cpp:62-68)
    return cv2.sepFilter2D(img, cv2.CV_32F, kx.reshape(1, -1), ky.reshape(-1, 1))


def _basis_cv(cv, kind, img, width, spacing):
    nb = cv.num_basis(kind)
    taps = [cv.make_taps(kind, i, width, spacing) for i in range(nb)]
    return [_sep(img, taps[kx], taps[ky]) for kx, ky in (cv.basis_taps(kind, p) for p in range(nb))]


def _wrap(a):
    # SteerableFilters.cpp:46-51
    return np.where(a > np.float32(np.pi), a - np.float32(2 * np.pi), a).astype(np.float32)


def _g2_setup_cv(b):
    g2a, g2b, g2c, h2a, h2b, h2c, h2d = b
    m = lambda x, y: x * y
    c1 = (0.5 * m(g2b, g2b) + 0.25 * m(g2a, g2c) + 0.375 * (m(g2a, g2a) + m(g2c, g2c)) + 0.3125 * (m(h2a, h2a) + m(h2d, h2d)) +
          0.5625 * (m(h2b, h2b) + m(h2c, h2c)) + 0.375 * (m(h2a, h2c) + m(h2b, h2d))).astype(np.float32)
    c2 = (0.5 * (m(g2a, g2a) - m(g2c, g2c)) + 0.46875 * (m(h2a, h2a) - m(h2d, h2d)) + 0.28125 * (m(h2b, h2b) - m(h2c, h2c)) +
          0.1875 * (m(h2a, h2c) - m(h2b, h2d))).astype(np.float32)
    c3 = (-m(g2a, g2b) - m(g2b, g2c) - 0.9375 * (m(h2c, h2d) + m(h2a, h2b)) - 1.6875 * m(h2b, h2c) - 0.1875 * m(h2a, h2d)).astype(np.float32)
    strength, theta = cv2.cartToPolar(c2, c3)               # G2.cpp:97
    theta = (_wrap(theta) * np.float32(0.5)).astype(np.float32)   # G2.cpp:98-99
    return c1, c2, c3, theta, strength


@pytest.mark.parametrize("shape", [(185, 256), (301, 449), (1080, 1920)])
def test_basis_planes_against_cv_sepfilter2d(cv, ora, shape):
    """north_star: <= 1e-5 max-abs vs cv::sepFilter2D on [0,1) inputs -- G2 (7 planes) and G4 (11 planes); the oracle is held
    against the same planes (it claims to restate OpenCV's f32 row / folded column filters)"""
    img = np.random.default_rng(shape[0]).random(shape, dtype=np.float32)
    for kind, cls, w, s in ((2, cv.SteerableFiltersG2, 4, 0.67), (4, cv.SteerableFiltersG4, 6, 0.5)):
        want = _basis_cv(cv, kind, img, w, s)
        f = cls(img, w, s)
        got = [f.basis(p) for p in range(len(want))]
        orc = ora.basis(kind, img, w, s)
        for p, (a, b) in enumerate(zip(got, want)):
            assert np.abs(a - b).max() <= TOL, (kind, p, "HIP vs OpenCV")
            assert np.abs(orc[p] - b).max() <= TOL, (kind, p, "oracle vs OpenCV")


def test_g2_setup_steer_and_features_against_cv(cv, ora):
    """the callers' sequence (test/test.cpp:85-90) stage by stage: each stage is fed OpenCV's own upstream planes"""
    from helpers import angle_diff, smooth_image
    img = (0.5 * np.random.default_rng(1).random((240, 320), dtype=np.float32) + smooth_image(240, 320)).astype(np.float32)
    b = _basis_cv(cv, 2, img, 4, 0.67)
    c1, c2, c3, theta, strength = _g2_setup_cv(b)
    f = cv.SteerableFiltersG2(img)
    th, st = f.getDominantOrientationAngle(), f.getDominantOrientationStrength()
    sel = strength > 1e-3
    assert np.abs(st - strength).max() <= TOL * max(1.0, float(strength.max()))
    assert angle_diff(th[sel], theta[sel], period=np.pi).max() <= TOL      # fastAtan2, wrap, * 0.5 -- modulo the branch cut
    # steer at OpenCV's theta map (G2.cpp:147-155): polarToCart + the weighted sums
    ct, stn = cv2.polarToCart(None, theta)
    g2 = ct * ct * b[0] + (-2.0 * ct * stn) * b[1] + stn * stn * b[2]
    h2 = ct * ct * ct * b[3] + (-3.0 * ct * ct * stn) * b[4] + (3.0 * ct * stn * stn) * b[5] + (-stn * stn * stn) * b[6]
    got = f.steer(theta, full=True)
    assert np.abs(got[0] - g2).max() <= TOL and np.abs(got[1] - h2).max() <= TOL
    mag, phase = cv2.cartToPolar(g2.astype(np.float32), h2.astype(np.float32))   # G2.cpp:107-112
    phase = np.nan_to_num(_wrap(phase), nan=0.0)
    gm, gp = f.computeMagnitudeAndPhase(g2.astype(np.float32), h2.astype(np.float32))
    assert np.abs(gm - mag).max() <= TOL
    big = mag > 1e-3
    assert angle_diff(gp[big], phase[big], period=2 * np.pi).max() <= TOL
    # the oracle's cv-compatible arctangent against the real cv::fastAtan2 -- the [recalled] polynomial
    om, op = ora.mag_phase(g2.astype(np.float32), h2.astype(np.float32))
    assert angle_diff(op[big], phase[big], period=2 * np.pi).max() <= 1e-6
