"""ctypes/numpy wrapper over oracle/liboracle_cvsteer.so (TEST INFRASTRUCTURE ONLY).

Function names mirror cvsteer_oracle.h; every array is a dense C-contiguous numpy array.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("ORACLE_LIB", os.path.join(_HERE, "liboracle_cvsteer.so"))  # override: the sanitizer twin (tools/run_sanitizers.sh)

KIND_G2, KIND_G4 = 2, 4
ATAN_CV, ATAN_EXACT = 0, 1

_f = C.POINTER(C.c_float)
_d = C.POINTER(C.c_double)


def build(force=False):
    """Compile the oracle with gcc (build() of __graft_entry__ calls this)."""
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.ora_time_g2_filter_steer.restype = C.c_double
        L.ora_time_g2_filter_steer_mt.restype = C.c_double
        _lib = L
    return _lib


def _fp(a):
    return None if a is None else a.ctypes.data_as(_f)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_filters(kind):
    return lib().ora_num_filters(kind)


def make_taps(kind, idx, width, spacing):
    out = np.empty(2 * width + 1, np.float32)
    rc = lib().ora_make_taps(kind, idx, width, C.c_float(spacing), _fp(out))
    assert rc == 0
    return out


def basis_pair(kind, p):
    a, b = C.c_int(), C.c_int()
    assert lib().ora_basis_pair(kind, p, C.byref(a), C.byref(b)) == 0
    return a.value, b.value


def reflect101(p, n):
    return lib().ora_reflect101(p, n)


def sepfilter2d(src, kx, ky, f64=False):
    src = _f32(src); kx = _f32(kx); ky = _f32(ky)
    rows, cols = src.shape
    w = (len(kx) - 1) // 2
    assert len(kx) == len(ky) == 2 * w + 1
    if f64:
        dst = np.empty((rows, cols), np.float64)
        lib().ora_sepfilter2d_f64(_fp(src), rows, cols, C.c_size_t(cols), _fp(kx), _fp(ky), w,
                                  dst.ctypes.data_as(_d))
    else:
        dst = np.empty((rows, cols), np.float32)
        lib().ora_sepfilter2d_f32(_fp(src), rows, cols, C.c_size_t(cols), _fp(kx), _fp(ky), w, _fp(dst))
    return dst


def basis(kind, src, width, spacing, f64=False):
    src = _f32(src)
    rows, cols = src.shape
    n = num_filters(kind)
    if f64:
        out = np.empty((n, rows, cols), np.float64)
        lib().ora_basis_f64(kind, _fp(src), rows, cols, C.c_size_t(cols), width, C.c_float(spacing),
                            out.ctypes.data_as(_d))
    else:
        out = np.empty((n, rows, cols), np.float32)
        lib().ora_basis(kind, _fp(src), rows, cols, C.c_size_t(cols), width, C.c_float(spacing), _fp(out))
    return out


def cart_to_polar(x, y, mode=ATAN_CV):
    x = _f32(x); y = _f32(y)
    mag = np.empty_like(x); ang = np.empty_like(x)
    lib().ora_cart_to_polar(_fp(x), _fp(y), C.c_size_t(x.size), _fp(mag), _fp(ang), mode)
    return mag, ang


def polar_to_cart(a):
    a = _f32(a)
    c = np.empty_like(a); s = np.empty_like(a)
    lib().ora_polar_to_cart(_fp(a), C.c_size_t(a.size), _fp(c), _fp(s))
    return c, s


def wrap(a):
    a = _f32(a)
    out = np.empty_like(a)
    lib().ora_wrap(_fp(a), C.c_size_t(a.size), _fp(out))
    return out


def g2_orientation(b, mode=ATAN_CV):
    """b: (7,H,W) -> c1,c2,c3,theta,strength"""
    b = _f32(b)
    shp = b.shape[1:]
    outs = [np.empty(shp, np.float32) for _ in range(5)]
    lib().ora_g2_orientation(_fp(b), C.c_size_t(b[0].size), *[_fp(o) for o in outs], mode)
    return tuple(outs)


def mag_phase(g, h, mode=ATAN_CV):
    g = _f32(g); h = _f32(h)
    m = np.empty_like(g); p = np.empty_like(g)
    lib().ora_mag_phase(_fp(g), _fp(h), C.c_size_t(g.size), _fp(m), _fp(p), mode)
    return m, p


def g2_steer_scalar(b, theta, c=None, mode=ATAN_CV):
    """-> g2,h2 (and e,mag,phase when c=(c1,c2,c3) is given)"""
    b = _f32(b)
    shp = b.shape[1:]
    full = c is not None
    outs = [np.empty(shp, np.float32) for _ in range(5 if full else 2)]
    cs = [_f32(x) for x in c] if full else [None] * 3
    ptrs = [_fp(o) for o in outs] + [None] * (5 - len(outs))
    lib().ora_g2_steer_scalar(_fp(b), *[_fp(x) for x in cs], C.c_size_t(b[0].size), C.c_float(theta),
                              *ptrs, mode)
    return tuple(outs)


def g2_steer_map(b, theta, c=None, mode=ATAN_CV):
    b = _f32(b); theta = _f32(theta)
    shp = b.shape[1:]
    full = c is not None
    outs = [np.empty(shp, np.float32) for _ in range(5 if full else 2)]
    cs = [_f32(x) for x in c] if full else [None] * 3
    ptrs = [_fp(o) for o in outs] + [None] * (5 - len(outs))
    lib().ora_g2_steer_map(_fp(b), *[_fp(x) for x in cs], C.c_size_t(b[0].size), _fp(theta), *ptrs, mode)
    return tuple(outs)


def g2_steer_point(b, c, row, col, theta):
    b = _f32(b)
    cs = [_f32(x) for x in c]
    out = np.empty(5, np.float32)
    i = row * b.shape[2] + col
    lib().ora_g2_steer_point(_fp(b), *[_fp(x) for x in cs], C.c_size_t(b[0].size), C.c_size_t(i),
                             C.c_float(theta), _fp(out))
    return out


def phase_weights(phase, phi, signum, k=2.0):
    phase = _f32(phase)
    lam = np.empty_like(phase)
    lib().ora_phase_weights(_fp(phase), C.c_size_t(phase.size), C.c_float(phi), int(bool(signum)),
                            C.c_float(k), _fp(lam))
    return lam


def find(e, phase):
    """-> edges, dark, bright"""
    e = _f32(e); phase = _f32(phase)
    outs = [np.empty_like(e) for _ in range(3)]
    lib().ora_find(_fp(e), _fp(phase), C.c_size_t(e.size), *[_fp(o) for o in outs])
    return tuple(outs)


def g4_steer_scalar(b, theta):
    b = _f32(b)
    g = np.empty(b.shape[1:], np.float32); h = np.empty_like(g)
    lib().ora_g4_steer_scalar(_fp(b), C.c_size_t(g.size), C.c_float(theta), _fp(g), _fp(h))
    return g, h


def g4_steer_map(b, theta):
    b = _f32(b); theta = _f32(theta)
    g = np.empty(b.shape[1:], np.float32); h = np.empty_like(g)
    lib().ora_g4_steer_map(_fp(b), C.c_size_t(g.size), _fp(theta), _fp(g), _fp(h))
    return g, h


def g4_orientation(b, mode=ATAN_CV):
    """EXTENSION (not in the reference).  b: (11,H,W) -> c1,c2,c3,theta,strength"""
    b = _f32(b)
    outs = [np.empty(b.shape[1:], np.float32) for _ in range(5)]
    lib().ora_g4_orientation(_fp(b), C.c_size_t(b[0].size), *[_fp(o) for o in outs], mode)
    return tuple(outs)


def pyr_down(src):
    src = _f32(src)
    rows, cols = src.shape
    dst = np.empty(((rows + 1) // 2, (cols + 1) // 2), np.float32)
    lib().ora_pyr_down(_fp(src), rows, cols, C.c_size_t(cols), _fp(dst))
    return dst


def sepfilter2d_f32_rows(src, kx, ky, y_lo, y_hi, dst):
    """rows [y_lo, y_hi) of sepfilter2d_f32 written into the dense plane dst"""
    src = _f32(src)
    kx, ky = _f32(kx), _f32(ky)
    lib().ora_sepfilter2d_f32_rows(_fp(src), src.shape[0], src.shape[1], C.c_size_t(src.shape[1]), _fp(kx), _fp(ky),
                                   (kx.size - 1) // 2, _fp(dst), int(y_lo), int(y_hi))
    return dst


def time_g2_filter_steer(src, theta, reps=1):
    src = _f32(src)
    return lib().ora_time_g2_filter_steer(_fp(src), src.shape[0], src.shape[1], C.c_float(theta), reps)


def time_g2_filter_steer_mt(src, theta, reps=1, threads=1):
    """one image, rows split over `threads` host threads"""
    src = _f32(src)
    return lib().ora_time_g2_filter_steer_mt(_fp(src), src.shape[0], src.shape[1], C.c_float(theta), reps, int(threads))


# ---- caller-side steps of the reference test (test/test.cpp:92-103), numpy only ----
def normalize_minmax_u8(a):
    """cv::normalize(src, dst, 0, 255, NORM_MINMAX, CV_8UC1): scale=(255/(max-min)), shift=-min*scale,
    saturate_cast<uchar> rounds half to even."""
    a = np.asarray(a, np.float64)
    lo, hi = a.min(), a.max()
    scale = 255.0 / (hi - lo) if hi > lo else 0.0
    v = np.rint(a * scale - lo * scale)
    return np.clip(v, 0, 255).astype(np.uint8)
