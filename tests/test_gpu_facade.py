"""GPU test (-m gpu): the C++ drop-in facade (fa::SteerableFiltersG2/G4 in libcvsteer.so) running the
reference's own test body (tests/cpp/test_basic.cpp mirrors test/test.cpp:70-108)."""
import io
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _recode(u8):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(u8).save(buf, format="JPEG", quality=95)
    return np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("L"))


def test_cpp_facade_runs_reference_test(ora, fish, golden_dir, tmp_path):
    exe = os.path.join(ROOT, "tests", "cpp", "test_basic")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cvsteer_amd", "facade"), "-s"])
    r = subprocess.run([exe, os.path.join(golden_dir, "fish_u8.npy"), str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cvsteer.basic OK" in r.stdout
    shape = fish.shape
    for name in ("edges", "linesDark", "linesBright"):
        plane = np.fromfile(os.path.join(str(tmp_path), name + ".f32"), np.float32).reshape(shape)
        gt = np.load(os.path.join(golden_dir, name + "_u8.npy")).astype(np.float64)
        # test.cpp:92-103: normalize(0,255,MINMAX,8U) -> recode -> L1/total <= 1.0
        err = np.abs(_recode(ora.normalize_minmax_u8(plane)).astype(np.float64) - gt).mean()
        assert err <= 1.0, (name, err)
        assert err <= 0.05, (name, err)
    # the facade's theta equals the oracle's end-to-end where orientation is well conditioned
    theta = np.fromfile(os.path.join(str(tmp_path), "theta.f32"), np.float32).reshape(shape)
    o = ora.g2_orientation(ora.basis(2, fish, 4, 0.67))
    ok = o[4] > 0.05 * o[4].max()
    d = np.abs(theta.astype(np.float64) - o[3]) % np.pi
    assert np.minimum(d, np.pi - d)[ok].max() <= 1e-3
    # G4 scalar steer through the facade vs the oracle
    g4 = np.fromfile(os.path.join(str(tmp_path), "g4.f32"), np.float32).reshape(shape)
    og, oh = ora.g4_steer_scalar(ora.basis(4, fish, 6, 0.5), 0.3)
    assert np.abs(g4 - og).max() <= 1e-5 * max(1.0, np.abs(og).max())


def test_cpp_facade_protected_create_and_wrap():
    """SteerableFilters::create / ::wrap (SteerableFilters.cpp:33-51) are protected statics: a tiny subclass in
    tests/cpp/test_protected.cpp calls them like user code would; wrap runs on the GPU through cvs_wrap."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_protected")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cvsteer_amd", "facade"), "-s"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cvsteer.protected OK" in r.stdout
