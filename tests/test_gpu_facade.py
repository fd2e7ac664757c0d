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


def test_cpp_subclass_written_like_the_reference_reads_protected_members():
    """code that EXTENDS the reference: tests/cpp/test_subclass.cpp is written inside _STEER_BEGIN / _STEER_END
    (cvsteer/cvsteer.h:12-15) and reads m_g2a..m_h2d, m_c1..m_c3, m_theta (SteerableFiltersG2.h:62-66) and m_g4a..m_h4f
    (SteerableFiltersG4.h:53-54) from subclasses; the facade fills those host copies for subclass objects."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_subclass")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cvsteer_amd", "facade"), "-s"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cvsteer.subclass OK" in r.stdout


def test_cpp_batch_driver_matches_goldens_and_python_driver(tmp_path):
    """cvsteer_amd/cvsteer-run (facade/cvsteer_run.cpp): the reference's example/steer.cpp flow in C++ over the C ABI --
    host planes into cvs_batch_run, maps kept on the GPUs, 8-bit conversion on the GPU.  Checked against the reference's
    golden images (same bar as test/test.cpp:96-106), bit for bit against the Python driver, with mixed sizes in one list,
    PGM and .npy, the --gain branch, a rehearsal world of three ranks, and an unreadable file."""
    import io, subprocess, sys
    import numpy as np
    from PIL import Image
    exe = os.path.join(ROOT, "cvsteer_amd", "cvsteer-run")
    assert os.path.exists(exe), "build it: make -C cvsteer_amd/facade"
    gold = os.path.join(ROOT, "tests", "golden")
    fish = np.load(os.path.join(gold, "fish_u8.npy"))
    src = tmp_path / "in"
    src.mkdir()
    np.save(str(src / "fish.npy"), fish)
    with open(str(src / "fishp.pgm"), "wb") as f:      # the same pixels as PGM, with a header comment
        f.write(b"P5\n# the reference's fish\n%d %d\n255\n" % (fish.shape[1], fish.shape[0]) + fish.tobytes())
    rng = np.random.default_rng(5)
    np.save(str(src / "noise_a.npy"), rng.random((70, 130), dtype=np.float32) * 255)
    np.save(str(src / "noise_b.npy"), rng.random((70, 130), dtype=np.float32) * 255)
    names = ["fish.npy", "fishp.pgm", "noise_a.npy", "noise_b.npy", "fish.npy"]
    lst = tmp_path / "files.txt"
    lst.write_text("".join(str(src / n) + "\n" for n in names) + str(src / "missing.npy") + "\n")
    out_c, out_p = tmp_path / "c", tmp_path / "p"
    out_c.mkdir(); out_p.mkdir()
    r = subprocess.run([exe, "--input", str(lst), "--output", str(out_c), "--ext", ".npy", "--devices", "0,0,0", "--verbose"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "missing.npy" in r.stderr, r.stdout + r.stderr      # the unreadable file is reported
    ok = tmp_path / "ok.txt"
    ok.write_text("".join(str(src / n) + "\n" for n in names if n.endswith(".npy")))
    r = subprocess.run([sys.executable, "-m", "cvsteer_amd.run", "--input", str(ok), "--output", str(out_p), "--ext", ".npy"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr

    def recode(u8):
        buf = io.BytesIO()
        Image.fromarray(u8).save(buf, format="JPEG", quality=95)
        return np.asarray(Image.open(io.BytesIO(buf.getvalue())))

    for name, suffix in (("edges", "_edges"), ("linesDark", "_lines_dark"), ("linesBright", "_lines_bright")):
        got = np.load(str(out_c / ("fish" + suffix + ".npy")))
        gt = np.load(os.path.join(gold, name + "_u8.npy")).astype(np.float64)
        assert got.dtype == np.uint8 and np.abs(recode(got).astype(np.float64) - gt).mean() <= 1.0
        assert np.array_equal(got, np.load(str(out_c / ("fishp" + suffix + ".npy"))))           # PGM input = npy input
        for base in ("fish", "noise_a", "noise_b"):
            assert np.array_equal(np.load(str(out_c / (base + suffix + ".npy"))), np.load(str(out_p / (base + suffix + ".npy")))), (base, suffix)
    # --gain (steer.cpp:92-97) and PGM output, one GPU, one file
    r = subprocess.run([exe, "-i", str(src / "fish.npy"), "-o", str(out_c), "--gain", "2.0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    pg = np.asarray(Image.open(str(out_c / "fish_edges.pgm")))
    r = subprocess.run([sys.executable, "-m", "cvsteer_amd.run", "--input", str(src / "fish.npy"), "--output", str(out_p), "--gain", "2.0", "--ext", ".npy"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert pg.dtype == np.uint8 and pg.max() == 255 and np.array_equal(pg, np.load(str(out_p / "fish_edges.npy")))
