"""CPU tests (-m "not gpu"): the C-ABI library loads without a GPU, exports every symbol that
include/cvsteer_hip.h declares, its host-side math is right, and it refuses to run without HIP."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cv():
    so = os.path.join(ROOT, "cvsteer_amd", "libcvsteer_hip.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cvsteer_amd", "csrc"), "-s", "-j4"])
    import cvsteer_amd
    return cvsteer_amd


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cvsteer_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cvs_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(cv):
    from cvsteer_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 28
    raw = C.CDLL(cv.lib_path())
    for name in declared:
        assert hasattr(raw, name), "library does not export " + name
        assert name in _lib.SIGNATURES, "python binding misses " + name
    assert sorted(_lib.SIGNATURES) == declared
    assert cv.abi_version() == 2


def test_host_taps_bit_exact_vs_reference_functions(cv, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "taps_ref.json")))
    for section, kind in (("g2", 2), ("g4", 4), ("g2_w6_s05", 2), ("g4_w8_s04", 4)):
        sec = g[section]
        sp = float(np.frombuffer(bytes.fromhex(sec["spacing_hex"])[::-1], dtype=np.float32)[0])
        names = g["g2"]["order"] if kind == 2 else g["g4"]["order"]
        for i, nm in enumerate(names):
            t = cv.make_taps(kind, i, sec["width"], sp)
            assert [format(int(u), "08x") for u in t.view(np.uint32)] == sec[nm], (section, nm)


def test_basis_tap_pairs_and_weights_match_oracle(cv, ora):
    for kind in (2, 4):
        n = cv.num_basis(kind)
        assert n == ora.num_filters(kind)
        for p in range(n):
            assert cv.basis_taps(kind, p) == ora.basis_pair(kind, p)
    # scalar steering weights: identities of the Freeman-Adelson interpolation functions
    for th in (0.0, 0.3, -1.1, 2.5):
        w = cv.steer_weights(2, th).astype(np.float64)
        c, s = np.cos(np.float32(th)), np.sin(np.float32(th))
        assert np.allclose(w, [c * c, -2 * c * s, s * s, c ** 3, -3 * c * c * s, 3 * c * s * s, -s ** 3], atol=2e-7)
        w4 = cv.steer_weights(4, th).astype(np.float64)
        assert np.allclose(w4[:5], [c ** 4, -4 * c ** 3 * s, 6 * c * c * s * s, -4 * c * s ** 3, s ** 4], atol=3e-7)
        assert np.allclose(w4[5:], [c ** 5, -5 * c ** 4 * s, 10 * c ** 3 * s * s, -10 * c * c * s ** 3, 5 * c * s ** 4, -s ** 5], atol=3e-7)
    with pytest.raises(cv.CvsError):
        cv.make_taps(3, 0, 4, 0.67)
    with pytest.raises(cv.CvsError):
        cv.basis_taps(2, 7)


def test_no_cpu_fallback(cv):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(cv.CvsError) as ei:
        cv.SteerableFiltersG2(np.zeros((8, 8), np.float32))
    assert ei.value.status == -3  # CVS_E_HIP: the product refuses to run without the GPU


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under cvsteer_amd/ or include/ may reference it"""
    bad = []
    for base in ("cvsteer_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"\boracle\b|liboracle|cvsteer_oracle", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_facade_library_exports_reference_class_surface():
    so = os.path.join(ROOT, "cvsteer_amd", "libcvsteer.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cvsteer_amd", "facade"), "-s"])
    syms = subprocess.check_output(["nm", "-DC", "--defined-only", so], text=True)
    for needle in ("fa::SteerableFiltersG2::SteerableFiltersG2(fa::Mat1f const&, int, float)",
                   "fa::SteerableFiltersG2::setup(fa::Mat1f const&)",
                   "fa::SteerableFiltersG2::steer(float, fa::Mat1f&, fa::Mat1f&)",
                   "fa::SteerableFiltersG2::steer(fa::Mat1f const&, fa::Mat1f&, fa::Mat1f&, fa::Mat1f&, fa::Mat1f&, fa::Mat1f&)",
                   "fa::SteerableFiltersG2::steer(fa::Point const&, float, float&, float&)",
                   "fa::SteerableFiltersG2::computeMagnitudeAndPhase(",
                   "fa::SteerableFiltersG2::findEdges(", "fa::SteerableFiltersG2::findDarkLines(",
                   "fa::SteerableFiltersG2::findBrightLines(", "fa::SteerableFiltersG2::phaseWeights(",
                   "fa::SteerableFiltersG2::getDominantOrientationAngle() const",
                   "fa::SteerableFiltersG4::SteerableFiltersG4(fa::Mat1f const&, int, float)",
                   "fa::SteerableFiltersG4::steer(float, fa::Mat1f&, fa::Mat1f&)",
                   "fa::SteerableFiltersG4::steer(fa::Mat1f const&, fa::Mat1f&, fa::Mat1f&)",
                   "fa::SteerableFilters::create(int, float, float (*)(float))",
                   "fa::SteerableFilters::wrap(fa::Mat1f const&, fa::Mat1f&)"):
        assert needle in syms, needle


def test_literal_tap_table_is_the_reference_taps_fixture(tmp_path):
    """cvsteer_amd/csrc/cvs_lit_taps.h (the default G2 / H2 taps compiled into the k_basis_lit instances) is what tools/gen_lit_taps.py makes of
    tests/golden/taps_ref.json -- the bit patterns of the reference's own tap functions -- and nothing else"""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "cvsteer_amd", "csrc", "cvs_lit_taps.h")
    committed = open(hdr).read()
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_lit_taps.py")], capture_output=True, text=True, check=True).stdout
    assert open(hdr).read() == committed, "tools/gen_lit_taps.py changes the committed header"
    assert committed.strip() == out.strip()
    import json
    g2 = json.load(open(os.path.join(root, "tests", "golden", "taps_ref.json")))["g2"]
    words = re.findall(r"0x([0-9a-f]{8})u", committed)
    assert len(words) == 30
    assert words[:5] == [g2["G21"][4 + i] for i in range(5)] and words[25:] == [g2["H23"][4 + i] for i in range(5)]
