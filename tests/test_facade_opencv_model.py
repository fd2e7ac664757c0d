"""The facade's OpenCV branch (include/cvsteer/Mat.h: `typedef cv::Mat1f Mat1f; typedef cv::Point Point;` -- the types the
reference's public surface is written in, cvsteer/SteerableFilters.h:37,44-49) on an image that has no OpenCV.

`tests/cpp/opencv_model/opencv2/core/core.hpp` is a declaration-level MODEL of cv::Mat / cv::Mat_<float> / cv::Point (member
names and types as OpenCV documents them: `uchar* data`, `MatStep step`, `uchar* ptr(int)`, `Mat_<T>::operator()(Point)` ...);
it is not OpenCV and nothing of it is shipped.  With it on the include path `__has_include(<opencv2/core/core.hpp>)` is
true, so the facade and the reference's own test body compile -- and, on a GPU box, run -- with fa::Mat1f BEING that cv::Mat1f.
What this proves: the branch is valid C++ against those declarations and gives the same planes as the stand-in build.  What it
cannot prove: agreement with a real OpenCV installation (tests/test_gpu_opencv.py, and a build with the real headers)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODEL = os.path.join(ROOT, "tests", "cpp", "opencv_model")
FLAGS = ["-std=c++11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + MODEL]


def test_opencv_branch_is_taken_and_is_cv_mat1f(tmp_path):
    src = tmp_path / "chk.cpp"
    src.write_text('#include <type_traits>\n#include <cvsteer/SteerableFiltersG2.h>\n'
                   '#ifndef CVSTEER_HAVE_OPENCV\n#error "the OpenCV branch was not taken"\n#endif\n'
                   '#ifndef CVSTEER_TESTS_OPENCV_MODEL\n#error "a real OpenCV is on the include path: this test is about the model"\n#endif\n'
                   'static_assert(std::is_same<fa::Mat1f, cv::Mat1f>::value, "fa::Mat1f is cv::Mat1f");\n'
                   'static_assert(std::is_same<fa::Point, cv::Point>::value, "fa::Point is cv::Point");\n'
                   'static_assert(std::is_same<decltype(fa::Mat1f().data), unsigned char*>::value, "cv::Mat::data is uchar*");\n'
                   'int main() { return 0; }\n')
    r = subprocess.run(["g++"] + FLAGS + ["-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("unit", ["cvsteer_amd/facade/facade.cpp", "tests/cpp/test_basic.cpp", "tests/cpp/test_protected.cpp", "tests/cpp/test_subclass.cpp"])
def test_opencv_branch_compiles_without_warnings(unit):
    """the facade and the callers' code (the reference's gtest body, a subclass written with the reference's protected member
    names) with cv::Mat1f as the matrix type: -Wall -Wextra -Werror"""
    r = subprocess.run(["g++"] + FLAGS + ["-fPIC", "-fsyntax-only", os.path.join(ROOT, unit)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_opencv_branch_runs_the_reference_test_body_and_equals_the_stand_in_build(golden_dir, tmp_path):
    facade = os.path.join(ROOT, "cvsteer_amd", "facade")
    subprocess.check_call(["make", "-C", facade, "-s"])
    subprocess.check_call(["make", "-C", facade, "-s", "model"])
    outs = {}
    for tag in ("", "_cvmodel"):
        d = tmp_path / ("out" + tag)
        d.mkdir()
        r = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_basic" + tag), os.path.join(golden_dir, "fish_u8.npy"), str(d)],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "cvsteer.basic OK" in r.stdout, (tag, r.stdout + r.stderr)
        outs[tag] = {f: np.fromfile(str(d / f), np.float32) for f in sorted(os.listdir(str(d))) if f.endswith(".f32")}
    assert outs[""].keys() == outs["_cvmodel"].keys() and len(outs[""]) >= 5
    for k in outs[""]:
        assert np.array_equal(outs[""][k], outs["_cvmodel"][k], equal_nan=True), k
    for name, ok in (("test_protected_cvmodel", "cvsteer.protected OK"), ("test_subclass_cvmodel", "cvsteer.subclass OK")):
        r = subprocess.run([os.path.join(ROOT, "tests", "cpp", name)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and ok in r.stdout, (name, r.stdout + r.stderr)
