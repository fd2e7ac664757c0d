"""CPU test: the host side under AddressSanitizer + UndefinedBehaviorSanitizer (the reference's CI runs its gtest under ASan and
LSan, /root/reference .travis.yml:48-51).  Builds the two harnesses of tests/cpp with -fsanitize=address,undefined
and runs them: the batch driver's file readers on mutated PGM / .npy files, and the device-free host logic of libcvsteer_hip.so
(argument checks, overlap rules, CVS_OPTS parser, state layouts, taps).  tools/run_sanitizers.sh is the full campaign (it also runs
this CPU suite against instrumented twins of the libraries); its log is profiles/r06_asan_cpu.txt."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]


def _run(cmd, **kw):
    return subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, **kw)


def _clean(r):
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-1500:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-1500:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_file_readers_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "fuzz_readers")
    r = _run(["g++", "-std=c++11"] + SAN + ["-Iinclude", "tests/cpp/fuzz_readers.cpp", "-o", exe])
    assert r.returncode == 0, r.stderr[-2000:]
    r = _run([exe, "3000"], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    _clean(r)
    assert "no sanitizer report" in r.stdout


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.exists("/opt/rocm/lib/libamdhip64.so"), reason="needs g++ and the HIP runtime library to link against")
def test_device_free_host_logic_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_logic_san")
    src = ["tests/cpp/host_logic_san.cpp"] + ["cvsteer_amd/csrc/%s.cpp" % n for n in ("cvs_handle", "cvs_tune", "cvs_state", "cvs_taps")]
    r = _run(["g++", "-std=c++17"] + SAN + ["-Iinclude", "-Icvsteer_amd/csrc", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"] + src +
             ["-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    assert r.returncode == 0, r.stderr[-2000:]
    r = _run([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    _clean(r)
    assert "no sanitizer report" in r.stdout


def test_the_committed_campaign_log_is_clean():
    log = open(os.path.join(ROOT, "profiles", "r06_asan_cpu.txt")).read()
    assert "RESULT: clean" in log and "pytest: rc 0" in log and "fuzz_readers: rc 0" in log and "host_logic_san: rc 0" in log
