"""CPU tests of bench.py's host-side helpers (no GPU): the median, the telemetry lookup, the guard of the live counter passes."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_median(bench):
    assert bench._median([3.0, 1.0, 2.0]) == 2.0
    assert bench._median([4.0, 1.0, 3.0, 2.0]) == 2.5
    assert bench._median([7.0]) == 7.0


def test_telemetry_lookup_without_a_device_is_none(bench):
    """no HIP device (or a card whose hwmon files cannot be matched by PCI address): no telemetry block, no exception"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    assert bench._telemetry_files(torch, 0) is None


def test_live_counter_passes_refuse_to_nest_under_a_profiler(bench, monkeypatch):
    """bench.py under rocprofv3 (tools/profile.sh) must not start rocprofv3 children of its own"""
    monkeypatch.setenv("ROCPROFILER_TEST_MARK", "1")
    got, why = bench._live_traffic()
    assert got is None and "profiler" in why
    monkeypatch.delenv("ROCPROFILER_TEST_MARK")
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    got, why = bench._live_traffic()
    assert got is None and "profiler" in why


def test_bytes_per_pixel_table_matches_the_survey(bench):
    """SURVEY.md 8(d) / DESIGN.md 3: algorithmic bytes per pixel of the timed legs"""
    b = bench.BYTES_PER_PIX
    assert (b["M1"], b["M2"], b["M4"], b["M5"], b["M6"], b["M6s"]) == (32, 40, 52, 84, 48, 56)
    assert b["M2_u8"] == 37 and b["C4_u8_feat3"] == 13
