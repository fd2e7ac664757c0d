"""GPU test (-m gpu) of the hand-counted `s_waitcnt vmcnt(N)` of the strip kernels' input path (cvs_kernels_basis.hip, dma_row / row_step).

The product's correctness rests on vmcnt arithmetic done by hand per template variant (missing item 4).  The canary twins
(`make -C cvsteer_amd/csrc canary`; __graft_entry__.build() builds them) check it DIRECTLY: every ring line is filled with a pattern no image
contains before the load that refills it is issued, and a lane that still reads the pattern behind the wait that is supposed to cover the
row is counted; the stores of every output row are tallied against S_ROW, the compile-time lower bound the counts are built from.
tools/canary_run.py drives every kind of launch bench.py times at full size (4096^2, 8192^2, the 5-level pyramid, 32 x 1080p batches, G4,
8-bit, outputs-only with several masks, three launch orders, new and resident images) and a random mix of small shapes and options.
  * libcvsteer_hip_canary.so        -- the product's counts: 0 stale words, 0 short rows
  * libcvsteer_hip_canary_slack.so  -- counts 6 too high: MUST be caught (the test would otherwise prove nothing)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(twin, quick):
    path = os.path.join(ROOT, "tools", twin)
    assert os.path.exists(path), "%s missing: run `make -C cvsteer_amd/csrc canary` (or __graft_entry__.build())" % path
    cmd = [sys.executable, os.path.join(ROOT, "tools", "canary_run.py"), path] + (["--quick"] if quick else [])
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_hand_counted_waits_never_let_a_lane_read_a_row_that_has_not_landed():
    r = _run("libcvsteer_hip_canary.so", quick=False)
    assert r["reads"] > 10_000_000 and r["rows"] > 5_000_000, r      # the checks ran, at full size
    for name, c in r["sections"].items():
        assert c[2] > 0, (name, c)
        assert c[0] == 0, "stale ring-line words read in %s: %r" % (name, c)
        assert c[1] == 0, "output rows with fewer stores than S_ROW in %s: %r" % (name, c)


def test_a_count_that_is_too_high_is_caught():
    r = _run("libcvsteer_hip_canary_slack.so", quick=True)
    assert r["stale"] > 0, r      # waits that let six more operations stay in flight read lines that are still being written
    assert r["short_rows"] == 0, r
