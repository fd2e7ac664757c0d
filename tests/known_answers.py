"""Known answers that come from the REFERENCE, not from the oracle: the impulse response of cv::sepFilter2D (correlation,
anchor at the centre: dst(y, x) = sum_j sum_i ky[j + w] kx[i + w] src(y + j, x + i)) with the reference's own
tap tables (tests/golden/taps_ref.json = output of the reference's tap functions compiled in place, oracle/ref_taps.mk) and
the kernel pairs of SteerableFiltersG2.cpp:62-68 / SteerableFiltersG4.cpp:69-80 (row kernel, column kernel)."""
import json
import os

import numpy as np

PAIRS = {2: [("G21", "G22"), ("G23", "G23"), ("G22", "G21"), ("H21", "H22"), ("H24", "H23"), ("H23", "H24"), ("H22", "H21")],
         4: [("G41", "G42"), ("G43", "G44"), ("G45", "G45"), ("G44", "G43"), ("G42", "G41"),
             ("H41", "H42"), ("H43", "H44"), ("H45", "H46"), ("H46", "H45"), ("H44", "H43"), ("H42", "H41")]}
DEFAULTS = {2: (4, 0.67), 4: (6, 0.5)}


def reference_taps(golden_dir, kind):
    sec = json.load(open(os.path.join(golden_dir, "taps_ref.json")))["g2" if kind == 2 else "g4"]
    return {k: np.array([int(h, 16) for h in v], np.uint32).view(np.float32) for k, v in sec.items() if isinstance(v, list) and k != "order"}


def impulse_planes(golden_dir, kind, rows, cols, r, c):
    """basis planes of an image that is 1.0 at (r, c) and 0 elsewhere; (r, c) is either the corner (0, 0) -- REFLECT_101 maps
    only index 0 onto index 0, so nothing is added at the border -- or at least `width` away from every border.  Each value is
    ONE f32 product of two reference taps, so the expectation is exact to the bit."""
    taps = reference_taps(golden_dir, kind)
    w = DEFAULTS[kind][0]
    out = np.zeros((len(PAIRS[kind]), rows, cols), np.float32)
    for p, (kxn, kyn) in enumerate(PAIRS[kind]):
        kx, ky = taps[kxn], taps[kyn]
        for y in range(max(0, r - w), min(rows, r + w + 1)):
            for x in range(max(0, c - w), min(cols, c + w + 1)):
                out[p, y, x] = np.float32(ky[r - y + w]) * np.float32(kx[c - x + w])
    return out
