#!/usr/bin/env python3
"""tools/gen_g4_orient.py -- derive the oriented-energy coefficient tables from the steering polynomials.

E(theta) = g(theta)^2 + h(theta)^2 with g = sum_i kg_i(theta) G_i, h = sum_i kh_i(theta) H_i.  The mean, the
cos(2 theta) and the sin(2 theta) Fourier coefficients of every product k_i k_j give C1, C2, C3.  Run with
`g2` to see the procedure reproduce the reference's constants (SteerableFiltersG2.cpp:93-95), with `g4`
for the table used by the G4 extension (cvs_device_math.h kG4Terms, oracle_filter.c ORA_G4_TERMS)."""
import sys
from fractions import Fraction
import numpy as np

N = 4096
th = np.arange(N) * 2 * np.pi / N
c, s = np.cos(th), np.sin(th)
BANKS = {
    "g2": ([c**2, -2*c*s, s**2], [c**3, -3*c**2*s, 3*c*s**2, -s**3]),
    "g4": ([c**4, -4*c**3*s, 6*c**2*s**2, -4*c*s**3, s**4], [c**5, -5*c**4*s, 10*c**3*s**2, -10*c**2*s**3, 5*c*s**4, -s**5]),
}

def terms(k, off):
    out = []
    for i in range(len(k)):
        for j in range(i, len(k)):
            m = 1 if i == j else 2
            p = k[i] * k[j]
            for which, w in ((1, np.ones(N)), (2, 2 * np.cos(2 * th)), (3, 2 * np.sin(2 * th))):
                v = m * (p * w).mean()
                if abs(v) > 1e-12:
                    out.append((off + i, off + j, which, Fraction(v).limit_denominator(1 << 12)))
    return out

kind = sys.argv[1] if len(sys.argv) > 1 else "g4"
kg, kh = BANKS[kind]
for i, j, which, fr in terms(kg, 0) + terms(kh, len(kg)):
    print("{%d, %d, %d, %d.f / %d}," % (i, j, which, fr.numerator, fr.denominator))
