/*
 * oracle_taps.c -- tap generation restated (TEST INFRASTRUCTURE, see cvsteer_oracle.h).
 *
 * Follows reference cvsteer/SteerableFilters.cpp:33-42 (create: k[i+w] = f(float(i)*spacing))
 * and the tap formulas at SteerableFiltersG2.cpp:35-42 / SteerableFiltersG4.cpp:34-45.
 *
 * The reference evaluates these in *mixed* precision: x is float, products that involve a
 * double literal are double, pure x*x*... chains and -x*x stay float, exp(float) resolves
 * to the float overload, and the result narrows to float once on return.  Every cast below
 * spells that promotion out.  Build with -ffp-contract=off (see Makefile) so no FMA fuses.
 * Pinned bit-exactly by tests/golden/taps_ref.json (the reference's own functions compiled
 * in place by oracle/ref_taps.mk).
 */
#include <math.h>
#include "cvsteer_oracle.h"

typedef float (*tapfn)(float);

static float gauss(float x) { return expf(-x * x); } /* G22, H22, G42, H42 */

/* ---- G2 / H2 (SteerableFiltersG2.cpp:35-42) ---- */
static float t_g21(float x) { return (float)(0.9213 * (2.0 * (double)x * (double)x - 1.0) * (double)gauss(x)); }
static float t_g22(float x) { return gauss(x); }
static float t_g23(float x) { return (float)(sqrt(1.8430) * (double)x * (double)gauss(x)); }
static float t_h21(float x) { return (float)(0.9780 * (-2.254 * (double)x + (double)(x * x * x)) * (double)gauss(x)); }
static float t_h22(float x) { return gauss(x); }
static float t_h23(float x) { return x * gauss(x); }
static float t_h24(float x) { return (float)(0.9780 * (-0.7515 + (double)(x * x)) * (double)gauss(x)); }

/* ---- G4 / H4 (SteerableFiltersG4.cpp:34-45) ---- */
static float t_g41(float x) { return (float)(1.246 * (0.75 - (double)(3.0f * x * x) + (double)(x * x * x * x)) * (double)gauss(x)); }
static float t_g42(float x) { return gauss(x); }
static float t_g43(float x) { return (float)((-1.5 * (double)x + (double)(x * x * x)) * (double)gauss(x)); }
static float t_g44(float x) { return (float)(1.246 * (double)x * (double)gauss(x)); }
static float t_g45(float x) { return (float)(sqrt(1.246) * ((double)(x * x) - 0.5) * (double)gauss(x)); }
static float t_h41(float x) { return (float)(0.3975 * (7.189 * (double)x - 7.501 * (double)x * (double)x * (double)x + (double)(x * x * x * x * x)) * (double)gauss(x)); }
static float t_h42(float x) { return gauss(x); }
static float t_h43(float x) { return (float)(0.3975 * (1.438 - 4.501 * (double)x * (double)x + (double)(x * x * x * x)) * (double)gauss(x)); }
static float t_h44(float x) { return x * gauss(x); }
static float t_h45(float x) { return (float)(0.3975 * ((double)(x * x * x) - 2.225 * (double)x) * (double)gauss(x)); }
static float t_h46(float x) { return (float)(((double)(x * x) - 0.6638) * (double)gauss(x)); }

static const tapfn G2_FNS[7] = { t_g21, t_g22, t_g23, t_h21, t_h22, t_h23, t_h24 };
static const tapfn G4_FNS[11] = { t_g41, t_g42, t_g43, t_g44, t_g45, t_h41, t_h42, t_h43, t_h44, t_h45, t_h46 };

/* basis plane p = sepFilter2D(image, kx = taps[KX[p]], ky = taps[KY[p]])
 * G2: SteerableFiltersG2.cpp:62-68 ; member index m_g1..m_g3 = 0..2, m_h1..m_h4 = 3..6 */
static const int G2_KX[7] = { 0, 2, 1, 3, 6, 5, 4 };
static const int G2_KY[7] = { 1, 2, 0, 4, 5, 6, 3 };
/* G4: SteerableFiltersG4.cpp:69-80 ; m_g1..m_g5 = 0..4, m_h1..m_h6 = 5..10 */
static const int G4_KX[11] = { 0, 2, 4, 3, 1, 5, 7, 9, 10, 8, 6 };
static const int G4_KY[11] = { 1, 3, 4, 2, 0, 6, 8, 10, 9, 7, 5 };

int ora_num_filters(int kind) { return kind == ORA_KIND_G2 ? 7 : kind == ORA_KIND_G4 ? 11 : 0; }

int ora_make_taps(int kind, int idx, int width, float spacing, float* out)
{
    int n = ora_num_filters(kind);
    if (n == 0 || idx < 0 || idx >= n || width < 0 || !out) return -1;
    tapfn f = (kind == ORA_KIND_G2) ? G2_FNS[idx] : G4_FNS[idx];
    for (int i = -width; i <= width; i++) out[i + width] = f((float)i * spacing);
    return 0;
}

int ora_basis_pair(int kind, int p, int* kx_idx, int* ky_idx)
{
    int n = ora_num_filters(kind);
    if (n == 0 || p < 0 || p >= n) return -1;
    if (kind == ORA_KIND_G2) { *kx_idx = G2_KX[p]; *ky_idx = G2_KY[p]; }
    else { *kx_idx = G4_KX[p]; *ky_idx = G4_KY[p]; }
    return 0;
}
