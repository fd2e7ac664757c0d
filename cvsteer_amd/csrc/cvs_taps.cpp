// cvs_taps.cpp -- host-side scalar math of the boundary: 1-D tap vectors and scalar steering
// weights.  No HIP; usable (and tested) without a GPU.
//
// Tap vectors: SteerableFilters::create samples k[i+w] = f(float(i)*spacing)
// (reference cvsteer/SteerableFilters.cpp:33-42) for the Freeman-Adelson separable basis
// functions of SteerableFiltersG2.cpp:35-42 and SteerableFiltersG4.cpp:34-45.  To be a
// drop-in, the taps must come out bit-identical to the reference's, and the reference
// evaluates them in mixed precision: the argument is float, any product with a double
// literal is double, pure powers of x and -x*x stay float, exp() is the float overload, and
// the value narrows to float once.  Each basis function is described here as
//       value = float( lead * poly(x) * double(expf(-x*x)) )
// with poly a short list of terms whose precision class is explicit; pure-float functions
// (x^n * gaussian with no double literal) are flagged so they never leave float.
// Verified bit-for-bit against the reference's own functions: tests/golden/taps_ref.json.
#include <cmath>

#include "cvs_internal.h"

namespace cvs {
namespace {

enum TermKind {
    T_CONST,        // double constant c
    T_DBL_CHAIN,    // (((c * x) * x) ...) with x promoted to double at every step, `pow` factors of x
    T_FLT_POW,      // x*x*...*x in float (`pow` factors), optionally pre-multiplied by float literal cf, then promoted
};

struct Term {
    TermKind kind;
    double c;    // constant / double-chain coefficient / sign (+1/-1) for T_FLT_POW
    float cf;    // float literal leading a float chain (0 = none)
    int pow;
};

struct Basis1D {
    bool pure_float;   // value = x^pow0 * expf(-x*x) entirely in float (pow0 in terms[0].pow)
    double lead;       // leading double factor (1.0 = none; multiplication by it is skipped when `has_lead` is false)
    bool has_lead;
    bool lead_times_x; // lead * x * gauss form (no parenthesised polynomial): ((lead * x) * gauss)
    int nterms;
    Term terms[3];
};

inline float gaussian(float x) { return std::exp(-x * x); }

float flt_pow(float x, int n)
{
    float v = x;
    for (int i = 1; i < n; ++i) v = v * x;
    return v;
}

double eval_term(const Term& t, float x)
{
    switch (t.kind) {
        case T_CONST: return t.c;
        case T_DBL_CHAIN: {
            double v = t.c;
            for (int i = 0; i < t.pow; ++i) v = v * (double)x;
            return v;
        }
        case T_FLT_POW: {
            float v = t.cf != 0.0f ? t.cf * x : x;
            for (int i = 1; i < t.pow; ++i) v = v * x;
            return t.c * (double)v;  // c is +1/-1: exact
        }
    }
    return 0.0;
}

float eval(const Basis1D& b, float x)
{
    const float g = gaussian(x);
    if (b.pure_float) return b.terms[0].pow == 0 ? g : flt_pow(x, b.terms[0].pow) * g;
    if (b.lead_times_x) return (float)(b.lead * (double)x * (double)g);
    double poly = eval_term(b.terms[0], x);
    for (int i = 1; i < b.nterms; ++i) poly = poly + eval_term(b.terms[i], x);
    const double v = b.has_lead ? b.lead * poly : poly;
    return (float)(v * (double)g);
}

const double kSqrt1843 = std::sqrt(1.8430);
const double kSqrt1246 = std::sqrt(1.246);

// member order m_g1,m_g2,m_g3,m_h1,m_h2,m_h3,m_h4  (SteerableFiltersG2.cpp:47-55)
const Basis1D kG2[7] = {
    /* G21 = 0.9213 (2x^2 - 1) e */ {false, 0.9213, true, false, 2, {{T_DBL_CHAIN, 2.0, 0, 2}, {T_CONST, -1.0, 0, 0}}},
    /* G22 = e                   */ {true, 1, false, false, 1, {{T_FLT_POW, 1, 0, 0}}},
    /* G23 = sqrt(1.843) x e     */ {false, kSqrt1843, true, true, 0, {}},
    /* H21 = 0.978(-2.254x+x^3)e */ {false, 0.9780, true, false, 2, {{T_DBL_CHAIN, -2.254, 0, 1}, {T_FLT_POW, 1, 0, 3}}},
    /* H22 = e                   */ {true, 1, false, false, 1, {{T_FLT_POW, 1, 0, 0}}},
    /* H23 = x e                 */ {true, 1, false, false, 1, {{T_FLT_POW, 1, 0, 1}}},
    /* H24 = 0.978(-0.7515+x^2)e */ {false, 0.9780, true, false, 2, {{T_CONST, -0.7515, 0, 0}, {T_FLT_POW, 1, 0, 2}}},
};

// member order m_g1..m_g5, m_h1..m_h6  (SteerableFiltersG4.cpp:50-62)
const Basis1D kG4[11] = {
    /* G41 = 1.246(0.75-3x^2+x^4)e */ {false, 1.246, true, false, 3, {{T_CONST, 0.75, 0, 0}, {T_FLT_POW, -1, 3.0f, 2}, {T_FLT_POW, 1, 0, 4}}},
    /* G42 = e                     */ {true, 1, false, false, 1, {{T_FLT_POW, 1, 0, 0}}},
    /* G43 = (-1.5x + x^3) e       */ {false, 1, false, false, 2, {{T_DBL_CHAIN, -1.5, 0, 1}, {T_FLT_POW, 1, 0, 3}}},
    /* G44 = 1.246 x e             */ {false, 1.246, true, true, 0, {}},
    /* G45 = sqrt(1.246)(x^2-.5)e  */ {false, kSqrt1246, true, false, 2, {{T_FLT_POW, 1, 0, 2}, {T_CONST, -0.5, 0, 0}}},
    /* H41 = .3975(7.189x-7.501x^3+x^5)e */ {false, 0.3975, true, false, 3, {{T_DBL_CHAIN, 7.189, 0, 1}, {T_DBL_CHAIN, -7.501, 0, 3}, {T_FLT_POW, 1, 0, 5}}},
    /* H42 = e                     */ {true, 1, false, false, 1, {{T_FLT_POW, 1, 0, 0}}},
    /* H43 = .3975(1.438-4.501x^2+x^4)e */ {false, 0.3975, true, false, 3, {{T_CONST, 1.438, 0, 0}, {T_DBL_CHAIN, -4.501, 0, 2}, {T_FLT_POW, 1, 0, 4}}},
    /* H44 = x e                   */ {true, 1, false, false, 1, {{T_FLT_POW, 1, 0, 1}}},
    /* H45 = .3975(x^3 - 2.225x) e */ {false, 0.3975, true, false, 2, {{T_FLT_POW, 1, 0, 3}, {T_DBL_CHAIN, -2.225, 0, 1}}},
    /* H46 = (x^2 - 0.6638) e      */ {false, 1, false, false, 2, {{T_FLT_POW, 1, 0, 2}, {T_CONST, -0.6638, 0, 0}}},
};

// basis plane p = sepFilter2D(image, kx = taps[KX[p]], ky = taps[KY[p]])
const int kG2KX[7] = {0, 2, 1, 3, 6, 5, 4}, kG2KY[7] = {1, 2, 0, 4, 5, 6, 3};          // G2.cpp:62-68
const int kG4KX[11] = {0, 2, 4, 3, 1, 5, 7, 9, 10, 8, 6}, kG4KY[11] = {1, 3, 4, 2, 0, 6, 8, 10, 9, 7, 5};  // G4.cpp:69-80

}  // namespace

int host_num_basis(int kind) { return kind == 2 ? 7 : kind == 4 ? 11 : 0; }

int host_make_taps(int kind, int idx, int width, float spacing, float* out)
{
    const int n = host_num_basis(kind);
    if (n == 0 || idx < 0 || idx >= n || width < 0 || width > kMaxWidth || !out) return -1;
    const Basis1D& b = kind == 2 ? kG2[idx] : kG4[idx];
    for (int i = -width; i <= width; ++i) out[i + width] = eval(b, float(i) * spacing);
    return 0;
}

int host_basis_taps(int kind, int p, int* kx, int* ky)
{
    const int n = host_num_basis(kind);
    if (n == 0 || p < 0 || p >= n || !kx || !ky) return -1;
    *kx = kind == 2 ? kG2KX[p] : kG4KX[p];
    *ky = kind == 2 ? kG2KY[p] : kG4KY[p];
    return 0;
}

// G2.cpp:140-142 / G4.cpp:116-119: float cos/sin and float powers; the weights that carry a
// double literal (-2.0, 3.0, -4.0, 6.0, 10.0, 5.0) are double products narrowed once;
// -5.0f * ct4 * st stays float.
int host_steer_weights(int kind, float theta, float* out)
{
    if (!out) return -1;
    const float ct = std::cos(theta), st = std::sin(theta);
    const float ct2 = ct * ct, ct3 = ct2 * ct, st2 = st * st, st3 = st2 * st;
    if (kind == 2) {
        out[0] = ct2;
        out[1] = (float)(-2.0 * (double)ct * (double)st);
        out[2] = st2;
        out[3] = ct3;
        out[4] = (float)(-3.0 * (double)ct2 * (double)st);
        out[5] = (float)(3.0 * (double)ct * (double)st2);
        out[6] = -st3;
        return 0;
    }
    if (kind == 4) {
        const float ct4 = ct3 * ct, ct5 = ct4 * ct, st4 = st3 * st, st5 = st4 * st;
        out[0] = ct4;
        out[1] = (float)(-4.0 * (double)ct3 * (double)st);
        out[2] = (float)(6.0 * (double)ct2 * (double)st2);
        out[3] = (float)(-4.0 * (double)ct * (double)st3);
        out[4] = st4;
        out[5] = ct5;
        out[6] = -5.0f * ct4 * st;
        out[7] = (float)(10.0 * (double)ct3 * (double)st2);
        out[8] = (float)(-10.0 * (double)ct2 * (double)st3);
        out[9] = (float)(5.0 * (double)ct * (double)st4);
        out[10] = -st5;
        return 0;
    }
    return -1;
}

}  // namespace cvs
