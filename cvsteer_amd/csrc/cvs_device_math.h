// cvs_device_math.h -- per-pixel device helpers shared by the basis and pointwise kernels.
// gfx950 only.  Every helper states which reference / OpenCV step it stands for.
// Both kernel translation units are built with -ffp-contract=off, so each __f*_rn / operator
// below is one separately rounded f32 operation (HIP's __fmul_rn is a plain '*').
#pragma once
#include <hip/hip_runtime.h>

namespace cvs {

constexpr float kPiF = 3.14159274f;      // (float)M_PI      -- wrap threshold, SteerableFilters.cpp:50
constexpr float kTwoPiF = 6.2831855f;    // (float)(2*M_PI)  -- wrap shift,     SteerableFilters.cpp:49
constexpr float kHalfPiF = 1.57079637f;  // (float)M_PI_2    -- phaseWeights gate, G2.cpp:185

// BORDER_REFLECT_101 (sepFilter2D default border), any distance, len >= 1.
__host__ __device__ inline int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    const int period = 2 * len - 2;
    p %= period;
    if (p < 0) p += period;
    return p < len ? p : period - p;
}

// cos/sin for cv::polarToCart (G2.cpp:151,175,183; documented accuracy ~1e-6).
// The angles on this path are bounded -- theta_dom in (-pi/2, pi/2], 2*theta in (-pi, pi], the
// phase error in [0, pi] -- so a two-constant Cody-Waite reduction to |r| <= pi/4 plus the classic
// single-precision minimax polynomials (~1 ulp) is enough and is ~25 instructions, small enough to
// live in the fused basis kernel's unrolled epilogue.  |x| > 8 (only reachable through
// caller-supplied theta maps / phi) falls back to the library routine.
__device__ __forceinline__ void sincos_small(float x, float& s, float& c)
{
    const float k = rintf(__fmul_rn(x, 0.636619772f));            // x * 2/pi
    float r = fmaf(-k, 1.5707963705062866f, x);                    // - k * float(pi/2)
    r = fmaf(-k, -4.371139000186243e-08f, r);                      // - k * (pi/2 - float(pi/2))
    const float z = __fmul_rn(r, r);
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(z, ps, -1.6666654611e-1f);
    const float sr = fmaf(__fmul_rn(ps, z), r, r);
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(z, pc, 4.166664568298827e-2f);
    const float cr = fmaf(__fmul_rn(pc, z), z, fmaf(z, -0.5f, 1.0f));
    const int q = (int)k & 3;
    const float ss = (q & 1) ? cr : sr, cs = (q & 1) ? sr : cr;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cs : cs;
}

__device__ __forceinline__ void sincos_any(float x, float& s, float& c)
{
    if (fabsf(x) <= 8.0f) sincos_small(x, s, c);
    else sincosf(x, &s, &c);
}

__device__ __forceinline__ float cos_any(float x)
{
    float s, c;
    sincos_any(x, s, c);
    return c;
}

// cv::cartToPolar's angle (radians, [0, 2pi)).
// mode 0: the OpenCV 3.4 fastAtan2 polynomial in degrees, every op separately rounded
// (contraction is off), then * (float)(pi/180).  mode 1: atan2f.
__device__ inline float angle_0_2pi(float y, float x, int mode)
{
    if (mode == 0) {
        const float scale = (float)(180.0 / 3.14159265358979323846);
        const float p1 = 0.9997878412794807f * scale;
        const float p3 = -0.3258083974640975f * scale;
        const float p5 = 0.1555786518463281f * scale;
        const float p7 = -0.04432655554792128f * scale;
        const float eps = 2.2204460492503131e-16f;  // (float)DBL_EPSILON
        const float ax = fabsf(x), ay = fabsf(y);
        const bool xge = ax >= ay;  // NaNs take the 'else' arm and stay NaN, like the reference
        const float mn = xge ? ay : ax, mx = xge ? ax : ay;
        const float c = __fdiv_rn(mn, __fadd_rn(mx, eps));
        const float c2 = __fmul_rn(c, c);
        float a = __fadd_rn(__fmul_rn(p7, c2), p5);
        a = __fadd_rn(__fmul_rn(a, c2), p3);
        a = __fadd_rn(__fmul_rn(a, c2), p1);
        a = __fmul_rn(a, c);
        if (!xge) a = __fsub_rn(90.f, a);
        if (x < 0.f) a = __fsub_rn(180.f, a);
        if (y < 0.f) a = __fsub_rn(360.f, a);
        return __fmul_rn(a, (float)(3.14159265358979323846 / 180.0));
    }
    float a = atan2f(y, x);
    if (a < 0.f) a = __fadd_rn(a, kTwoPiF);
    return a;
}

// SteerableFilters::wrap, SteerableFilters.cpp:46-51
__device__ inline float wrap_pi(float a) { return a > kPiF ? __fsub_rn(a, kTwoPiF) : a; }

// computeMagnitudeAndPhase, G2.cpp:107-112 (cartToPolar + wrap + patchNaNs)
__device__ inline void mag_phase(float g, float h, int mode, float& mag, float& phase)
{
    mag = __fsqrt_rn(__fadd_rn(__fmul_rn(g, g), __fmul_rn(h, h)));
    float p = wrap_pi(angle_0_2pi(h, g, mode));
    phase = (p != p) ? 0.f : p;
}

// phaseWeights, G2.cpp:179-186: lambda = cos^2(err) gated at pi/2
// BOUNDED: the caller guarantees |err| <= 8 (phase in (-pi, pi], phi in {0, pi/2, pi})
template <bool BOUNDED = false>
__device__ __forceinline__ float phase_lambda(float phase, float phi, bool signum)
{
    float err = signum ? fabsf(__fsub_rn(phase, phi)) : fabsf(__fsub_rn(fabsf(phase), fabsf(phi)));
    err = fminf(err, __fsub_rn(kTwoPiF, err));
    float st_, ct;
    if constexpr (BOUNDED) sincos_small(err, st_, ct);
    else sincos_any(err, st_, ct);
    (void)st_;
    float l = __fmul_rn(ct, ct);
    if (fabsf(err) > kHalfPiF) l = 0.f;
    return l;
}

// findEdges / findDarkLines / findBrightLines (G2.cpp:194-212) for ONE phase value: the three phaseWeights of the callers'
// sequence (test/test.cpp:88-90) -- phi = pi/2 unsigned, phi = 0 signed, phi = pi signed -- from one cosine / sine pair.
// phaseWeights forms err_phi, folds it with min(err, 2 pi - err), and takes cos^2(err); cos^2 does not see that fold, and
//   phi = pi/2, unsigned:  cos^2(| |p| - pi/2 |) = sin^2(|p|)
//   phi = 0,    signed:    cos^2(| p |)          = cos^2(|p|)     (the very evaluation phaseWeights makes: bit-identical)
//   phi = pi,   signed:    cos^2(| p - pi |)     = cos^2(|p|)
// for every p, so one sincos(|p|) serves all three; the gates (lambda = 0 where the folded error exceeds pi/2) are
// evaluated exactly as phaseWeights evaluates them.  Against the one-at-a-time evaluation the values differ by what the
// reference's own float steps (|p| - float(pi/2), p - float(pi), 2 float(pi) - err) round away: <= 4e-7 for p in
// [-pi, pi], inside the 1e-6 stage tolerance (round-2 verdict item 5; tests/test_gpu_parity.py).  Two of the three
// bounded cos / sin evaluations per pixel go away -- the stateless pipeline launch is VALU-bound.
// BOUNDED: the caller guarantees |phase| <= 8 (the engine's own phase planes lie in (-pi, pi]).
// sin^2 / cos^2 of x >= 0 with sincos_small's reduction and polynomials: the squares do not see the quadrant's signs, so
// only the swap of the two polynomials (odd quadrant) is left of the quadrant logic -- same values as squaring
// sincos_small's results, six instructions fewer
__device__ __forceinline__ void sincos_squares_small(float x, float& s2, float& c2)
{
    const float k = rintf(__fmul_rn(x, 0.636619772f));
    float r = fmaf(-k, 1.5707963705062866f, x);
    r = fmaf(-k, -4.371139000186243e-08f, r);
    const float z = __fmul_rn(r, r);
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(z, ps, -1.6666654611e-1f);
    const float sr = fmaf(__fmul_rn(ps, z), r, r);
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(z, pc, 4.166664568298827e-2f);
    const float cr = fmaf(__fmul_rn(pc, z), z, fmaf(z, -0.5f, 1.0f));
    const bool odd = ((int)k & 1) != 0;
    const float ss = odd ? cr : sr, cs = odd ? sr : cr;
    s2 = __fmul_rn(ss, ss);
    c2 = __fmul_rn(cs, cs);
}

template <bool BOUNDED = false>
__device__ __forceinline__ void phase_lambda3(float phase, float& l_edges, float& l_dark, float& l_bright)
{
    const float ap = fabsf(phase);
    float s2, c2;
    if (BOUNDED || ap <= 8.0f) {
        sincos_squares_small(ap, s2, c2);
    } else {
        float s, c;
        sincosf(ap, &s, &c);
        s2 = __fmul_rn(s, s);
        c2 = __fmul_rn(c, c);
    }
    float ee = fabsf(__fsub_rn(ap, fabsf(kHalfPiF)));       // signum = false: | |phase| - |phi| |
    ee = fminf(ee, __fsub_rn(kTwoPiF, ee));
    const float ed = fminf(ap, __fsub_rn(kTwoPiF, ap));     // signum = true, phi = 0: |phase - 0|
    float eb = fabsf(__fsub_rn(phase, kPiF));               // signum = true, phi = pi
    eb = fminf(eb, __fsub_rn(kTwoPiF, eb));
    l_edges = fabsf(ee) > kHalfPiF ? 0.f : s2;
    l_dark = fabsf(ed) > kHalfPiF ? 0.f : c2;
    l_bright = fabsf(eb) > kHalfPiF ? 0.f : c2;
}

// G2.cpp:70-99: products, C1..C3, cartToPolar, wrap, *0.5.  b = {g2a,g2b,g2c,h2a,h2b,h2c,h2d}.
// need_c1 = false skips C1 (only the oriented energy e uses it); callers that store C1 pass true
__device__ inline void g2_orientation(const float b[7], int mode, float& c1, float& c2, float& c3,
                                      float& theta, float& strength, bool need_c1 = true)
{
    const float A = b[0], B = b[1], C = b[2], HA = b[3], HB = b[4], HC = b[5], HD = b[6];
    const float g2aa = __fmul_rn(A, A), g2ab = __fmul_rn(A, B), g2ac = __fmul_rn(A, C);
    const float g2bb = __fmul_rn(B, B), g2bc = __fmul_rn(B, C), g2cc = __fmul_rn(C, C);
    const float h2aa = __fmul_rn(HA, HA), h2ab = __fmul_rn(HA, HB), h2ac = __fmul_rn(HA, HC), h2ad = __fmul_rn(HA, HD);
    const float h2bb = __fmul_rn(HB, HB), h2bc = __fmul_rn(HB, HC), h2bd = __fmul_rn(HB, HD);
    const float h2cc = __fmul_rn(HC, HC), h2cd = __fmul_rn(HC, HD), h2dd = __fmul_rn(HD, HD);
    // every MatExpr node of G2.cpp:93-95 rounds to f32; scalars distribute over (a +/- b)
    float v1 = 0.f;
    if (need_c1) {
        v1 = __fadd_rn(__fmul_rn(0.5f, g2bb), __fmul_rn(0.25f, g2ac));
        v1 = __fadd_rn(v1, __fadd_rn(__fmul_rn(0.375f, g2aa), __fmul_rn(0.375f, g2cc)));
        v1 = __fadd_rn(v1, __fadd_rn(__fmul_rn(0.3125f, h2aa), __fmul_rn(0.3125f, h2dd)));
        v1 = __fadd_rn(v1, __fadd_rn(__fmul_rn(0.5625f, h2bb), __fmul_rn(0.5625f, h2cc)));
        v1 = __fadd_rn(v1, __fadd_rn(__fmul_rn(0.375f, h2ac), __fmul_rn(0.375f, h2bd)));
    }
    float v2 = __fsub_rn(__fmul_rn(0.5f, g2aa), __fmul_rn(0.5f, g2cc));
    v2 = __fadd_rn(v2, __fsub_rn(__fmul_rn(0.46875f, h2aa), __fmul_rn(0.46875f, h2dd)));
    v2 = __fadd_rn(v2, __fsub_rn(__fmul_rn(0.28125f, h2bb), __fmul_rn(0.28125f, h2cc)));
    v2 = __fadd_rn(v2, __fsub_rn(__fmul_rn(0.1875f, h2ac), __fmul_rn(0.1875f, h2bd)));
    float v3 = __fsub_rn(-g2ab, g2bc);
    v3 = __fsub_rn(v3, __fadd_rn(__fmul_rn(0.9375f, h2cd), __fmul_rn(0.9375f, h2ab)));
    v3 = __fsub_rn(v3, __fmul_rn(1.6875f, h2bc));
    v3 = __fsub_rn(v3, __fmul_rn(0.1875f, h2ad));
    c1 = v1; c2 = v2; c3 = v3;
    strength = __fsqrt_rn(__fadd_rn(__fmul_rn(v2, v2), __fmul_rn(v3, v3)));
    theta = __fmul_rn(wrap_pi(angle_0_2pi(v3, v2, mode)), 0.5f);
}

// ---- EXTENSION beyond the reference (opt-in through CVS_OPT_G4_EXTENSIONS) ----
// The reference never computes G4/H4 orientation (G4.h:55 members unused).  The coefficients follow
// from its own steering polynomials (G4.cpp:116-119) exactly as G2.cpp:93-95 follow from G2's:
//   E(theta) = g4(theta)^2 + h4(theta)^2 = C1 + C2 cos 2theta + C3 sin 2theta + [4..10 theta terms],
// with g4 = sum kg_i(theta) G_i, h4 = sum kh_i(theta) H_i; the table holds the mean, the cos 2theta and
// the sin 2theta Fourier coefficients of every product kg_i kg_j / kh_i kh_j (tools/gen_g4_orient.py;
// the same procedure reproduces the reference's G2 constants 0.5, 0.25, 0.375, ...).  All entries are
// dyadic rationals, exact in f32.  theta = wrap(atan2(C3, C2)) / 2 and strength = |(C2, C3)|, the
// convention of G2.cpp:97-99.  b = {g4a..g4e, h4a..h4f}.
struct G4Term { signed char i, j, which; float k; };  // which: 1 -> C1, 2 -> C2, 3 -> C3
__device__ __constant__ const G4Term kG4Terms[] = {
    // G part (plane indices 0..4)
    {0, 0, 1, 35.f / 128}, {0, 0, 2, 7.f / 16}, {0, 1, 3, -7.f / 8}, {0, 2, 1, 15.f / 32}, {0, 2, 2, 3.f / 8},
    {0, 3, 3, -3.f / 8}, {0, 4, 1, 3.f / 64}, {1, 1, 1, 5.f / 8}, {1, 1, 2, 1.f / 2}, {1, 2, 3, -9.f / 4},
    {1, 3, 1, 3.f / 4}, {1, 4, 3, -3.f / 8}, {2, 2, 1, 27.f / 32}, {2, 3, 3, -9.f / 4}, {2, 4, 1, 15.f / 32},
    {2, 4, 2, -3.f / 8}, {3, 3, 1, 5.f / 8}, {3, 3, 2, -1.f / 2}, {3, 4, 3, -7.f / 8}, {4, 4, 1, 35.f / 128},
    {4, 4, 2, -7.f / 16},
    // H part (plane indices 5..10)
    {5, 5, 1, 63.f / 256}, {5, 5, 2, 105.f / 256}, {5, 6, 3, -105.f / 128}, {5, 7, 1, 35.f / 64}, {5, 7, 2, 35.f / 64},
    {5, 8, 3, -35.f / 64}, {5, 9, 1, 15.f / 128}, {5, 9, 2, 5.f / 128}, {5, 10, 3, -5.f / 128}, {6, 6, 1, 175.f / 256},
    {6, 6, 2, 175.f / 256}, {6, 7, 3, -175.f / 64}, {6, 8, 1, 75.f / 64}, {6, 8, 2, 25.f / 64}, {6, 9, 3, -125.f / 128},
    {6, 10, 1, 15.f / 128}, {6, 10, 2, -5.f / 128}, {7, 7, 1, 75.f / 64}, {7, 7, 2, 25.f / 64}, {7, 8, 3, -125.f / 32},
    {7, 9, 1, 75.f / 64}, {7, 9, 2, -25.f / 64}, {7, 10, 3, -35.f / 64}, {8, 8, 1, 75.f / 64}, {8, 8, 2, -25.f / 64},
    {8, 9, 3, -175.f / 64}, {8, 10, 1, 35.f / 64}, {8, 10, 2, -35.f / 64}, {9, 9, 1, 175.f / 256}, {9, 9, 2, -175.f / 256},
    {9, 10, 3, -105.f / 128}, {10, 10, 1, 63.f / 256}, {10, 10, 2, -105.f / 256},
};

__device__ inline void g4_orientation(const float b[11], int mode, float& c1, float& c2, float& c3,
                                      float& theta, float& strength)
{
    float v1 = 0.f, v2 = 0.f, v3 = 0.f;
    constexpr int n = sizeof(kG4Terms) / sizeof(kG4Terms[0]);
#pragma unroll
    for (int t = 0; t < n; ++t) {  // table order, every product and sum rounded to f32
        const float term = __fmul_rn(kG4Terms[t].k, __fmul_rn(b[kG4Terms[t].i], b[kG4Terms[t].j]));
        if (kG4Terms[t].which == 1) v1 = __fadd_rn(v1, term);
        else if (kG4Terms[t].which == 2) v2 = __fadd_rn(v2, term);
        else v3 = __fadd_rn(v3, term);
    }
    c1 = v1; c2 = v2; c3 = v3;
    strength = __fsqrt_rn(__fadd_rn(__fmul_rn(v2, v2), __fmul_rn(v3, v3)));
    theta = __fmul_rn(wrap_pi(angle_0_2pi(v3, v2, mode)), 0.5f);
}

// steer(float theta, ...) G2.cpp:143-144: (ga*A + gb*B) + gc*C, every node rounded
__device__ inline void g2_steer_weights(const float b[7], const float w[7], float& g, float& h)
{
    g = __fadd_rn(__fadd_rn(__fmul_rn(w[0], b[0]), __fmul_rn(w[1], b[1])), __fmul_rn(w[2], b[2]));
    h = __fadd_rn(__fmul_rn(w[3], b[3]), __fmul_rn(w[4], b[4]));
    h = __fadd_rn(h, __fmul_rn(w[5], b[5]));
    h = __fadd_rn(h, __fmul_rn(w[6], b[6]));
}

// steer(const Mat1f& theta, ...) G2.cpp:147-155: cos/sin per pixel, (scale*(a*b))*M products
template <bool BOUNDED = false>
__device__ __forceinline__ void g2_steer_angle(const float b[7], float theta, float& g, float& h)
{
    float st, ct;
    if constexpr (BOUNDED) sincos_small(theta, st, ct);
    else sincos_any(theta, st, ct);
    const float ct2 = __fmul_rn(ct, ct), ct3 = __fmul_rn(ct2, ct);
    const float st2 = __fmul_rn(st, st), st3 = __fmul_rn(st2, st);
    g = __fmul_rn(ct2, b[0]);
    g = __fadd_rn(g, __fmul_rn(__fmul_rn(-2.0f, __fmul_rn(ct, st)), b[1]));
    g = __fadd_rn(g, __fmul_rn(st2, b[2]));
    h = __fmul_rn(ct3, b[3]);
    h = __fadd_rn(h, __fmul_rn(__fmul_rn(-3.0f, __fmul_rn(ct2, st)), b[4]));
    h = __fadd_rn(h, __fmul_rn(__fmul_rn(3.0f, __fmul_rn(ct, st2)), b[5]));
    h = __fadd_rn(h, __fmul_rn(-st3, b[6]));
}

// G4.cpp:120-121: sums left to right
__device__ inline void g4_steer_weights(const float b[11], const float w[11], float& g, float& h)
{
    g = __fadd_rn(__fmul_rn(w[0], b[0]), __fmul_rn(w[1], b[1]));
#pragma unroll
    for (int p = 2; p < 5; p++) g = __fadd_rn(g, __fmul_rn(w[p], b[p]));
    h = __fadd_rn(__fmul_rn(w[5], b[5]), __fmul_rn(w[6], b[6]));
#pragma unroll
    for (int p = 7; p < 11; p++) h = __fadd_rn(h, __fmul_rn(w[p], b[p]));
}

// G4.cpp:92-112: weight planes from cos/sin powers, then weight.mul(basis) summed l-to-r
__device__ inline void g4_steer_angle(const float b[11], float theta, float& g, float& h)
{
    float st, ct;
    sincos_any(theta, st, ct);
    const float ct2 = __fmul_rn(ct, ct), ct3 = __fmul_rn(ct2, ct), ct4 = __fmul_rn(ct3, ct), ct5 = __fmul_rn(ct4, ct);
    const float st2 = __fmul_rn(st, st), st3 = __fmul_rn(st2, st), st4 = __fmul_rn(st3, st), st5 = __fmul_rn(st4, st);
    const float w[11] = { ct4, __fmul_rn(-4.0f, __fmul_rn(ct3, st)), __fmul_rn(6.0f, __fmul_rn(ct2, st2)),
                          __fmul_rn(-4.0f, __fmul_rn(ct, st3)), st4,
                          ct5, __fmul_rn(-5.0f, __fmul_rn(ct4, st)), __fmul_rn(10.0f, __fmul_rn(ct3, st2)),
                          __fmul_rn(-10.0f, __fmul_rn(ct2, st3)), __fmul_rn(5.0f, __fmul_rn(ct, st4)), -st5 };
    g = __fmul_rn(w[0], b[0]);
#pragma unroll
    for (int p = 1; p < 5; p++) g = __fadd_rn(g, __fmul_rn(w[p], b[p]));
    h = __fmul_rn(w[5], b[5]);
#pragma unroll
    for (int p = 6; p < 11; p++) h = __fadd_rn(h, __fmul_rn(w[p], b[p]));
}

}  // namespace cvs
