/*
 * oracle_filter.c -- sepFilter2D / polar / steer / phase restated (TEST INFRASTRUCTURE,
 * see cvsteer_oracle.h).  Plain C, f32 arithmetic exactly where the reference's OpenCV path
 * computes in f32; build with -ffp-contract=off so every op rounds like a separate
 * cv::Mat expression node.
 */
#define _POSIX_C_SOURCE 200809L
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "cvsteer_oracle.h"

#define ORA_MAXW 32

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101): gfedcb|abcdefgh|gfedcba, repeated
 * until in range; len == 1 -> 0.  (sepFilter2D default border, G2.cpp:62-68 default args) */
int ora_reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

static int symmetry(const float* k, int w)
{ /* +1 mirror, -1 anti-mirror (centre exactly 0), 0 neither */
    int sym = 1, asym = (k[w] == 0.0f);
    for (int i = 1; i <= w; i++) {
        if (k[w + i] != k[w - i]) sym = 0;
        if (k[w + i] != -k[w - i]) asym = 0;
    }
    return sym ? 1 : asym ? -1 : 0;
}

/* SteerableFiltersG2.cpp:62-68: cv::sepFilter2D(image, dst, CV_32FC1, kx, ky.t()).
 * Correlation (no flip), anchor centre, delta 0, BORDER_REFLECT_101.
 * [recalled, OpenCV 3.4 filter.cpp] 9/13-tap row kernels take the generic RowFilter
 * (sequential s += k[i]*src[x+i]); (anti)symmetric column kernels take SymmColumnFilter
 * (k0*c + sum k_i*(S[+i] +/- S[-i])); the row buffer is f32. */
/* rows [y_lo, y_hi) of the filtered plane: the row pass runs over those rows plus w halo rows either side
 * (reflected at the image borders), the column pass over the band.  Every output value is computed exactly as by
 * the whole-image call, which is this with the band = the image. */
static void sepfilter2d_f32_band(const float* src, int rows, int cols, size_t sstep,
                                 const float* kx, const float* ky, int w, float* dst, int y_lo, int y_hi)
{
    int n = 2 * w + 1;
    int nb = y_hi - y_lo + 2 * w;  /* row-filtered lines kept: source rows y_lo - w .. y_hi + w - 1, reflected */
    float* rowbuf = (float*)malloc((size_t)nb * cols * sizeof(float));
    int* xi = (int*)malloc((size_t)(cols + 2 * w) * sizeof(int));
    for (int x = -w; x < cols + w; x++) xi[x + w] = ora_reflect101(x, cols);
    for (int k = 0; k < nb; k++) {
        const float* s = src + (size_t)ora_reflect101(y_lo - w + k, rows) * sstep;
        float* r = rowbuf + (size_t)k * cols;
        for (int x = 0; x < cols; x++) {
            float acc = kx[0] * s[xi[x]];
            for (int i = 1; i < n; i++) acc = acc + kx[i] * s[xi[x + i]];
            r[x] = acc;
        }
    }
    int sy = symmetry(ky, w);
    for (int y = y_lo; y < y_hi; y++) {
        const float* rp[2 * ORA_MAXW + 1];
        for (int j = -w; j <= w; j++) rp[j + w] = rowbuf + (size_t)(y - y_lo + w + j) * cols;
        float* d = dst + (size_t)y * cols;
        for (int x = 0; x < cols; x++) {
            float acc;
            if (sy == 1) {
                acc = ky[w] * rp[w][x];
                for (int j = 1; j <= w; j++) acc = acc + ky[w + j] * (rp[w + j][x] + rp[w - j][x]);
            } else if (sy == -1) {
                acc = 0.0f;
                for (int j = 1; j <= w; j++) acc = acc + ky[w + j] * (rp[w + j][x] - rp[w - j][x]);
            } else {
                acc = ky[0] * rp[0][x];
                for (int j = 1; j < n; j++) acc = acc + ky[j] * rp[j][x];
            }
            d[x] = acc;
        }
    }
    free(xi);
    free(rowbuf);
}

void ora_sepfilter2d_f32(const float* src, int rows, int cols, size_t sstep,
                         const float* kx, const float* ky, int w, float* dst)
{
    sepfilter2d_f32_band(src, rows, cols, sstep, kx, ky, w, dst, 0, rows);
}

void ora_sepfilter2d_f32_rows(const float* src, int rows, int cols, size_t sstep,
                              const float* kx, const float* ky, int w, float* dst, int y_lo, int y_hi)
{
    sepfilter2d_f32_band(src, rows, cols, sstep, kx, ky, w, dst, y_lo, y_hi);
}

void ora_sepfilter2d_f64(const float* src, int rows, int cols, size_t sstep,
                         const float* kx, const float* ky, int w, double* dst)
{
    int n = 2 * w + 1;
    double* rowbuf = (double*)malloc((size_t)rows * cols * sizeof(double));
    int* xi = (int*)malloc((size_t)(cols + 2 * w) * sizeof(int));
    for (int x = -w; x < cols + w; x++) xi[x + w] = ora_reflect101(x, cols);
    for (int y = 0; y < rows; y++) {
        const float* s = src + (size_t)y * sstep;
        double* r = rowbuf + (size_t)y * cols;
        for (int x = 0; x < cols; x++) {
            double acc = 0.0;
            for (int i = 0; i < n; i++) acc += (double)kx[i] * (double)s[xi[x + i]];
            r[x] = acc;
        }
    }
    for (int y = 0; y < rows; y++) {
        double* d = dst + (size_t)y * cols;
        for (int x = 0; x < cols; x++) d[x] = 0.0;
        for (int j = 0; j < n; j++) {
            const double* r = rowbuf + (size_t)ora_reflect101(y + j - w, rows) * cols;
            double k = (double)ky[j];
            for (int x = 0; x < cols; x++) d[x] += k * r[x];
        }
    }
    free(xi);
    free(rowbuf);
}

void ora_basis(int kind, const float* src, int rows, int cols, size_t sstep,
               int width, float spacing, float* basis)
{
    int np = ora_num_filters(kind);
    float taps[11][2 * ORA_MAXW + 1];
    for (int i = 0; i < np; i++) ora_make_taps(kind, i, width, spacing, taps[i]);
    for (int p = 0; p < np; p++) {
        int a, b;
        ora_basis_pair(kind, p, &a, &b);
        ora_sepfilter2d_f32(src, rows, cols, sstep, taps[a], taps[b], width,
                            basis + (size_t)p * rows * cols);
    }
}

void ora_basis_f64(int kind, const float* src, int rows, int cols, size_t sstep,
                   int width, float spacing, double* basis)
{
    int np = ora_num_filters(kind);
    float taps[11][2 * ORA_MAXW + 1];
    for (int i = 0; i < np; i++) ora_make_taps(kind, i, width, spacing, taps[i]);
    for (int p = 0; p < np; p++) {
        int a, b;
        ora_basis_pair(kind, p, &a, &b);
        ora_sepfilter2d_f64(src, rows, cols, sstep, taps[a], taps[b], width,
                            basis + (size_t)p * rows * cols);
    }
}

/* [recalled, OpenCV 3.4 mathfuncs_core: fastAtan32f] degrees, 7th-order odd polynomial on
 * min/max, then octant fix-ups; cartToPolar(angleInDegrees=false) scales by (float)(pi/180). */
static float cv_fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

static float angle_0_2pi(float y, float x, int mode)
{
    if (mode == ORA_ATAN_CV) return cv_fast_atan2_deg(y, x) * (float)(3.14159265358979323846 / 180.0);
    float a = atan2f(y, x);
    if (a < 0) a = a + 6.2831855f;
    return a;
}

/* cv::cartToPolar (G2.cpp:97, :109): mag = sqrt(x^2+y^2), angle in [0, 2pi) */
void ora_cart_to_polar(const float* x, const float* y, size_t n, float* mag, float* angle, int mode)
{
    for (size_t i = 0; i < n; i++) {
        if (mag) mag[i] = sqrtf(x[i] * x[i] + y[i] * y[i]);
        if (angle) angle[i] = angle_0_2pi(y[i], x[i], mode);
    }
}

/* cv::polarToCart(Mat(), angle, c, s) (G2.cpp:151,175,183): documented ~1e-6 accurate;
 * restated with libm in double, narrowed once. */
void ora_polar_to_cart(const float* a, size_t n, float* c, float* s)
{
    for (size_t i = 0; i < n; i++) {
        c[i] = (float)cos((double)a[i]);
        s[i] = (float)sin((double)a[i]);
    }
}

/* SteerableFilters.cpp:46-51: out = angle > pi ? angle - 2pi : angle, scalars narrowed to f32 */
static float wrap1(float a)
{
    const float PI_F = 3.14159274f, TWO_PI_F = 6.2831855f;
    return a > PI_F ? a - TWO_PI_F : a;
}
void ora_wrap(const float* a, size_t n, float* out)
{
    for (size_t i = 0; i < n; i++) out[i] = wrap1(a[i]);
}

/* SteerableFiltersG2.cpp:70-99.  Every product plane and every MatExpr node is an f32
 * rounding; sums are taken left to right as the expression is written. */
void ora_g2_orientation(const float* basis, size_t n, float* c1, float* c2, float* c3,
                        float* theta, float* strength, int mode)
{
    const float *A = basis, *B = basis + n, *C = basis + 2 * n;
    const float *HA = basis + 3 * n, *HB = basis + 4 * n, *HC = basis + 5 * n, *HD = basis + 6 * n;
    for (size_t i = 0; i < n; i++) {
        float g2aa = A[i] * A[i], g2ab = A[i] * B[i], g2ac = A[i] * C[i];
        float g2bb = B[i] * B[i], g2bc = B[i] * C[i], g2cc = C[i] * C[i];
        float h2aa = HA[i] * HA[i], h2ab = HA[i] * HB[i], h2ac = HA[i] * HC[i], h2ad = HA[i] * HD[i];
        float h2bb = HB[i] * HB[i], h2bc = HB[i] * HC[i], h2bd = HB[i] * HD[i];
        float h2cc = HC[i] * HC[i], h2cd = HC[i] * HD[i], h2dd = HD[i] * HD[i];
        /* [recalled] MatExpr folds alpha*(a +/- b) into addWeighted(a, alpha, b, +/-alpha):
         * the scalar is distributed, each product rounds, then the sum rounds. */
        float v1 = 0.5f * g2bb + 0.25f * g2ac;
        v1 = v1 + (0.375f * g2aa + 0.375f * g2cc);
        v1 = v1 + (0.3125f * h2aa + 0.3125f * h2dd);
        v1 = v1 + (0.5625f * h2bb + 0.5625f * h2cc);
        v1 = v1 + (0.375f * h2ac + 0.375f * h2bd);
        float v2 = 0.5f * g2aa - 0.5f * g2cc;
        v2 = v2 + (0.46875f * h2aa - 0.46875f * h2dd);
        v2 = v2 + (0.28125f * h2bb - 0.28125f * h2cc);
        v2 = v2 + (0.1875f * h2ac - 0.1875f * h2bd);
        float v3 = (-g2ab) - g2bc;
        v3 = v3 - (0.9375f * h2cd + 0.9375f * h2ab);
        v3 = v3 - 1.6875f * h2bc;
        v3 = v3 - 0.1875f * h2ad;
        if (c1) c1[i] = v1;
        if (c2) c2[i] = v2;
        if (c3) c3[i] = v3;
        if (strength) strength[i] = sqrtf(v2 * v2 + v3 * v3);
        if (theta) theta[i] = wrap1(angle_0_2pi(v3, v2, mode)) * 0.5f;
    }
}

/* SteerableFiltersG2.cpp:107-112 */
void ora_mag_phase(const float* g, const float* h, size_t n, float* mag, float* phase, int mode)
{
    for (size_t i = 0; i < n; i++) {
        if (mag) mag[i] = sqrtf(g[i] * g[i] + h[i] * h[i]);
        if (phase) {
            float p = wrap1(angle_0_2pi(h[i], g[i], mode));
            phase[i] = (p != p) ? 0.0f : p; /* patchNaNs */
        }
    }
}

/* steering weights from c = cos(theta), s = sin(theta): G2.cpp:118-120 / 140-142.
 * -2.0*ct*st and 3.0*... are double products narrowed to f32 (literals are double). */
static void g2_weights(float ct, float st, float g[3], float h[4])
{
    float ct2 = ct * ct, ct3 = ct2 * ct, st2 = st * st, st3 = st2 * st;
    g[0] = ct2;
    g[1] = (float)(-2.0 * (double)ct * (double)st);
    g[2] = st2;
    h[0] = ct3;
    h[1] = (float)(-3.0 * (double)ct2 * (double)st);
    h[2] = (float)(3.0 * (double)ct * (double)st2);
    h[3] = -st3;
}

/* SteerableFiltersG2.cpp:137-145 and :157-165 */
void ora_g2_steer_scalar(const float* basis, const float* c1, const float* c2, const float* c3,
                         size_t n, float theta, float* g2, float* h2, float* e, float* mag,
                         float* phase, int mode)
{
    float gw[3], hw[4];
    g2_weights(cosf(theta), sinf(theta), gw, hw);
    float c2t = (float)cos((double)theta * 2.0), s2t = (float)sin((double)theta * 2.0);
    for (size_t i = 0; i < n; i++) {
        float g = gw[0] * basis[i] + gw[1] * basis[n + i];
        g = g + gw[2] * basis[2 * n + i];
        float h = hw[0] * basis[3 * n + i] + hw[1] * basis[4 * n + i];
        h = h + hw[2] * basis[5 * n + i];
        h = h + hw[3] * basis[6 * n + i];
        if (g2) g2[i] = g;
        if (h2) h2[i] = h;
        if (e) {
            float v = c1[i] + c2t * c2[i];
            e[i] = v + s2t * c3[i];
        }
        if (mag || phase) {
            float m, p;
            ora_mag_phase(&g, &h, 1, &m, &p, mode);
            if (mag) mag[i] = m;
            if (phase) phase[i] = p;
        }
    }
}

/* SteerableFiltersG2.cpp:147-155 and :167-177: every .mul and every scalar scale is a node */
void ora_g2_steer_map(const float* basis, const float* c1, const float* c2, const float* c3,
                      size_t n, const float* theta, float* g2, float* h2, float* e, float* mag,
                      float* phase, int mode)
{
    for (size_t i = 0; i < n; i++) {
        float ct, st;
        ora_polar_to_cart(theta + i, 1, &ct, &st);
        float ct2 = ct * ct, ct3 = ct2 * ct, st2 = st * st, st3 = st2 * st;
        /* [recalled] s * a.mul(b).mul(M) is one cv::multiply(tmp, M, scale=s) = (s*tmp)*M */
        float g = ct2 * basis[i];
        g = g + ((-2.0f * (ct * st)) * basis[n + i]);
        g = g + st2 * basis[2 * n + i];
        float h = ct3 * basis[3 * n + i];
        h = h + ((-3.0f * (ct2 * st)) * basis[4 * n + i]);
        h = h + ((3.0f * (ct * st2)) * basis[5 * n + i]);
        h = h + ((-st3) * basis[6 * n + i]);
        if (g2) g2[i] = g;
        if (h2) h2[i] = h;
        if (e) {
            float t2 = theta[i] * 2.0f, c2t, s2t;
            ora_polar_to_cart(&t2, 1, &c2t, &s2t);
            float v = c1[i] + c2[i] * c2t;
            e[i] = v + c3[i] * s2t;
        }
        if (mag || phase) {
            float m, p;
            ora_mag_phase(&g, &h, 1, &m, &p, mode);
            if (mag) mag[i] = m;
            if (phase) phase[i] = p;
        }
    }
}

/* SteerableFiltersG2.cpp:115-134: libm cos/sin/atan2/sqrt, no wrap, no NaN patch */
void ora_g2_steer_point(const float* basis, const float* c1, const float* c2, const float* c3,
                        size_t n, size_t i, float theta, float out[5])
{
    float gw[3], hw[4];
    g2_weights(cosf(theta), sinf(theta), gw, hw);
    float g = gw[0] * basis[i] + gw[1] * basis[n + i] + gw[2] * basis[2 * n + i];
    float h = hw[0] * basis[3 * n + i] + hw[1] * basis[4 * n + i] + hw[2] * basis[5 * n + i] + hw[3] * basis[6 * n + i];
    out[0] = g;
    out[1] = h;
    float c2t = (float)cos((double)theta * 2.0), s2t = (float)sin((double)theta * 2.0);
    out[2] = c1[i] + (c2t * c2[i]) + (s2t * c3[i]);
    out[3] = sqrtf(h * h + g * g);
    out[4] = atan2f(h, g);
}

/* SteerableFiltersG2.cpp:179-186.  k is accepted and ignored, like the reference. */
void ora_phase_weights(const float* phase, size_t n, float phi, int signum, float k, float* lambda)
{
    (void)k;
    const float TWO_PI_F = 6.2831855f, HALF_PI_F = 1.57079637f;
    for (size_t i = 0; i < n; i++) {
        float err = signum ? fabsf(phase[i] - phi) : fabsf(fabsf(phase[i]) - fabsf(phi));
        float alt = TWO_PI_F - err;
        err = err < alt ? err : alt; /* cv::min(error, 2pi - error) */
        float ct = (float)cos((double)err);
        float l = ct * ct;
        if (fabsf(err) > HALF_PI_F) l = 0.0f;
        lambda[i] = l;
    }
}

/* SteerableFiltersG2.cpp:194-212: edges phi=pi/2 unsigned; dark phi=0 signed; bright phi=pi signed */
void ora_find(const float* e, const float* phase, size_t n, float* edges, float* dark, float* bright)
{
    float l;
    for (size_t i = 0; i < n; i++) {
        if (edges) { ora_phase_weights(phase + i, 1, 1.57079637f, 0, 2.0f, &l); edges[i] = e[i] * l; }
        if (dark) { ora_phase_weights(phase + i, 1, 0.0f, 1, 2.0f, &l); dark[i] = e[i] * l; }
        if (bright) { ora_phase_weights(phase + i, 1, 3.14159274f, 1, 2.0f, &l); bright[i] = e[i] * l; }
    }
}

/* SteerableFiltersG4.cpp:114-122 */
static void g4_weights_scalar(float theta, float g[5], float h[6])
{
    float ct = cosf(theta), ct2 = ct * ct, ct3 = ct2 * ct, ct4 = ct3 * ct, ct5 = ct4 * ct;
    float st = sinf(theta), st2 = st * st, st3 = st2 * st, st4 = st3 * st, st5 = st4 * st;
    g[0] = ct4;
    g[1] = (float)(-4.0 * (double)ct3 * (double)st);
    g[2] = (float)(6.0 * (double)ct2 * (double)st2);
    g[3] = (float)(-4.0 * (double)ct * (double)st3);
    g[4] = st4;
    h[0] = ct5;
    h[1] = -5.0f * ct4 * st;
    h[2] = (float)(10.0 * (double)ct3 * (double)st2);
    h[3] = (float)(-10.0 * (double)ct2 * (double)st3);
    h[4] = (float)(5.0 * (double)ct * (double)st4);
    h[5] = -st5;
}

void ora_g4_steer_scalar(const float* b, size_t n, float theta, float* g4, float* h4)
{
    float gw[5], hw[6];
    g4_weights_scalar(theta, gw, hw);
    for (size_t i = 0; i < n; i++) {
        float g = gw[0] * b[i] + gw[1] * b[n + i];
        for (int p = 2; p < 5; p++) g = g + gw[p] * b[p * n + i];
        float h = hw[0] * b[5 * n + i] + hw[1] * b[6 * n + i];
        for (int p = 2; p < 6; p++) h = h + hw[p] * b[(5 + p) * n + i];
        if (g4) g4[i] = g;
        if (h4) h4[i] = h;
    }
}

/* SteerableFiltersG4.cpp:92-112: weight planes first, then weight.mul(basis) summed l-to-r */
void ora_g4_steer_map(const float* b, size_t n, const float* theta, float* g4, float* h4)
{
    for (size_t i = 0; i < n; i++) {
        float ct, st;
        ora_polar_to_cart(theta + i, 1, &ct, &st);
        float ct2 = ct * ct, ct3 = ct2 * ct, ct4 = ct3 * ct, ct5 = ct4 * ct;
        float st2 = st * st, st3 = st2 * st, st4 = st3 * st, st5 = st4 * st;
        float gw[5] = { ct4, -4.0f * (ct3 * st), 6.0f * (ct2 * st2), -4.0f * (ct * st3), st4 };
        float hw[6] = { ct5, -5.0f * (ct4 * st), 10.0f * (ct3 * st2), -10.0f * (ct2 * st3), 5.0f * (ct * st4), -st5 };
        float g = gw[0] * b[i];
        for (int p = 1; p < 5; p++) g = g + gw[p] * b[p * n + i];
        float h = hw[0] * b[5 * n + i];
        for (int p = 1; p < 6; p++) h = h + hw[p] * b[(5 + p) * n + i];
        if (g4) g4[i] = g;
        if (h4) h4[i] = h;
    }
}

/* EXTENSION beyond the reference (it computes no G4 orientation, G4.h:55): C1..C3 of
 * E(theta) = g4(theta)^2 + h4(theta)^2 from the steering polynomials of G4.cpp:116-119, derived like
 * G2.cpp:93-95 (tools/gen_g4_orient.py prints this table and, for G2, the reference's own constants).
 * Unpinned by any reference data; checked against brute-force projection of E(theta) in the tests. */
typedef struct { int i, j, which; float k; } ora_g4_term;
static const ora_g4_term ORA_G4_TERMS[] = {
    {0, 0, 1, 35.f / 128},
    {0, 0, 2, 7.f / 16},
    {0, 1, 3, -7.f / 8},
    {0, 2, 1, 15.f / 32},
    {0, 2, 2, 3.f / 8},
    {0, 3, 3, -3.f / 8},
    {0, 4, 1, 3.f / 64},
    {1, 1, 1, 5.f / 8},
    {1, 1, 2, 1.f / 2},
    {1, 2, 3, -9.f / 4},
    {1, 3, 1, 3.f / 4},
    {1, 4, 3, -3.f / 8},
    {2, 2, 1, 27.f / 32},
    {2, 3, 3, -9.f / 4},
    {2, 4, 1, 15.f / 32},
    {2, 4, 2, -3.f / 8},
    {3, 3, 1, 5.f / 8},
    {3, 3, 2, -1.f / 2},
    {3, 4, 3, -7.f / 8},
    {4, 4, 1, 35.f / 128},
    {4, 4, 2, -7.f / 16},
    {5, 5, 1, 63.f / 256},
    {5, 5, 2, 105.f / 256},
    {5, 6, 3, -105.f / 128},
    {5, 7, 1, 35.f / 64},
    {5, 7, 2, 35.f / 64},
    {5, 8, 3, -35.f / 64},
    {5, 9, 1, 15.f / 128},
    {5, 9, 2, 5.f / 128},
    {5, 10, 3, -5.f / 128},
    {6, 6, 1, 175.f / 256},
    {6, 6, 2, 175.f / 256},
    {6, 7, 3, -175.f / 64},
    {6, 8, 1, 75.f / 64},
    {6, 8, 2, 25.f / 64},
    {6, 9, 3, -125.f / 128},
    {6, 10, 1, 15.f / 128},
    {6, 10, 2, -5.f / 128},
    {7, 7, 1, 75.f / 64},
    {7, 7, 2, 25.f / 64},
    {7, 8, 3, -125.f / 32},
    {7, 9, 1, 75.f / 64},
    {7, 9, 2, -25.f / 64},
    {7, 10, 3, -35.f / 64},
    {8, 8, 1, 75.f / 64},
    {8, 8, 2, -25.f / 64},
    {8, 9, 3, -175.f / 64},
    {8, 10, 1, 35.f / 64},
    {8, 10, 2, -35.f / 64},
    {9, 9, 1, 175.f / 256},
    {9, 9, 2, -175.f / 256},
    {9, 10, 3, -105.f / 128},
    {10, 10, 1, 63.f / 256},
    {10, 10, 2, -105.f / 256},
};

void ora_g4_orientation(const float* basis, size_t n, float* c1, float* c2, float* c3,
                        float* theta, float* strength, int mode)
{
    const int nt = (int)(sizeof(ORA_G4_TERMS) / sizeof(ORA_G4_TERMS[0]));
    for (size_t p = 0; p < n; p++) {
        float v1 = 0.f, v2 = 0.f, v3 = 0.f;
        for (int t = 0; t < nt; t++) {
            float term = ORA_G4_TERMS[t].k * (basis[(size_t)ORA_G4_TERMS[t].i * n + p] * basis[(size_t)ORA_G4_TERMS[t].j * n + p]);
            if (ORA_G4_TERMS[t].which == 1) v1 = v1 + term;
            else if (ORA_G4_TERMS[t].which == 2) v2 = v2 + term;
            else v3 = v3 + term;
        }
        if (c1) c1[p] = v1;
        if (c2) c2[p] = v2;
        if (c3) c3[p] = v3;
        if (strength) strength[p] = sqrtf(v2 * v2 + v3 * v3);
        if (theta) theta[p] = wrap1(angle_0_2pi(v3, v2, mode)) * 0.5f;
    }
}

/* cv::pyrDown(src, dst) with default size ((cols+1)/2, (rows+1)/2) and BORDER_REFLECT_101
 * (BORDER_DEFAULT): 5x5 Gaussian [1 4 6 4 1]/16 (x) [1 4 6 4 1]/16, then every second pixel.
 * NOT part of the reference (it has no pyramid code): restated from the
 * OpenCV documentation for BASELINE config 3; [recalled] float path = integer-weight row sum,
 * integer-weight column sum, one multiply by 1/256.  Parity for this function is unpinned. */
void ora_pyr_down(const float* src, int rows, int cols, size_t sstep, float* dst)
{
    int orows = (rows + 1) / 2, ocols = (cols + 1) / 2;
    float* rowbuf = (float*)malloc((size_t)rows * ocols * sizeof(float));
    for (int y = 0; y < rows; y++) {
        const float* s = src + (size_t)y * sstep;
        for (int x = 0; x < ocols; x++) {
            float c = s[ora_reflect101(2 * x, cols)];
            float l1 = s[ora_reflect101(2 * x - 1, cols)], r1 = s[ora_reflect101(2 * x + 1, cols)];
            float l2 = s[ora_reflect101(2 * x - 2, cols)], r2 = s[ora_reflect101(2 * x + 2, cols)];
            rowbuf[(size_t)y * ocols + x] = c * 6.0f + (l1 + r1) * 4.0f + l2 + r2;
        }
    }
    for (int y = 0; y < orows; y++) {
        const float* r0 = rowbuf + (size_t)ora_reflect101(2 * y - 2, rows) * ocols;
        const float* r1 = rowbuf + (size_t)ora_reflect101(2 * y - 1, rows) * ocols;
        const float* r2 = rowbuf + (size_t)ora_reflect101(2 * y, rows) * ocols;
        const float* r3 = rowbuf + (size_t)ora_reflect101(2 * y + 1, rows) * ocols;
        const float* r4 = rowbuf + (size_t)ora_reflect101(2 * y + 2, rows) * ocols;
        for (int x = 0; x < ocols; x++)
            dst[(size_t)y * ocols + x] = (r2[x] * 6.0f + (r1[x] + r3[x]) * 4.0f + r0[x] + r4[x]) * (1.0f / 256.0f);
    }
    free(rowbuf);
}

/* cpu_baseline leg: the reference call sequence for the headline unit of work --
 * 7 x sepFilter2D (G2.cpp:62-68) + scalar steer (G2.cpp:137-145) -- single thread. */
double ora_time_g2_filter_steer(const float* src, int rows, int cols, float theta, int reps)
{
    size_t n = (size_t)rows * cols;
    float* basis = (float*)malloc(7 * n * sizeof(float));
    float* g2 = (float*)malloc(n * sizeof(float));
    float* h2 = (float*)malloc(n * sizeof(float));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < reps; r++) {
        ora_basis(ORA_KIND_G2, src, rows, cols, (size_t)cols, 4, 0.67f, basis);
        ora_g2_steer_scalar(basis, NULL, NULL, NULL, n, theta, g2, h2, NULL, NULL, NULL, ORA_ATAN_CV);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    volatile float sink = g2[n / 2] + h2[n / 3];
    (void)sink;
    free(basis); free(g2); free(h2);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- the same G2 filter + steer sequence with the image rows split over `threads` host threads ---- */
#include <pthread.h>
typedef struct {
    const float* src; int rows, cols; float theta; float* basis; float* g2; float* h2; int y_lo, y_hi;
    float taps[7][2 * ORA_MAXW + 1];
} band_job;

static void* band_worker(void* arg)
{
    band_job* j = (band_job*)arg;
    size_t n = (size_t)j->rows * j->cols;
    for (int p = 0; p < 7; p++) {
        int a, b;
        ora_basis_pair(ORA_KIND_G2, p, &a, &b);
        sepfilter2d_f32_band(j->src, j->rows, j->cols, (size_t)j->cols, j->taps[a], j->taps[b], 4, j->basis + (size_t)p * n, j->y_lo, j->y_hi);
    }
    /* steer on the band: planes are dense, so a band is a contiguous range of every plane */
    size_t off = (size_t)j->y_lo * j->cols, cnt = (size_t)(j->y_hi - j->y_lo) * j->cols;
    float* bb = (float*)malloc(7 * cnt * sizeof(float));
    for (int p = 0; p < 7; p++) memcpy(bb + (size_t)p * cnt, j->basis + (size_t)p * n + off, cnt * sizeof(float));
    ora_g2_steer_scalar(bb, NULL, NULL, NULL, cnt, j->theta, j->g2 + off, j->h2 + off, NULL, NULL, NULL, ORA_ATAN_CV);
    free(bb);
    return NULL;
}

double ora_time_g2_filter_steer_mt(const float* src, int rows, int cols, float theta, int reps, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > rows / 16) threads = rows / 16 > 0 ? rows / 16 : 1;
    size_t n = (size_t)rows * cols;
    float* basis = (float*)malloc(7 * n * sizeof(float));
    float* g2 = (float*)malloc(n * sizeof(float));
    float* h2 = (float*)malloc(n * sizeof(float));
    band_job* jobs = (band_job*)malloc((size_t)threads * sizeof(band_job));
    pthread_t* tid = (pthread_t*)malloc((size_t)threads * sizeof(pthread_t));
    for (int t = 0; t < threads; t++) {
        band_job* j = &jobs[t];
        j->src = src; j->rows = rows; j->cols = cols; j->theta = theta; j->basis = basis; j->g2 = g2; j->h2 = h2;
        j->y_lo = (int)((long long)rows * t / threads);
        j->y_hi = (int)((long long)rows * (t + 1) / threads);
        for (int i = 0; i < 7; i++) ora_make_taps(ORA_KIND_G2, i, 4, 0.67f, j->taps[i]);
    }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < reps; r++) {
        for (int t = 0; t < threads; t++) pthread_create(&tid[t], NULL, band_worker, &jobs[t]);
        for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    volatile float sink = g2[n / 2] + h2[n / 3];
    (void)sink;
    free(tid); free(jobs); free(basis); free(g2); free(h2);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

