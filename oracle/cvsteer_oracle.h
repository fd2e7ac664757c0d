/*
 * cvsteer_oracle.h -- CPU restatement of the cvsteer hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X build.  It restates, in plain C with no
 * dependencies, what headupinclouds/cvsteer computes through OpenCV on the CPU:
 *   - tap generation            (reference cvsteer/SteerableFilters.cpp:33-42,
 *                                SteerableFiltersG2.cpp:35-42, SteerableFiltersG4.cpp:34-45)
 *   - cv::sepFilter2D           (call sites SteerableFiltersG2.cpp:62-68, G4.cpp:69-80)
 *   - C1..C3 / theta / strength (SteerableFiltersG2.cpp:70-99)
 *   - wrap                      (SteerableFilters.cpp:46-51)
 *   - magnitude / phase         (SteerableFiltersG2.cpp:107-112)
 *   - steer (point/scalar/map)  (SteerableFiltersG2.cpp:115-177, G4.cpp:92-122)
 *   - phaseWeights / find*      (SteerableFiltersG2.cpp:179-212)
 *
 * The arithmetic of the reference lives in OpenCV (third party, not vendored in the
 * reference tree; pinned indirectly through Hunter v0.19.238 => OpenCV 3.4.x).  OpenCV is
 * absent from this image, so the OpenCV primitives are restated from their documented
 * semantics (and, where marked "recalled", from the published 3.4.x implementation).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - taps: bit-exact against the reference's own tap functions compiled in place
 *     (oracle/ref_taps.mk -> tests/golden/taps_ref.json).
 *   - full G2 pipeline: the reference's own golden JPEGs (test/test.cpp:70-108) at the
 *     reference's own tolerance (mean-L1 <= 1.0 of 255); we score ~0.01-0.03.
 *   - per-plane float values at the 1e-5 level: NOT pinned by any reference data
 *     ("parity unpinned" at that level) -- checked against an f64-accumulated restatement only.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this.
 * The product path (cvsteer_amd/, include/) never links or imports it.
 */
#ifndef CVSTEER_ORACLE_H
#define CVSTEER_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORA_KIND_G2 = 2, ORA_KIND_G4 = 4 };
enum { ORA_ATAN_CV = 0, ORA_ATAN_EXACT = 1 };

/* number of 1-D tap vectors / basis planes for a kind: G2 -> 7 / 7, G4 -> 11 / 11 */
int ora_num_filters(int kind);

/* 1-D taps: idx order follows the reference members m_g1..m_g3,m_h1..m_h4 (G2) and
 * m_g1..m_g5,m_h1..m_h6 (G4).  out has 2*width+1 floats. */
int ora_make_taps(int kind, int idx, int width, float spacing, float* out);

/* which (kx, ky) tap indices build basis plane p (reference setup() call order) */
int ora_basis_pair(int kind, int p, int* kx_idx, int* ky_idx);

/* BORDER_REFLECT_101 index map (cv::borderInterpolate semantics) */
int ora_reflect101(int p, int len);

/* cv::sepFilter2D restatement, f32 data, f32 intermediate row buffer, f32 accumulation:
 * row pass = plain left-to-right MAC, column pass = symmetric/antisymmetric folded form
 * when the kernel is exactly (anti)symmetric, else plain.  dst dense rows*cols. */
void ora_sepfilter2d_f32(const float* src, int rows, int cols, size_t src_step_elems,
                         const float* kx, const float* ky, int width, float* dst);

/* rows [y_lo, y_hi) only of the same result (dst is still the whole dense plane): what one thread of the
 * row-parallel timing leg computes */
void ora_sepfilter2d_f32_rows(const float* src, int rows, int cols, size_t src_step_elems,
                              const float* kx, const float* ky, int width, float* dst, int y_lo, int y_hi);

/* same correlation with double accumulation and no intermediate rounding -> double plane.
 * This is the "truth" both the f32 oracle and the GPU are compared against. */
void ora_sepfilter2d_f64(const float* src, int rows, int cols, size_t src_step_elems,
                         const float* kx, const float* ky, int width, double* dst);

/* all basis planes of a kind; basis = nplanes dense planes back to back */
void ora_basis(int kind, const float* src, int rows, int cols, size_t src_step_elems,
               int width, float spacing, float* basis);
void ora_basis_f64(int kind, const float* src, int rows, int cols, size_t src_step_elems,
                   int width, float spacing, double* basis);

/* cv::cartToPolar (radians).  mode ORA_ATAN_CV = OpenCV fastAtan2 polynomial, EXACT = atan2f */
void ora_cart_to_polar(const float* x, const float* y, size_t n, float* mag, float* angle, int mode);
/* cv::polarToCart with empty magnitude: c = cos(a), s = sin(a) */
void ora_polar_to_cart(const float* a, size_t n, float* c, float* s);
/* SteerableFilters::wrap */
void ora_wrap(const float* a, size_t n, float* out);

/* SteerableFiltersG2::setup steps (ii)-(iv): from 7 basis planes to C1,C2,C3,theta,strength */
void ora_g2_orientation(const float* basis, size_t n, float* c1, float* c2, float* c3,
                        float* theta, float* strength, int mode);

/* computeMagnitudeAndPhase */
void ora_mag_phase(const float* g, const float* h, size_t n, float* mag, float* phase, int mode);

/* steer, scalar theta.  e/mag/phase may be NULL (then c1..c3 may be NULL). */
void ora_g2_steer_scalar(const float* basis, const float* c1, const float* c2, const float* c3,
                         size_t n, float theta, float* g2, float* h2, float* e, float* mag,
                         float* phase, int mode);
/* steer, per-pixel theta map */
void ora_g2_steer_map(const float* basis, const float* c1, const float* c2, const float* c3,
                      size_t n, const float* theta, float* g2, float* h2, float* e, float* mag,
                      float* phase, int mode);
/* steer at one pixel index i: out = {g2, h2, e, magnitude, phase} (libm atan2, no wrap) */
void ora_g2_steer_point(const float* basis, const float* c1, const float* c2, const float* c3,
                        size_t n, size_t i, float theta, float out[5]);

/* phaseWeights + find*: out = e * lambda(phase; phi, signum).  k ignored like the reference. */
void ora_phase_weights(const float* phase, size_t n, float phi, int signum, float k, float* lambda);
void ora_find(const float* e, const float* phase, size_t n, float* edges, float* dark, float* bright);

/* G4: steer only (reference has no orientation / magnitude for G4) */
void ora_g4_steer_scalar(const float* basis, size_t n, float theta, float* g4, float* h4);
void ora_g4_steer_map(const float* basis, size_t n, const float* theta, float* g4, float* h4);

/* EXTENSION (not in the reference): G4/H4 oriented-energy coefficients, dominant angle, strength */
void ora_g4_orientation(const float* basis, size_t n, float* c1, float* c2, float* c3,
                        float* theta, float* strength, int mode);

/* cv::pyrDown restated (config 3 pyramid; not in the reference, unpinned). dst is ((rows+1)/2) x ((cols+1)/2) dense */
void ora_pyr_down(const float* src, int rows, int cols, size_t src_step_elems, float* dst);

/* ---- timing legs for bench.py cpu_baseline (reference call sequence, one thread) ---- */
/* G2: 7 sepFilter2D + scalar steer (M2).  returns seconds for `reps` repetitions */
double ora_time_g2_filter_steer(const float* src, int rows, int cols, float theta, int reps);
/* the same, one image with its rows split over `threads` host threads (bands with halo rows; same values) */
double ora_time_g2_filter_steer_mt(const float* src, int rows, int cols, float theta, int reps, int threads);

#ifdef __cplusplus
}
#endif
#endif
