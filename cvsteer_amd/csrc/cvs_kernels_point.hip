// cvs_kernels_point.hip -- "K2..K5": the per-pixel stages of the cvsteer hot path for gfx950.
//
// Each reference stage is a chain of full-image OpenCV temporaries (cv::Mat::mul, MatExpr,
// cartToPolar, polarToCart, compare/copyTo ...).  Here every stage is ONE streaming pass:
// read each needed plane once (16 B per lane when rows are 16-B aligned), do the whole
// per-pixel expression in registers, write each requested output once.  All are HBM-bound.
// This file is built with -ffp-contract=off: products and sums round separately, like the
// reference's separate cv::Mat nodes.
//
//   OP_G2_ORIENT        SteerableFiltersG2.cpp:70-99
//   OP_G2_STEER_SCALAR  SteerableFiltersG2.cpp:137-145, 157-165
//   OP_G2_STEER_MAP     SteerableFiltersG2.cpp:147-155, 167-177
//   OP_G4_STEER_SCALAR  SteerableFiltersG4.cpp:114-122
//   OP_G4_STEER_MAP     SteerableFiltersG4.cpp:92-112
//   OP_MAG_PHASE        SteerableFiltersG2.cpp:107-112
//   OP_PHASE_WEIGHTS    SteerableFiltersG2.cpp:179-186
//   OP_FIND             SteerableFiltersG2.cpp:194-212 (three maps in one pass)
//   OP_WRAP             SteerableFilters.cpp:46-51
//   OP_G4_ORIENT        (extension, not in the reference) C1..C3 / theta / strength for G4+H4
//   OP_G2_PIPELINE      test/test.cpp:86-90 / example/steer.cpp:87-90 in one pass
#include <hip/hip_runtime.h>

#include "cvs_device_math.h"
#include "cvs_internal.h"

namespace cvs {

template <PointOp OP> struct OpShape;
template <> struct OpShape<OP_G2_ORIENT> { static constexpr int NIN = 7, NOUT = 5; };
template <> struct OpShape<OP_G2_STEER_SCALAR> { static constexpr int NIN = 10, NOUT = 5; };  // 7 basis, c1..c3
template <> struct OpShape<OP_G2_STEER_MAP> { static constexpr int NIN = 11, NOUT = 5; };     // 7 basis, c1..c3, theta
template <> struct OpShape<OP_G4_STEER_SCALAR> { static constexpr int NIN = 14, NOUT = 5; };  // 11 basis, [c1..c3: extension]
template <> struct OpShape<OP_G4_STEER_MAP> { static constexpr int NIN = 15, NOUT = 5; };     // 11 basis, [c1..c3], theta at 14
template <> struct OpShape<OP_G4_ORIENT> { static constexpr int NIN = 11, NOUT = 5; };        // extension
template <> struct OpShape<OP_MAG_PHASE> { static constexpr int NIN = 2, NOUT = 2; };
template <> struct OpShape<OP_PHASE_WEIGHTS> { static constexpr int NIN = 1, NOUT = 1; };
template <> struct OpShape<OP_FIND> { static constexpr int NIN = 2, NOUT = 3; };
template <> struct OpShape<OP_WRAP> { static constexpr int NIN = 1, NOUT = 1; };
template <> struct OpShape<OP_G2_PIPELINE> { static constexpr int NIN = 11, NOUT = 8; };      // 7 basis, c1..c3, theta

// one pixel.  in[] / out[] follow the OpShape order; `need_*` are wave-uniform.
template <PointOp OP>
__device__ __forceinline__ void point_eval(const float* in, float* out, const PointArgs& a, bool need_e, bool need_mp)
{
    if constexpr (OP == OP_G2_ORIENT) {
        g2_orientation(in, a.atan_mode, out[0], out[1], out[2], out[3], out[4]);
    } else if constexpr (OP == OP_G2_STEER_SCALAR) {
        g2_steer_weights(in, a.w, out[0], out[1]);
        if (need_e) out[2] = __fadd_rn(__fadd_rn(in[7], __fmul_rn(a.c2t, in[8])), __fmul_rn(a.s2t, in[9]));
        if (need_mp) mag_phase(out[0], out[1], a.atan_mode, out[3], out[4]);
    } else if constexpr (OP == OP_G2_STEER_MAP) {
        const float th = in[10];
        g2_steer_angle(in, th, out[0], out[1]);
        if (need_e) {  // G2.cpp:174-176
            float s2, c2;
            sincos_any(__fmul_rn(th, 2.0f), s2, c2);
            out[2] = __fadd_rn(__fadd_rn(in[7], __fmul_rn(in[8], c2)), __fmul_rn(in[9], s2));
        }
        if (need_mp) mag_phase(out[0], out[1], a.atan_mode, out[3], out[4]);
    } else if constexpr (OP == OP_G4_STEER_SCALAR) {
        g4_steer_weights(in, a.w, out[0], out[1]);
        // e / magnitude / phase for G4 are an opt-in extension (the reference has none, G4.cpp:88-90)
        if (need_e) out[2] = __fadd_rn(__fadd_rn(in[11], __fmul_rn(a.c2t, in[12])), __fmul_rn(a.s2t, in[13]));
        if (need_mp) mag_phase(out[0], out[1], a.atan_mode, out[3], out[4]);
    } else if constexpr (OP == OP_G4_STEER_MAP) {
        const float th = in[14];
        g4_steer_angle(in, th, out[0], out[1]);
        if (need_e) {
            float s2, c2;
            sincos_any(__fmul_rn(th, 2.0f), s2, c2);
            out[2] = __fadd_rn(__fadd_rn(in[11], __fmul_rn(in[12], c2)), __fmul_rn(in[13], s2));
        }
        if (need_mp) mag_phase(out[0], out[1], a.atan_mode, out[3], out[4]);
    } else if constexpr (OP == OP_G4_ORIENT) {
        g4_orientation(in, a.atan_mode, out[0], out[1], out[2], out[3], out[4]);
    } else if constexpr (OP == OP_MAG_PHASE) {
        mag_phase(in[0], in[1], a.atan_mode, out[0], out[1]);
    } else if constexpr (OP == OP_PHASE_WEIGHTS) {
        out[0] = phase_lambda(in[0], a.phi, a.signum != 0);
    } else if constexpr (OP == OP_FIND) {
        float le, ld, lb;
        phase_lambda3(in[1], le, ld, lb);   // one cos / sin pair for the three maps (cvs_device_math.h)
        out[0] = __fmul_rn(in[0], le);      // findEdges      G2.cpp:201-204
        out[1] = __fmul_rn(in[0], ld);      // findDarkLines  G2.cpp:205-208
        out[2] = __fmul_rn(in[0], lb);      // findBrightLines G2.cpp:209-212
    } else if constexpr (OP == OP_WRAP) {
        out[0] = wrap_pi(in[0]);
    } else if constexpr (OP == OP_G2_PIPELINE) {
        const float th = in[10];
        g2_steer_angle(in, th, out[0], out[1]);
        float s2, c2;
        sincos_any(__fmul_rn(th, 2.0f), s2, c2);
        out[2] = __fadd_rn(__fadd_rn(in[7], __fmul_rn(in[8], c2)), __fmul_rn(in[9], s2));
        mag_phase(out[0], out[1], a.atan_mode, out[3], out[4]);
        const float en = a.find_on_e ? out[2] : out[3];
        float le, ld, lb;
        phase_lambda3(out[4], le, ld, lb);
        out[5] = __fmul_rn(en, le);
        out[6] = __fmul_rn(en, ld);
        out[7] = __fmul_rn(en, lb);
    }
}

// VEC = 4: rows are walked in float4 units (host guarantees 16-B aligned pointers/pitches and
// cols % 4 == 0); VEC = 1: plain dwords.  blockIdx.y strides rows, x threads stride columns.
// NTL: the inputs are state planes of a large image that a pass long ago wrote and nobody reads again soon: nontemporal
// loads keep them from displacing what the caches hold (steer-by-map 67.5 -> 70.8 %, scalar steer 70.6 -> 71.8 % at 4096^2,
// tools/ab_m3.py).  Not for stages whose inputs the kernel before has just written (magnitude / phase, find*: 86 -> 75 %).
template <PointOp OP, int VEC, bool NT, bool NTL = false>
__global__ __launch_bounds__(256) void k_point(const PointArgs a)
{
    constexpr int NIN = OpShape<OP>::NIN, NOUT = OpShape<OP>::NOUT;
    const int ncv = a.cols / VEC;
    // wave-uniform "is this optional stage requested" flags
    bool need_e = false, need_mp = false;
    if constexpr (OP == OP_G2_STEER_SCALAR || OP == OP_G2_STEER_MAP || OP == OP_G4_STEER_SCALAR || OP == OP_G4_STEER_MAP) {
        need_e = a.out[2].p != nullptr;
        need_mp = a.out[3].p != nullptr || a.out[4].p != nullptr;
    }
    for (int row = blockIdx.y; row < a.rows; row += gridDim.y) {
        for (int cv = blockIdx.x * blockDim.x + threadIdx.x; cv < ncv; cv += gridDim.x * blockDim.x) {
            float vin[NIN][VEC];
#pragma unroll
            for (int i = 0; i < NIN; ++i) {
                if (a.in[i].p) {
                    const float* src = a.in[i].p + (size_t)row * a.in[i].pitch + (size_t)cv * VEC;
                    if constexpr (VEC == 4) {
                        typedef float f4l __attribute__((ext_vector_type(4)));
                        f4l v;
                        if constexpr (NTL) v = __builtin_nontemporal_load(reinterpret_cast<const f4l*>(src));
                        else v = *reinterpret_cast<const f4l*>(src);
                        vin[i][0] = v.x; vin[i][1] = v.y; vin[i][2] = v.z; vin[i][3] = v.w;
                    } else {
                        vin[i][0] = *src;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) vin[i][k] = 0.f;
                }
            }
            float vout[NOUT][VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float pin[NIN], pout[NOUT];
#pragma unroll
                for (int i = 0; i < NIN; ++i) pin[i] = vin[i][k];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) pout[o] = 0.f;
                point_eval<OP>(pin, pout, a, need_e, need_mp);
#pragma unroll
                for (int o = 0; o < NOUT; ++o) vout[o][k] = pout[o];
            }
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                if (a.out[o].p) {
                    float* dst = a.out[o].p + (size_t)row * a.out[o].pitch + (size_t)cv * VEC;
                    if constexpr (VEC == 4) {
                        typedef float f4 __attribute__((ext_vector_type(4)));
                        const f4 v = {vout[o][0], vout[o][1], vout[o][2], vout[o][3]};
                        if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst));
                        else *reinterpret_cast<f4*>(dst) = v;
                    } else {
                        if constexpr (NT) __builtin_nontemporal_store(vout[o][0], dst);
                        else *dst = vout[o][0];
                    }
                }
            }
        }
    }
}

static bool vec4_ok(const PointArgs& a, int nin, int nout)
{
    if (a.cols % 4) return false;
    for (int i = 0; i < nin; ++i)
        if (a.in[i].p && (((uintptr_t)a.in[i].p & 15) || (a.in[i].pitch & 3))) return false;
    for (int o = 0; o < nout; ++o)
        if (a.out[o].p && (((uintptr_t)a.out[o].p & 15) || (a.out[o].pitch & 3))) return false;
    return true;
}

template <PointOp OP>
static hipError_t launch_op(const PointArgs& a, hipStream_t s)
{
    constexpr int NIN = OpShape<OP>::NIN, NOUT = OpShape<OP>::NOUT;
    const bool v4 = vec4_ok(a, NIN, NOUT);
    const int ncv = v4 ? a.cols / 4 : a.cols;
    dim3 block(256);
    int gx = (ncv + 255) / 256;
    if (gx > 64) gx = 64;
    int gy = a.rows;
    // workgroups per CU before rows are strided: 16 for the plain weighted sums; the stages with per-pixel
    // transcendental work hide it better with more, shorter workgroups (interleaved A/B: steer-by-map 67 -> 70 %,
    // magnitude/phase 80 -> 84 % of the HBM roofline at 4096^2, scalar steer 71 -> 70 % the other way)
    constexpr bool kHeavy = OP == OP_G2_STEER_MAP || OP == OP_G4_STEER_MAP || OP == OP_MAG_PHASE || OP == OP_FIND;
    const int cap = 256 * (kHeavy ? 64 : 16);
    if ((long)gx * gy > cap) gy = cap / gx > 0 ? cap / gx : 1;
    dim3 grid(gx, gy);
    constexpr bool kStateIn = OP == OP_G2_STEER_SCALAR || OP == OP_G2_STEER_MAP || OP == OP_G4_STEER_SCALAR || OP == OP_G4_STEER_MAP;
    if constexpr (kStateIn) {
        if (v4 && a.nt_stores && a.nt_loads) {
            hipLaunchKernelGGL((k_point<OP, 4, true, true>), grid, block, 0, s, a);
            return hipGetLastError();
        }
    }
    if (v4 && a.nt_stores) hipLaunchKernelGGL((k_point<OP, 4, true>), grid, block, 0, s, a);
    else if (v4) hipLaunchKernelGGL((k_point<OP, 4, false>), grid, block, 0, s, a);
    else if (a.nt_stores) hipLaunchKernelGGL((k_point<OP, 1, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_point<OP, 1, false>), grid, block, 0, s, a);
    return hipGetLastError();
}

hipError_t launch_point(PointOp op, const PointArgs& a, hipStream_t s)
{
    if (a.rows <= 0 || a.cols <= 0) return hipErrorInvalidValue;
    switch (op) {
        case OP_G2_ORIENT: return launch_op<OP_G2_ORIENT>(a, s);
        case OP_G2_STEER_SCALAR: return launch_op<OP_G2_STEER_SCALAR>(a, s);
        case OP_G2_STEER_MAP: return launch_op<OP_G2_STEER_MAP>(a, s);
        case OP_G4_STEER_SCALAR: return launch_op<OP_G4_STEER_SCALAR>(a, s);
        case OP_G4_STEER_MAP: return launch_op<OP_G4_STEER_MAP>(a, s);
        case OP_MAG_PHASE: return launch_op<OP_MAG_PHASE>(a, s);
        case OP_PHASE_WEIGHTS: return launch_op<OP_PHASE_WEIGHTS>(a, s);
        case OP_FIND: return launch_op<OP_FIND>(a, s);
        case OP_G2_PIPELINE: return launch_op<OP_G2_PIPELINE>(a, s);
        case OP_WRAP: return launch_op<OP_WRAP>(a, s);
        case OP_G4_ORIENT: return launch_op<OP_G4_ORIENT>(a, s);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------
// cv::normalize(src, dst, 0, 255, NORM_MINMAX, CV_8UC1)  (test/test.cpp:92-94,
// example/steer.cpp:96-98): per-image min/max, then saturate_cast<uchar>(v*scale + shift).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int float_key(float v)
{
    const int b = __float_as_int(v);
    return b >= 0 ? b : b ^ 0x7fffffff;  // monotone map float -> int
}
__device__ __forceinline__ float key_float(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

__global__ void k_minmax_init(int* mm)
{
    mm[0] = 0x7fffffff;
    mm[1] = (int)0x80000000;
}

__global__ __launch_bounds__(256) void k_minmax(const float* src, size_t pitch, int rows, int cols, int* mm)
{
    float lo = INFINITY, hi = -INFINITY;
    for (int row = blockIdx.y; row < rows; row += gridDim.y)
        for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x) {
            const float v = src[(size_t)row * pitch + c];
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    // one atomic pair per workgroup (all of them land on one cache line)
    __shared__ float wlo[4], whi[4];
    if ((threadIdx.x & 63) == 0) {
        wlo[threadIdx.x >> 6] = lo;
        whi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = fminf(fminf(wlo[0], wlo[1]), fminf(wlo[2], wlo[3]));
        hi = fmaxf(fmaxf(whi[0], whi[1]), fmaxf(whi[2], whi[3]));
        atomicMin(&mm[0], float_key(lo));
        atomicMax(&mm[1], float_key(hi));
    }
}

__global__ __launch_bounds__(256) void k_quantize_u8(const float* src, size_t pitch, int rows, int cols,
                                                      const int* mm, uint8_t* dst, size_t dst_step)
{
    const float lo = key_float(mm[0]), hi = key_float(mm[1]);
    const double d = (double)hi - (double)lo;
    const double scale_d = d > 2.2204460492503131e-16 ? 255.0 / d : 0.0;
    const float scale = (float)scale_d, shift = (float)(-(double)lo * scale_d);
    for (int row = blockIdx.y; row < rows; row += gridDim.y)
        for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x) {
            const float v = __fadd_rn(__fmul_rn(src[(size_t)row * pitch + c], scale), shift);
            int q = __float2int_rn(v);  // cvRound: half to even
            q = q < 0 ? 0 : q > 255 ? 255 : q;
            dst[(size_t)row * dst_step + c] = (uint8_t)q;
        }
}

// ---------------------------------------------------------------------------------------
// steer(const cv::Point& p, theta, g2, h2, e, magnitude, phase)  (G2.cpp:115-134): one pixel.
// One lane gathers the 7 (+3) state values at (y, x) and evaluates the reference's scalar
// expression: float sums left to right, libm-style atan2 (no wrap, no NaN patch) and sqrt.
// ---------------------------------------------------------------------------------------
__global__ void k_steer_point(const float* basis, size_t plane_stride, size_t offset, const float* orient, size_t orient_stride,
                              size_t orient_offset, PointArgs a, float* out5)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool have_orient = orient != nullptr;
    float v[10];
    for (int i = 0; i < 7; ++i) v[i] = basis[(size_t)i * plane_stride + offset];
    for (int i = 0; i < 3; ++i) v[7 + i] = have_orient ? orient[(size_t)i * orient_stride + orient_offset] : 0.f;
    const float g2 = __fadd_rn(__fadd_rn(__fmul_rn(a.w[0], v[0]), __fmul_rn(a.w[1], v[1])), __fmul_rn(a.w[2], v[2]));
    float h2 = __fadd_rn(__fmul_rn(a.w[3], v[3]), __fmul_rn(a.w[4], v[4]));
    h2 = __fadd_rn(h2, __fmul_rn(a.w[5], v[5]));
    h2 = __fadd_rn(h2, __fmul_rn(a.w[6], v[6]));
    out5[0] = g2;
    out5[1] = h2;
    out5[2] = have_orient ? __fadd_rn(__fadd_rn(v[7], __fmul_rn(a.c2t, v[8])), __fmul_rn(a.s2t, v[9])) : __int_as_float(0x7fc00000);
    out5[3] = __fsqrt_rn(__fadd_rn(__fmul_rn(h2, h2), __fmul_rn(g2, g2)));
    out5[4] = atan2f(h2, g2);
}

hipError_t launch_steer_point(const float* basis, size_t plane_stride, size_t offset, const float* orient, size_t orient_stride,
                              size_t orient_offset, const PointArgs& a, float* out5, hipStream_t s)
{
    hipLaunchKernelGGL(k_steer_point, dim3(1), dim3(64), 0, s, basis, plane_stride, offset, orient, orient_stride, orient_offset, a, out5);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Gaussian pyramid level (BASELINE config 3; the reference has no pyramid code): cv::pyrDown
// semantics -- [1 4 6 4 1]/16 separable blur, REFLECT_101, keep every second pixel, output
// ((rows+1)/2) x ((cols+1)/2).  One thread makes one output pixel from its 5x5 neighbourhood;
// the 25 reads of neighbouring threads overlap and are served by L1/L2, HBM sees each input
// line once: 16 B read + 4 B written per output pixel.
// ---------------------------------------------------------------------------------------
// horizontal [1 4 6 4 1] at output column x of input row s (integer weights, like the CPU restatement)
template <bool VEC2>
__device__ __forceinline__ float pyr_row(const float* s, int x, int cols, bool interior)
{
    float a, b, c, d, e;
    if (VEC2 && interior) {  // 2x-2 .. 2x+2 inside the row: two aligned 8-byte loads + one dword
        const float2 lo = *reinterpret_cast<const float2*>(s + 2 * x - 2);
        const float2 mi = *reinterpret_cast<const float2*>(s + 2 * x);
        a = lo.x; b = lo.y; c = mi.x; d = mi.y; e = s[2 * x + 2];
    } else {
        a = s[reflect101(2 * x - 2, cols)]; b = s[reflect101(2 * x - 1, cols)]; c = s[reflect101(2 * x, cols)];
        d = s[reflect101(2 * x + 1, cols)]; e = s[reflect101(2 * x + 2, cols)];
    }
    return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(c, 6.0f), __fmul_rn(__fadd_rn(b, d), 4.0f)), a), e);
}

// each thread makes TWO vertically adjacent output pixels from 7 input rows (3 of the 5 rows of
// each are shared) with 3 loads per row: 10.5 loads per output pixel instead of 25.
template <bool VEC2>
__global__ __launch_bounds__(256) void k_pyr_down(const float* src, size_t spitch, int rows, int cols,
                                                   float* dst, size_t dpitch, int orows, int ocols)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 2;
    if (x >= ocols || y >= orows) return;
    const bool interior = 2 * x - 2 >= 0 && 2 * x + 2 < cols;
    float r[7];
#pragma unroll
    for (int j = 0; j < 7; ++j)
        r[j] = pyr_row<VEC2>(src + (size_t)reflect101(2 * y + j - 2, rows) * spitch, x, cols, interior);
    const float v0 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(r[2], 6.0f), __fmul_rn(__fadd_rn(r[1], r[3]), 4.0f)), r[0]), r[4]);
    dst[(size_t)y * dpitch + x] = __fmul_rn(v0, 1.0f / 256.0f);
    if (y + 1 < orows) {
        const float v1 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(r[4], 6.0f), __fmul_rn(__fadd_rn(r[3], r[5]), 4.0f)), r[2]), r[6]);
        dst[(size_t)(y + 1) * dpitch + x] = __fmul_rn(v1, 1.0f / 256.0f);
    }
}

hipError_t launch_pyr_down(const float* src, size_t spitch, int rows, int cols, float* dst, size_t dpitch, hipStream_t s)
{
    const int orows = (rows + 1) / 2, ocols = (cols + 1) / 2;
    const dim3 grid((ocols + 63) / 64, ((orows + 1) / 2 + 3) / 4), block(256);
    const bool vec2 = (((uintptr_t)src & 7) == 0) && (spitch % 2 == 0);
    if (vec2) hipLaunchKernelGGL(k_pyr_down<true>, grid, block, 0, s, src, spitch, rows, cols, dst, dpitch, orows, ocols);
    else hipLaunchKernelGGL(k_pyr_down<false>, grid, block, 0, s, src, spitch, rows, cols, dst, dpitch, orows, ocols);
    return hipGetLastError();
}

// cv::Mat1f(const cv::Mat&) on an 8-bit image (test/test.cpp:85): widen to f32, unscaled
__global__ __launch_bounds__(256) void k_u8_to_f32(const uint8_t* src, size_t sstep, int rows, int cols, float* dst, size_t dpitch)
{
    for (int row = blockIdx.y; row < rows; row += gridDim.y)
        for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x)
            dst[(size_t)row * dpitch + c] = (float)src[(size_t)row * sstep + c];
}

hipError_t launch_u8_to_f32(const uint8_t* src, size_t sstep, int rows, int cols, float* dst, size_t dpitch, hipStream_t s)
{
    int gx = (cols + 255) / 256; if (gx > 16) gx = 16;
    int gy = rows > 256 ? 256 : rows;
    hipLaunchKernelGGL(k_u8_to_f32, dim3(gx, gy), dim3(256), 0, s, src, sstep, rows, cols, dst, dpitch);
    return hipGetLastError();
}

// Mat::convertTo(dst, CV_8UC1, alpha, beta) (example/steer.cpp:94-96): saturate_cast<uchar>(v*alpha + beta)
__global__ __launch_bounds__(256) void k_convert_u8(const float* src, size_t pitch, int rows, int cols, float alpha, float beta,
                                                     uint8_t* dst, size_t dst_step)
{
    for (int row = blockIdx.y; row < rows; row += gridDim.y)
        for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x) {
            int q = __float2int_rn(__fadd_rn(__fmul_rn(src[(size_t)row * pitch + c], alpha), beta));
            q = q < 0 ? 0 : q > 255 ? 255 : q;
            dst[(size_t)row * dst_step + c] = (uint8_t)q;
        }
}

hipError_t launch_convert_u8(const float* src, size_t pitch, int rows, int cols, float alpha, float beta, uint8_t* dst,
                             size_t dst_step, hipStream_t s)
{
    int gx = (cols + 255) / 256; if (gx > 16) gx = 16;
    int gy = rows > 256 ? 256 : rows;
    hipLaunchKernelGGL(k_convert_u8, dim3(gx, gy), dim3(256), 0, s, src, pitch, rows, cols, alpha, beta, dst, dst_step);
    return hipGetLastError();
}

// ---- the same for n equally sized planes at a constant stride: blockIdx.z = plane, one launch for all of them ----
__global__ void k_minmax_init_n(int* mm, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        mm[2 * i] = 0x7fffffff;
        mm[2 * i + 1] = (int)0x80000000;
    }
}

// VEC: rows are 16-byte aligned and cols % 4 == 0 -- four pixels per lane and access
template <bool VEC>
__global__ __launch_bounds__(256) void k_minmax_n(const float* src, size_t plane_stride, size_t pitch, int rows, int cols, int* mm)
{
    src += (size_t)blockIdx.z * plane_stride;
    mm += 2 * blockIdx.z;
    float lo = INFINITY, hi = -INFINITY;
    for (int row = blockIdx.y; row < rows; row += gridDim.y) {
        if constexpr (VEC) {
            const float4* r4 = reinterpret_cast<const float4*>(src + (size_t)row * pitch);
            for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cols / 4; c += gridDim.x * blockDim.x) {
                const float4 v = r4[c];
                lo = fminf(fminf(lo, v.x), fminf(v.y, fminf(v.z, v.w)));
                hi = fmaxf(fmaxf(hi, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
            }
        } else {
            for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x) {
                const float v = src[(size_t)row * pitch + c];
                lo = fminf(lo, v);
                hi = fmaxf(hi, v);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    // one atomic pair per workgroup: with hundreds of planes in one launch the per-wave atomics of the single-plane
    // kernel (thousands on one cache line) would cost more than the reads
    __shared__ float wlo[4], whi[4];
    if ((threadIdx.x & 63) == 0) {
        wlo[threadIdx.x >> 6] = lo;
        whi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = fminf(fminf(wlo[0], wlo[1]), fminf(wlo[2], wlo[3]));
        hi = fmaxf(fmaxf(whi[0], whi[1]), fmaxf(whi[2], whi[3]));
        atomicMin(&mm[0], float_key(lo));
        atomicMax(&mm[1], float_key(hi));
    }
}

// mm == nullptr: convertTo(alpha, beta); else normalize with plane z's own min / max (same arithmetic as k_quantize_u8)
template <bool VEC>
__global__ __launch_bounds__(256) void k_to_u8_n(const float* src, size_t plane_stride, size_t pitch, int rows, int cols, const int* mm,
                                                  float alpha, float beta, uint8_t* dst, size_t dst_plane_stride, size_t dst_step)
{
    src += (size_t)blockIdx.z * plane_stride;
    dst += (size_t)blockIdx.z * dst_plane_stride;
    float scale = alpha, shift = beta;
    if (mm) {
        const float lo = key_float(mm[2 * blockIdx.z]), hi = key_float(mm[2 * blockIdx.z + 1]);
        const double d = (double)hi - (double)lo;
        const double scale_d = d > 2.2204460492503131e-16 ? 255.0 / d : 0.0;
        scale = (float)scale_d;
        shift = (float)(-(double)lo * scale_d);
    }
    auto quant = [&](float v) {
        int q = __float2int_rn(__fadd_rn(__fmul_rn(v, scale), shift));  // cvRound: half to even
        return (unsigned)(q < 0 ? 0 : q > 255 ? 255 : q);
    };
    for (int row = blockIdx.y; row < rows; row += gridDim.y) {
        if constexpr (VEC) {  // four pixels per lane: one 16-byte load, one 4-byte store
            const float4* r4 = reinterpret_cast<const float4*>(src + (size_t)row * pitch);
            unsigned* d4 = reinterpret_cast<unsigned*>(dst + (size_t)row * dst_step);
            for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cols / 4; c += gridDim.x * blockDim.x) {
                const float4 v = r4[c];
                d4[c] = quant(v.x) | (quant(v.y) << 8) | (quant(v.z) << 16) | (quant(v.w) << 24);
            }
        } else {
            for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x)
                dst[(size_t)row * dst_step + c] = (uint8_t)quant(src[(size_t)row * pitch + c]);
        }
    }
}

hipError_t launch_to_u8_n(const float* src, size_t plane_stride, size_t pitch, int rows, int cols, int n, bool minmax, float* minmax2n,
                          float alpha, float beta, uint8_t* dst, size_t dst_plane_stride, size_t dst_step, hipStream_t s)
{
    // four pixels per lane where every row of source and destination allows it
    const bool vec = cols % 4 == 0 && pitch % 4 == 0 && plane_stride % 4 == 0 && reinterpret_cast<uintptr_t>(src) % 16 == 0 &&
                     dst_step % 4 == 0 && dst_plane_stride % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 4 == 0;
    const int lanes = vec ? cols / 4 : cols;
    int gx = (lanes + 255) / 256; if (gx > 16) gx = 16;
    int gy = rows > 256 ? 256 : rows;
    if (n >= 8 && gy > 32) gy = 32;  // plenty of workgroups from the planes alone; fewer atomics per plane
    int* mm = reinterpret_cast<int*>(minmax2n);
    for (int z0 = 0; z0 < n; z0 += 65535) {  // grid.z limit
        const int nz = n - z0 < 65535 ? n - z0 : 65535;
        const float* sp = src + (size_t)z0 * plane_stride;
        if (minmax) {
            hipLaunchKernelGGL(k_minmax_init_n, dim3((nz + 255) / 256), dim3(256), 0, s, mm + 2 * z0, nz);
            if (vec) hipLaunchKernelGGL(k_minmax_n<true>, dim3(gx, gy, nz), dim3(256), 0, s, sp, plane_stride, pitch, rows, cols, mm + 2 * z0);
            else hipLaunchKernelGGL(k_minmax_n<false>, dim3(gx, gy, nz), dim3(256), 0, s, sp, plane_stride, pitch, rows, cols, mm + 2 * z0);
        }
        if (vec) hipLaunchKernelGGL(k_to_u8_n<true>, dim3(gx, gy, nz), dim3(256), 0, s, sp, plane_stride, pitch, rows, cols, minmax ? mm + 2 * z0 : nullptr, alpha, beta,
                                    dst + (size_t)z0 * dst_plane_stride, dst_plane_stride, dst_step);
        else hipLaunchKernelGGL(k_to_u8_n<false>, dim3(gx, gy, nz), dim3(256), 0, s, sp, plane_stride, pitch, rows, cols, minmax ? mm + 2 * z0 : nullptr, alpha, beta,
                                dst + (size_t)z0 * dst_plane_stride, dst_plane_stride, dst_step);
    }
    return hipGetLastError();
}

hipError_t launch_minmax(const float* src, size_t pitch, int rows, int cols, float* minmax2, hipStream_t s)
{
    int* mm = reinterpret_cast<int*>(minmax2);
    hipLaunchKernelGGL(k_minmax_init, dim3(1), dim3(1), 0, s, mm);
    int gx = (cols + 255) / 256; if (gx > 16) gx = 16;
    int gy = rows > 256 ? 256 : rows;
    hipLaunchKernelGGL(k_minmax, dim3(gx, gy), dim3(256), 0, s, src, pitch, rows, cols, mm);
    return hipGetLastError();
}

hipError_t launch_quantize_u8(const float* src, size_t pitch, int rows, int cols, const float* minmax2,
                              uint8_t* dst, size_t dst_step, hipStream_t s)
{
    int gx = (cols + 255) / 256; if (gx > 16) gx = 16;
    int gy = rows > 256 ? 256 : rows;
    hipLaunchKernelGGL(k_quantize_u8, dim3(gx, gy), dim3(256), 0, s, src, pitch, rows, cols,
                       reinterpret_cast<const int*>(minmax2), dst, dst_step);
    return hipGetLastError();
}

}  // namespace cvs
