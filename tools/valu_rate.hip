// tools/valu_rate.hip -- what does a wave64 f32 vector instruction cost on this GPU, plain and packed?  (diagnostic)
// Four loops of 16 independent fma chains per lane, 8 waves per SIMD, every CU busy:
//   plain   v_fma_f32  v, v, s, s          (inline asm: the compiler's SLP pass would pack a C++ loop into v_pk_fma_f32)
//   packed  v_pk_fma_f32 v[2], v[2], v[2], v[2]
//   packed, the multiplier a pair of SGPRs and the multiplicand ONE register broadcast to both halves (op_sel_hi = 0) --
//           the shape a filter tap pair x one sample takes
//   packed  v_pk_add_f32
// Prints wave-instructions per SIMD per cycle at the shader clock it measures itself (s_memtime ticks at 100 MHz; the clock is
// taken from a dependent-chain kernel whose length in cycles is known).
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ unsigned long long g_ticks[2];   // {shader-clock ticks, 100 MHz ticks} of one wave in the middle of the grid
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (MODE == 0 || (MODE >= 4 && MODE < 34)) {
        float x[16];
        for (int i = 0; i < 16; ++i) x[i] = (float)threadIdx.x + i;
        const float c2 = b * 3.0f;
        asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21");
        asm volatile("s_mov_b64 vcc, 0x3333" ::: "vcc");
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(b));
                else if constexpr (MODE == 4) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(b));
                else if constexpr (MODE == 5) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                else if constexpr (MODE == 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                else if constexpr (MODE == 7) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                else if constexpr (MODE == 8) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(x[i]) : "s"(a));
                else if constexpr (MODE == 9) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(x[i]) : "v"(b));
                else if constexpr (MODE == 10) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                else if constexpr (MODE == 11) asm volatile("v_mul_f32_e32 %0, 0x3ec00000, %0" : "+v"(x[i]));            // literal 0.375
                else if constexpr (MODE == 12) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3ec00000" : "+v"(x[i]) : "v"(b));
                else if constexpr (MODE == 13) asm volatile("v_mul_f32_e32 %0, 0.5, %0" : "+v"(x[i]));                   // inline constant
                else if constexpr (MODE == 14) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b));
                else if constexpr (MODE == 15) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(x[i]), "v"(b) : "vcc");
                else if constexpr (MODE == 16) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(x[i]));
                else if constexpr (MODE == 17) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(x[i]));
                else if constexpr (MODE == 18) asm volatile("v_cvt_f32_u32_e32 %0, %0" : "+v"(x[i]));
                else if constexpr (MODE == 19) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(x[i]) : "s20");
                else if constexpr (MODE == 20) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(x[i]) : "s"(a));
                else if constexpr (MODE == 21) asm volatile("v_rndne_f32_e32 %0, %0" : "+v"(x[i]));
                else if constexpr (MODE == 22) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                else if constexpr (MODE == 23) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
                else if constexpr (MODE == 24) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x[i]) : "v"(b));
                else if constexpr (MODE == 25) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(x[i]) : "v"(b), "v"(c2));
                else if constexpr (MODE == 26) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                else if constexpr (MODE == 27) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                else if constexpr (MODE == 28) asm volatile("v_min_f32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                else if constexpr (MODE == 29) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                else if constexpr (MODE == 30) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(x[i]) : "v"(b));
                else if constexpr (MODE == 31) asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(x[i]) : "v"(b));
                else if constexpr (MODE == 32) asm volatile("v_cvt_i32_f32_e32 %0, %0" : "+v"(x[i]));
                else asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(x[i]));
            }
        float s = x[0];
        for (int i = 1; i < 16; ++i) s += x[i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
        if (blockIdx.x == 1000 && threadIdx.x == 0) { g_ticks[0] = __builtin_amdgcn_s_memtime() - c0; g_ticks[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    } else {
        f2 x[8];
        for (int i = 0; i < 8; ++i) x[i] = f2{(float)threadIdx.x + i, (float)i};
        const f2 va = {a, a * 0.5f}, vb = {b, b};
        f2 vs[8];
        for (int i = 0; i < 8; ++i) vs[i] = f2{a + 0.001f * i, a - 0.002f * i};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(va), "v"(vb));
                else if constexpr (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(x[i]) : "s"(va), "v"(vb));
                else if constexpr (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(vb));
                else if constexpr (MODE == 34) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(x[i]) : "s"(va), "v"(vb));   // tap halves swapped
                else if constexpr (MODE == 35) asm volatile("v_pk_add_f32 %0, %0, %0 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "+v"(x[i]));                  // {x + y, x - y}
                else if constexpr (MODE == 36) asm volatile("v_pk_add_f32 %0, %0, %1 neg_hi:[0,1]" : "+v"(x[i]) : "v"(vb));
                else if constexpr (MODE == 37) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(x[i]) : "s"(va));
                else if constexpr (MODE == 38) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x[i & 1]) : "s"(va), "v"(vb), "v"(x[i & 1]));                   // two dependent chains only
                else if constexpr (MODE == 39) asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n\ts_nop 0" : "+v"(x[i]) : "s"(va), "v"(vb));
                else if constexpr (MODE == 40) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "s"(vs[i]), "v"(vb));                                  // a different SGPR pair every time
                else if constexpr (MODE == 41) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "s"(vs[i]), "v"(x[(i + 3) & 7]));                      // ... and a different VGPR pair
                else asm volatile("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(x[i]) : "v"(x[(i + 3) & 7]), "v"(x[(i + 5) & 7]));                                                      // with the hazard nop the compiler puts between inline-asm packed ops
            }
        f2 s = x[0];
        for (int i = 1; i < 8; ++i) s += x[i];
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
        if (blockIdx.x == 1000 && threadIdx.x == 0) { g_ticks[0] = __builtin_amdgcn_s_memtime() - c0; g_ticks[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    }
}

// one wave per SIMD, a dependent chain of n v_add_u32: n x 4 cycles (a lone wave issues one vector instruction per 4 cycles)
__global__ __launch_bounds__(64) void k_clock(unsigned long long* t, int n)
{
    unsigned v = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) asm volatile("v_add_u32 %0, %0, %0" : "+v"(v));
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) t[0] = t1 - t0 + (v == 12345u);
}

int main(int argc, char** argv)
{
    float* d;
    hipMalloc(&d, 256 * 2048 * 4 * sizeof(float));
    unsigned long long* dt;
    hipMalloc(&dt, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = argc > 1 ? atoi(argv[1]) : 1 << 14, blocks = 256 * 8;   // tools/profile_all_r05.sh: 4096 (counter calibration)
    const char* names[43] = {"v_fma_f32 v,v,s,v (VOP3)", "v_pk_fma_f32 (VGPR pairs)", "v_pk_fma_f32 (SGPR pair x broadcast VGPR)", "v_pk_add_f32",
                             "v_fmac_f32_e32 v,s,v (VOP2)", "v_fmac_f32_e32 v,v,v (VOP2)", "v_fma_f32 v,v,v,v (VOP3)", "v_add_f32_e32", "v_mul_f32_e32 v,s,v", "v_mov_b32_e32", "v_add_u32_e32",
                             "v_mul_f32_e32 v,literal,v", "v_fmaak_f32 (literal addend)", "v_mul_f32_e32 v,0.5,v (inline constant)", "v_cndmask_b32_e32 (reads vcc)", "v_cmp_lt_f32_e32 (writes vcc)",
                             "v_rcp_f32_e32", "v_sqrt_f32_e32", "v_cvt_f32_u32_e32", "v_readlane_b32", "v_writelane_b32", "v_rndne_f32_e32", "v_max_f32_e32", "v_div_fixup_f32",
                             "v_cndmask_b32_e64 s[20:21] (dst = src0)", "v_cndmask_b32_e64 s[20:21] (dst != srcs)", "v_and_b32_e32", "v_ldexp_f32", "v_min_f32_e32", "v_sub_f32_e32", "v_bfi_b32",
                             "v_add_f32_e64 |abs|", "v_cvt_i32_f32_e32", "v_lshlrev_b32_e32",
                             "v_pk_fma_f32 SGPR pair, halves swapped (op_sel)", "v_pk_add_f32 {x+y, x-y} (op_sel, neg_hi)", "v_pk_add_f32 neg_hi", "v_pk_mul_f32 SGPR pair", "v_pk_fma_f32, two dependent chains",
                             "v_pk_fma_f32 + s_nop 0", "v_pk_fma_f32, eight SGPR pairs in turn", "v_pk_fma_f32, eight SGPR pairs, rotating VGPR pairs", "v_pk_add_f32 neg_hi, rotating VGPR pairs"};
    for (int mode = 0; mode < 43; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 9) hipLaunchKernelGGL(k<9>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 10) hipLaunchKernelGGL(k<10>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 11) hipLaunchKernelGGL(k<11>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 12) hipLaunchKernelGGL(k<12>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 13) hipLaunchKernelGGL(k<13>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 14) hipLaunchKernelGGL(k<14>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 15) hipLaunchKernelGGL(k<15>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 17) hipLaunchKernelGGL(k<17>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 18) hipLaunchKernelGGL(k<18>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 19) hipLaunchKernelGGL(k<19>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 20) hipLaunchKernelGGL(k<20>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 21) hipLaunchKernelGGL(k<21>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 22) hipLaunchKernelGGL(k<22>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 23) hipLaunchKernelGGL(k<23>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 24) hipLaunchKernelGGL(k<24>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 25) hipLaunchKernelGGL(k<25>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 26) hipLaunchKernelGGL(k<26>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 27) hipLaunchKernelGGL(k<27>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 28) hipLaunchKernelGGL(k<28>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 29) hipLaunchKernelGGL(k<29>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 30) hipLaunchKernelGGL(k<30>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 31) hipLaunchKernelGGL(k<31>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 33) hipLaunchKernelGGL(k<33>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 34) hipLaunchKernelGGL(k<34>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 35) hipLaunchKernelGGL(k<35>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 36) hipLaunchKernelGGL(k<36>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 37) hipLaunchKernelGGL(k<37>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 38) hipLaunchKernelGGL(k<38>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 39) hipLaunchKernelGGL(k<39>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 40) hipLaunchKernelGGL(k<40>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 41) hipLaunchKernelGGL(k<41>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            if (mode == 42) hipLaunchKernelGGL(k<42>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        unsigned long long tk[2] = {0, 1};
        hipMemcpyFromSymbol(tk, HIP_SYMBOL(g_ticks), 16);
        const double mhz = (double)tk[0] / ((double)tk[1] / 100.0);   // shader-clock ticks per microsecond while the loop ran
        const bool pk = (mode >= 1 && mode <= 3) || mode >= 34;
        const double winst = (pk ? 8.0 : 16.0) * iters * 4.0 * blocks;          // wave-instructions of the loop
        const double lane_ops = winst * 64 * (pk ? 2 : 1);
        printf("%-50s %7.3f ms %7.1f G wave-instr/s %5.1f T f32 lane-ops/s  %5.2f cycles/instr/SIMD at %4.0f MHz\n", names[mode], best,
               winst / best / 1e6, lane_ops / best / 1e9, (mhz * 1e6) * 1024.0 / (winst / (best * 1e-3)), mhz);
    }
    return 0;
}
