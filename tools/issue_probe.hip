// tools/issue_probe.hip -- how fast can ONE wave issue f32 VALU work on gfx950, as a function of the
// number of independent dependency chains (ILP) and of the waves resident per SIMD?  (diagnostic)
// Every wave runs ITERS x 16 v_fma_f32 split into CH independent chains and stamps its lifetime in
// shader-clock ticks (s_memtime); reported: cycles per wave64 instruction per wave, and per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o tools/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// MIX: after every 2 VALU instructions one independent SALU instruction (the basis kernels carry ~1 SALU per 2 VALU)
template <int CH, int MIX = 0>
__global__ __launch_bounds__(256) void k(unsigned long long* ticks, float* out, float a, float b, int iters)
{
    float x[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) x[i] = (float)threadIdx.x + i;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / CH; ++r)
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if constexpr (MIX == 1) { if (i & 1) { int d; asm volatile("s_mov_b32 %0, 1" : "=s"(d)); } }
                if constexpr (MIX == 2) { if (i & 1) { float d; asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(0)); } }
            }
        if constexpr (MIX == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CH, int MIX = 0>
static void run(int waves_per_simd, unsigned long long* dt, float* dout, int iters)
{
    const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves = one per SIMD) x waves_per_simd
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<CH, MIX>), dim3(blocks), dim3(256), 0, 0, dt, dout, 0.999f, 0.001f, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), dt, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double ninstr = 16.0 * iters;
    printf("%s chains %2d  waves/SIMD %d : %6.2f cycles per instr per wave, %5.2f per SIMD  (launch %.3f ms, %.1f TFLOP/s)\n", MIX == 1 ? "valu+salu 2:1" : MIX == 2 ? "valu+lds 2:1 " : "valu only    ", CH,
           waves_per_simd, med / ninstr, med / ninstr / waves_per_simd, ms, 2.0 * 64 * ninstr * blocks * 4 / ms / 1e9);
}

int main()
{
    unsigned long long* dt;
    float* dout;
    hipMalloc(&dt, 256 * 8 * 4 * 8);
    hipMalloc(&dout, 256 * 8 * 256 * 4);
    const int iters = 8192;
    for (int w = 1; w <= 8; ++w) {
        run<1>(w, dt, dout, iters);
        run<2>(w, dt, dout, iters);
        run<4>(w, dt, dout, iters);
        run<8>(w, dt, dout, iters);
        run<4, 1>(w, dt, dout, iters);
        run<8, 1>(w, dt, dout, iters);
        run<8, 2>(w, dt, dout, iters);
    }
    return 0;
}
