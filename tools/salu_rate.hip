// tools/salu_rate.hip -- how many scalar-ALU instructions does a CU issue per cycle (all four SIMDs busy with scalar work), and do
// scalar and vector instructions of different waves issue side by side?  (round 5: the strip kernels carry 65-150 scalar
// instructions per row step beside 110-280 vector ones.)
// build: hipcc --offload-arch=gfx950 -O3 tools/salu_rate.hip -o tools/salu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>   // 0 = scalar only, 1 = vector only, 2 = both interleaved in every wave
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE != 1) { REP16(asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 3\n\ts_add_u32 %2, %2, 5\n\ts_add_u32 %3, %3, 7" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");) }
        if (MODE != 0) { REP16(asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + (float)(s0 + s1 + s2 + s3);
}
int main()
{
    float* d; hipMalloc(&d, 256 * 4096 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2048;
    for (int wg_per_cu : {1, 2, 4}) {
        const int blocks = 256 * wg_per_cu;   // 256 CUs
        for (int mode = 0; mode < 3; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
                else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double per_wave = 64.0 * iters;                       // instructions of each kind per wave
            const double waves_per_cu = 4.0 * wg_per_cu;
            const double ns_per_inst_cu = ms * 1e6 / (per_wave * waves_per_cu);   // per CU, per instruction of one kind
            printf("%d workgroup(s) per CU, %s: %.3f ms -> %.3f ns per %s instruction per CU (%.2f per ns)\n", wg_per_cu,
                   mode == 0 ? "scalar only" : mode == 1 ? "vector only" : "scalar + vector", ms, ns_per_inst_cu, mode == 1 ? "vector" : "scalar (and vector)", 1.0 / ns_per_inst_cu);
        }
    }
    return 0;
}
