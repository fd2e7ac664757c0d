// tools/valu_rate.hip -- is v_pk_fma_f32 really twice the f32 rate of v_fma_f32 on this GPU?  (diagnostic)
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <bool PK>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters)
{
    if constexpr (PK) {
        f2 x[8];
        for (int i = 0; i < 8; ++i) x[i] = f2{(float)threadIdx.x + i, (float)i};
        const f2 va = {a, a}, vb = {b, b};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], va, vb);
        f2 s = x[0];
        for (int i = 1; i < 8; ++i) s += x[i];
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    } else {
        float x[16];
        for (int i = 0; i < 16; ++i) x[i] = (float)threadIdx.x + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], a, b);
        float s = x[0];
        for (int i = 1; i < 16; ++i) s += x[i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}
int main()
{
    float* d;
    hipMalloc(&d, 256 * 2048 * 4 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 256 * 8;
    for (int pk = 0; pk < 2; ++pk) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (pk) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = 2.0 * 16 * iters * 256.0 * blocks;   // both variants: 16 f32 fma per thread per iteration
            if (rep == 2) printf("%s: %.3f ms  %.1f TFLOP/s (f32 fma)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", ms, flop / ms / 1e9);
        }
    }
    return 0;
}
