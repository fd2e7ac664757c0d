// strip_sweep.hip -- how does the height of the strips (rows a wave writes before it ends) change what the memory
// system delivers for "1 plane in (cached), 9 planes out, nontemporal dword stores" -- the headline kernel's traffic
// with nothing but loads and stores?  Also: the same bytes as a linear sweep (float4, 4 KB per workgroup), and waves
// stacked vertically in a workgroup (4 x strip height per workgroup, 64 columns) instead of side by side.
// Build: hipcc --offload-arch=gfx950 -O3 tools/strip_sweep.hip -o tools/strip_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int N = 4096, NP = 9;

template <bool VERT>
__global__ __launch_bounds__(256) void k_strip(const float* in, float* out, int sr, size_t plane)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = VERT ? blockIdx.x * 64 + lane : (blockIdx.x * 4 + wv) * 64 + lane;
    const int y0 = VERT ? (blockIdx.y * 4 + wv) * sr : blockIdx.y * sr;
    for (int y = y0; y < y0 + sr && y < N; ++y) {
        const float v = in[(size_t)y * N + x];
#pragma unroll
        for (int p = 0; p < NP; ++p) __builtin_nontemporal_store(v + p, out + p * plane + (size_t)y * N + x);
    }
}
__global__ __launch_bounds__(256) void k_linear(const float4* in, float* out, size_t n4, size_t plane)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = in[i];
#pragma unroll
        for (int p = 0; p < NP; ++p) { f4 w = {v.x + p, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<f4*>(out + p * plane) + i); }
    }
}
int main()
{
    const size_t plane = (size_t)N * N;
    float *in, *out;
    CK(hipMalloc(&in, plane * 4)); CK(hipMalloc(&out, plane * 4 * NP));
    CK(hipMemset(in, 0, plane * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int srs[] = {2, 3, 4, 5, 6, 8, 10, 12, 16, 19, 24, 32, 40, 64};
    const int ncfg = 14 * 2 + 3;
    std::vector<std::vector<float>> t(ncfg);
    auto timeit = [&](int c, auto&& launch) {
        launch();
        CK(hipEventRecord(a));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        t[c].push_back(ms / 10);
    };
    for (int round = 0; round < 5; ++round) {
        int c = 0;
        for (int sr : srs) {
            timeit(c++, [&] { k_strip<false><<<dim3(N / 256, (N + sr - 1) / sr), 256>>>(in, out, sr, plane); });
            timeit(c++, [&] { k_strip<true><<<dim3(N / 64, (N + 4 * sr - 1) / (4 * sr)), 256>>>(in, out, sr, plane); });
        }
        for (int g : {4096, 16384, 65536}) timeit(c++, [&] { k_linear<<<g, 256>>>((const float4*)in, out, plane / 4, plane); });
    }
    CK(hipGetLastError());
    const double bytes = plane * 4.0 * (NP + 1);
    auto med = [&](int c) { std::sort(t[c].begin(), t[c].end()); return t[c][t[c].size() / 2]; };
    int c = 0;
    for (int sr : srs) {
        const float m0 = med(c++), m1 = med(c++);
        printf("strip rows %3d: side by side %.4f ms %6.0f GB/s (%4.1f %%) | stacked %.4f ms %6.0f GB/s (%4.1f %%)\n", sr, m0, bytes / m0 / 1e6,
               bytes / m0 / 8e7, m1, bytes / m1 / 1e6, bytes / m1 / 8e7);
    }
    for (int g : {4096, 16384, 65536}) { const float m = med(c++); printf("linear float4, grid %5d: %.4f ms %6.0f GB/s (%4.1f %%)\n", g, m, bytes / m / 1e6, bytes / m / 8e7); }
    return 0;
}
