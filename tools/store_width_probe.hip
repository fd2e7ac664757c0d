// store_width_probe.hip -- round 5: does the WIDTH of a wave's store matter?  "1 plane in (cached), NP planes out, row-interleaved [row][plane][column],
// nontemporal stores, 10-row strips": a wave owns 64 columns and stores dwords (the product), 128 columns and stores dwordx2, or 256 columns and stores dwordx4
// (per plane row 256 B / 512 B / 1 KiB contiguous per instruction); workgroup = 4 waves side by side; with and without a cap of 3 workgroups per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/store_width_probe.hip -o tools/store_width_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int N = 4096;
template <int V> struct vec { typedef float type __attribute__((ext_vector_type(V))); };
template <> struct vec<1> { typedef float type; };

template <int V, int NP>
__global__ __launch_bounds__(256) void k_strip(const float* __restrict__ in, float* __restrict__ out, int sr)
{
    extern __shared__ float pad[];
    typedef typename vec<V>::type T;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = ((blockIdx.x * 4 + wv) * 64 + lane) * V;
    const int y0 = blockIdx.y * sr;
    for (int y = y0; y < y0 + sr && y < N; ++y) {
        const T v = *reinterpret_cast<const T*>(in + (size_t)y * N + x);
#pragma unroll
        for (int p = 0; p < NP; ++p) __builtin_nontemporal_store(v + (float)p, reinterpret_cast<T*>(out + ((size_t)y * NP + p) * N + x));
    }
}

template <int NP>
static void run(const float* in, float* out, hipEvent_t a, hipEvent_t b)
{
    std::vector<std::vector<float>> t(6);
    auto timeit = [&](int c, auto&& launch) {
        launch();
        CK(hipEventRecord(a));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        t[c].push_back(ms / 10);
    };
    const int sr = 10;
    const size_t cap3 = 46 * 1024;   // dynamic LDS nobody touches: three workgroups per CU
    for (int round = 0; round < 5; ++round) {
        int c = 0;
        for (size_t lds : {(size_t)0, cap3}) {
            timeit(c++, [&] { k_strip<1, NP><<<dim3(N / 256, (N + sr - 1) / sr), 256, lds>>>(in, out, sr); });
            timeit(c++, [&] { k_strip<2, NP><<<dim3(N / 512, (N + sr - 1) / sr), 256, lds>>>(in, out, sr); });
            timeit(c++, [&] { k_strip<4, NP><<<dim3(N / 1024, (N + sr - 1) / sr), 256, lds>>>(in, out, sr); });
        }
    }
    CK(hipGetLastError());
    const double bytes = (double)N * N * 4.0 * (NP + 1);
    const char* nm[6] = {"dword   64 columns per wave            ", "dwordx2 128 columns per wave            ", "dwordx4 256 columns per wave            ",
                         "dword   64 columns, 3 workgroups per CU", "dwordx2 128 columns, 3 workgroups per CU", "dwordx4 256 columns, 3 workgroups per CU"};
    for (int c = 0; c < 6; ++c) {
        std::sort(t[c].begin(), t[c].end());
        const float m = t[c][t[c].size() / 2];
        printf("NP %2d %s %.4f ms %6.0f GB/s (%4.1f %% of 8 TB/s)\n", NP, nm[c], m, bytes / m / 1e6, bytes / m / 8e7);
    }
}

int main()
{
    float *in, *out;
    CK(hipMalloc(&in, (size_t)N * N * 4));
    CK(hipMalloc(&out, (size_t)N * N * 4 * 12));
    CK(hipMemset(in, 0, (size_t)N * N * 4));
    CK(hipFuncSetAttribute((const void*)k_strip<1, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    run<7>(in, out, a, b);
    run<9>(in, out, a, b);
    run<12>(in, out, a, b);
    return 0;
}
