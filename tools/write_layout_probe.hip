// write_layout_probe.hip -- round 5: what does the LAYOUT of the output planes (and the width of the store instruction) do to the
// rate at which "1 plane in (cached), NP planes out, nontemporal stores, 10-row strips of 64 columns per wave" streams?
// The traffic of the basis kernel with nothing but its loads and stores.  Variants:
//   planar      plane after plane (rounds 1-3)
//   rowint      [row][plane][column]               (round 4 default; dword stores, NP x 256 B per wave-row, 16 KiB apart at 4096 columns)
//   skew        rowint with every plane's row segment 256 B longer (no power-of-two distance between a wave's stores)
//   tile64      [row][64-column strip][plane][64]  (a wave-row writes NP x 256 B = one contiguous block)
//   tile256     [row][256-column block][plane][256] (a workgroup-row writes NP x 1 KiB contiguous)
//   tile64x4    tile64, but the block is written with dwordx4 stores (1 KiB per instruction, as after an LDS transpose)
//   aos         [row][column][plane]: dwordx4 + dwordx3 per pixel
//   linear      the same bytes as one float4 sweep (what a fill reaches)
// Build: hipcc --offload-arch=gfx950 -O3 tools/write_layout_probe.hip -o tools/write_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int N = 4096;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

enum { PLANAR, ROWINT, SKEW, TILE64, TILE256, TILE64X4, AOS };

template <int L, int NP>
__global__ __launch_bounds__(256) void k_strip(const float* __restrict__ in, float* __restrict__ out, int sr, size_t plane)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = (blockIdx.x * 4 + wv) * 64 + lane;
    const int y0 = blockIdx.y * sr;
    for (int y = y0; y < y0 + sr && y < N; ++y) {
        const float v = in[(size_t)y * N + x];
        if constexpr (L == TILE64X4) {
            float* seg = out + (size_t)y * NP * N + (size_t)(x >> 6) * NP * 64;
#pragma unroll
            for (int q = 0; q < (NP + 3) / 4; ++q) {
                const int planes = NP - 4 * q >= 4 ? 4 : NP - 4 * q;
                if (lane < planes * 16) {
                    f4 w = {v + q, v, v, v};
                    __builtin_nontemporal_store(w, reinterpret_cast<f4*>(seg + q * 256) + lane);
                }
            }
        } else if constexpr (L == AOS) {
            float* px = out + ((size_t)y * N + x) * NP;
#pragma unroll
            for (int q = 0; q < NP / 4; ++q) {
                f4 w = {v + q, v, v, v};
                __builtin_nontemporal_store(w, reinterpret_cast<f4u*>(px + 4 * q));   // 4-byte aligned dwordx4
            }
#pragma unroll
            for (int r = NP / 4 * 4; r < NP; ++r) __builtin_nontemporal_store(v + r, px + r);
        } else {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                size_t o;
                if constexpr (L == PLANAR) o = p * plane + (size_t)y * N + x;
                else if constexpr (L == ROWINT) o = ((size_t)y * NP + p) * N + x;
                else if constexpr (L == SKEW) o = ((size_t)y * NP + p) * (N + 64) + x;
                else if constexpr (L == TILE64) o = (size_t)y * NP * N + (size_t)(x >> 6) * NP * 64 + p * 64 + (x & 63);
                else o = (size_t)y * NP * N + (size_t)(x >> 8) * NP * 256 + p * 256 + (x & 255);
                __builtin_nontemporal_store(v + p, out + o);
            }
        }
    }
}

template <int NP>
__global__ __launch_bounds__(256) void k_linear(const f4* in, float* out, size_t n4)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f4 v = in[i];
#pragma unroll
        for (int p = 0; p < NP; ++p) { f4 w = {v.x + p, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<f4*>(out) + i * NP + p); }
    }
}

template <int NP>
static void run(const float* in, float* out, hipEvent_t a, hipEvent_t b)
{
    const size_t plane = (size_t)N * N;
    const char* names[] = {"planar", "rowint", "skew", "tile64", "tile256", "tile64x4", "aos"};
    const int srs[] = {10, 19};
    std::vector<std::vector<float>> t(7 * 2 + 1);
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
            const dim3 g(N / 256, (N + sr - 1) / sr);
            timeit(c++, [&] { k_strip<PLANAR, NP><<<g, 256>>>(in, out, sr, plane); });
            timeit(c++, [&] { k_strip<ROWINT, NP><<<g, 256>>>(in, out, sr, plane); });
            timeit(c++, [&] { k_strip<SKEW, NP><<<g, 256>>>(in, out, sr, plane); });
            timeit(c++, [&] { k_strip<TILE64, NP><<<g, 256>>>(in, out, sr, plane); });
            timeit(c++, [&] { k_strip<TILE256, NP><<<g, 256>>>(in, out, sr, plane); });
            timeit(c++, [&] { k_strip<TILE64X4, NP><<<g, 256>>>(in, out, sr, plane); });
            timeit(c++, [&] { k_strip<AOS, NP><<<g, 256>>>(in, out, sr, plane); });
        }
        timeit(c++, [&] { k_linear<NP><<<16384, 256>>>((const f4*)in, out, plane / 4); });
    }
    CK(hipGetLastError());
    const double bytes = plane * 4.0 * (NP + 1);
    auto med = [&](int c) { std::sort(t[c].begin(), t[c].end()); return t[c][t[c].size() / 2]; };
    int c = 0;
    for (int sr : srs)
        for (int l = 0; l < 7; ++l) {
            const float m = med(c++);
            printf("NP %2d strip %2d %-9s %.4f ms %6.0f GB/s (%4.1f %% of 8 TB/s)\n", NP, sr, names[l], m, bytes / m / 1e6, bytes / m / 8e7);
        }
    const float m = med(c++);
    printf("NP %2d          linear f4 %.4f ms %6.0f GB/s (%4.1f %%)\n", NP, m, bytes / m / 1e6, bytes / m / 8e7);
}

int main()
{
    const size_t plane = (size_t)N * N;
    float *in, *out;
    CK(hipMalloc(&in, plane * 4));
    CK(hipMalloc(&out, (plane + 64 * N) * 4 * 12));
    CK(hipMemset(in, 0, plane * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    run<7>(in, out, a, b);
    run<9>(in, out, a, b);
    run<12>(in, out, a, b);
    return 0;
}
