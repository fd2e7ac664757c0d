// membench.hip -- store/load pattern micro-benchmarks for MI355X: what access shape does the
// HBM system want for "1 plane in, 7 planes out" (the basis kernel's traffic)?
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int N = 4096;          // rows = cols
constexpr int NP = 7;

// P1: ideal linear streaming: each thread float4, block = 4 KiB contiguous, grid-stride
__global__ __launch_bounds__(256) void p1_linear(const float4* in, float4* out, size_t n4, size_t plane4)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = in[i];
#pragma unroll
        for (int p = 0; p < NP; ++p) { float4 w = v; w.x += p; out[p * plane4 + i] = w; }
    }
}

// P2: wave marches down a 64-col strip, dword per lane (K1 today). block = 4 adjacent strips.
template <bool NT>
__global__ __launch_bounds__(256) void p2_strip(const float* in, float* out, int strip_rows, size_t plane)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = (blockIdx.x * 4 + wv) * 64 + lane;
    const int y0 = blockIdx.y * strip_rows;
    for (int y = y0; y < y0 + strip_rows && y < N; ++y) {
        float v = in[(size_t)y * N + x];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float* d = out + p * plane + (size_t)y * N + x;
            if (NT) __builtin_nontemporal_store(v + p, d); else *d = v + p;
        }
    }
}

// P3: wave marches down a (64*V)-col strip, V floats per lane (8 or 16 B stores)
template <int V, bool NT>
__global__ __launch_bounds__(256) void p3_strip_vec(const float* in, float* out, int strip_rows, size_t plane)
{
    typedef float vec __attribute__((ext_vector_type(V)));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = ((blockIdx.x * 4 + wv) * 64 + lane) * V;
    const int y0 = blockIdx.y * strip_rows;
    for (int y = y0; y < y0 + strip_rows && y < N; ++y) {
        vec v = *reinterpret_cast<const vec*>(in + (size_t)y * N + x);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            vec w = v; w[0] += p;
            vec* d = reinterpret_cast<vec*>(out + p * plane + (size_t)y * N + x);
            if (NT) __builtin_nontemporal_store(w, d); else *d = w;
        }
    }
}

// P4: block of 256 threads covers 256 cols x R rows tile, threads iterate rows; stores dword but the
// 4 waves write 1 KiB contiguous (same as P2) -- variant: block = 4 waves stacked vertically
__global__ __launch_bounds__(256) void p4_vert(const float* in, float* out, int strip_rows, size_t plane)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const int y0 = (blockIdx.y * 4 + wv) * strip_rows;
    for (int y = y0; y < y0 + strip_rows && y < N; ++y) {
        float v = in[(size_t)y * N + x];
#pragma unroll
        for (int p = 0; p < NP; ++p) out[p * plane + (size_t)y * N + x] = v + p;
    }
}

// P5: write-only (no read), strip pattern dword
__global__ __launch_bounds__(256) void p5_wonly(float* out, int strip_rows, size_t plane)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = (blockIdx.x * 4 + wv) * 64 + lane;
    const int y0 = blockIdx.y * strip_rows;
    for (int y = y0; y < y0 + strip_rows && y < N; ++y) {
#pragma unroll
        for (int p = 0; p < NP; ++p) out[p * plane + (size_t)y * N + x] = (float)(y + p);
    }
}

// P1nt: linear streaming with nontemporal stores
__global__ __launch_bounds__(256) void p1_linear_nt(const float4* in, float4* out, size_t n4, size_t plane4)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = in[i];
#pragma unroll
        for (int p = 0; p < NP; ++p) { f4 w = {v.x + p, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<f4*>(out + p * plane4 + i)); }
    }
}

// P5nt: write-only strips, nontemporal
template <int V>
__global__ __launch_bounds__(256) void p5_wonly_nt(float* out, int strip_rows, size_t plane)
{
    typedef float vec __attribute__((ext_vector_type(V)));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = ((blockIdx.x * 4 + wv) * 64 + lane) * V;
    const int y0 = blockIdx.y * strip_rows;
    for (int y = y0; y < y0 + strip_rows && y < N; ++y) {
#pragma unroll
        for (int p = 0; p < NP; ++p) { vec w; for (int k = 0; k < V; ++k) w[k] = (float)(y + p + k); __builtin_nontemporal_store(w, reinterpret_cast<vec*>(out + p * plane + (size_t)y * N + x)); }
    }
}

// P9: read-only strips (7 planes in), to see the read ceiling in the same shape
__global__ __launch_bounds__(256) void p9_ronly(const float* in, float* sink, int strip_rows, size_t plane)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = (blockIdx.x * 4 + wv) * 64 + lane;
    const int y0 = blockIdx.y * strip_rows;
    float acc = 0.f;
    for (int y = y0; y < y0 + strip_rows && y < N; ++y) {
#pragma unroll
        for (int p = 0; p < NP; ++p) acc += in[p * plane + (size_t)y * N + x];
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// P10: K1-like strips writing NPL planes, separate-plane layout vs 64-column blocked interleave
// ([row][xblk][plane][64]): one wave-row = NPL*256 contiguous bytes.
template <int NPL, bool BLOCKED>
__global__ __launch_bounds__(256) void p10_planes(const float* in, float* out, int strip_rows, int rows, int cols)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int xblk = blockIdx.x * 4 + wv;
    if (xblk * 64 >= cols) return;
    const int x = xblk * 64 + lane;
    const int y0 = blockIdx.y * strip_rows;
    const size_t plane = (size_t)rows * cols;
    for (int y = y0; y < y0 + strip_rows && y < rows; ++y) {
        float v = in[(size_t)y * cols + x];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            float* d = BLOCKED ? out + ((size_t)y * (cols / 64) + xblk) * (NPL * 64) + p * 64 + lane
                               : out + p * plane + (size_t)y * cols + x;
            __builtin_nontemporal_store(v + p, d);
        }
    }
}

// P6: pure copy float4 (1 in, 1 out)
__global__ __launch_bounds__(256) void p6_copy(const float4* in, float4* out, size_t n4)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

// P7: strip pattern where a wave covers 64 cols but processes RB rows per iteration with lanes
// remapped so each store instruction writes 4 rows x 64 B?  (worse locality; control)
// P8: transposed-through-registers: wave owns 256 cols x strip; each lane float4 (=P3<4>)

template <class F>
static float timeit(F f, int reps = 20)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / reps;
}

int main()
{
    const size_t plane = (size_t)N * N;
    float *in, *out;
    CK(hipMalloc(&in, plane * 4)); CK(hipMalloc(&out, plane * 4 * NP));
    std::vector<float> h(plane); for (size_t i = 0; i < plane; ++i) h[i] = (float)(i % 977) * 1e-3f;
    CK(hipMemcpy(in, h.data(), plane * 4, hipMemcpyHostToDevice));
    const double bytes8 = plane * 4.0 * 8, bytes7 = plane * 4.0 * 7, bytes2 = plane * 4.0 * 2;
    auto rep = [&](const char* name, float ms, double bytes) { printf("%-44s %8.4f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6); };

    for (int g : {1024, 2048, 4096, 8192})
        { char nm[64]; snprintf(nm, 64, "P6 copy float4 grid=%d", g); rep(nm, timeit([&] { p6_copy<<<g, 256>>>((const float4*)in, (float4*)out, plane / 4); }), bytes2); }
    for (int g : {1024, 2048, 4096, 8192, 16384})
        { char nm[64]; snprintf(nm, 64, "P1 linear 1in/7out float4 grid=%d", g); rep(nm, timeit([&] { p1_linear<<<g, 256>>>((const float4*)in, (float4*)out, plane / 4, plane / 4); }), bytes8); }
    for (int sr : {16, 32, 64, 128, 256, 512}) {
        char nm[64];
        snprintf(nm, 64, "P2 strip dword sr=%d", sr); rep(nm, timeit([&] { p2_strip<false><<<dim3(N / 256, (N + sr - 1) / sr), 256>>>(in, out, sr, plane); }), bytes8);
        snprintf(nm, 64, "P2 strip dword NT sr=%d", sr); rep(nm, timeit([&] { p2_strip<true><<<dim3(N / 256, (N + sr - 1) / sr), 256>>>(in, out, sr, plane); }), bytes8);
    }
    for (int sr : {16, 32, 64, 128, 256}) {
        char nm[64];
        snprintf(nm, 64, "P3 strip float2 sr=%d", sr); rep(nm, timeit([&] { p3_strip_vec<2, false><<<dim3(N / 512, (N + sr - 1) / sr), 256>>>(in, out, sr, plane); }), bytes8);
        snprintf(nm, 64, "P3 strip float4 sr=%d", sr); rep(nm, timeit([&] { p3_strip_vec<4, false><<<dim3(N / 1024, (N + sr - 1) / sr), 256>>>(in, out, sr, plane); }), bytes8);
        snprintf(nm, 64, "P3 strip float4 NT sr=%d", sr); rep(nm, timeit([&] { p3_strip_vec<4, true><<<dim3(N / 1024, (N + sr - 1) / sr), 256>>>(in, out, sr, plane); }), bytes8);
    }
    for (int g : {2048, 8192, 16384, 65536})
        { char nm[64]; snprintf(nm, 64, "P1nt linear 1in/7out float4 NT grid=%d", g); rep(nm, timeit([&] { p1_linear_nt<<<g, 256>>>((const float4*)in, (float4*)out, plane / 4, plane / 4); }), bytes8); }
    for (int sr : {8, 16, 32, 64}) {
        char nm[64];
        snprintf(nm, 64, "P5nt write-only dword NT sr=%d", sr); rep(nm, timeit([&] { p5_wonly_nt<1><<<dim3(N / 256, (N + sr - 1) / sr), 256>>>(out, sr, plane); }), bytes7);
        snprintf(nm, 64, "P5nt write-only float4 NT sr=%d", sr); rep(nm, timeit([&] { p5_wonly_nt<4><<<dim3(N / 1024, (N + sr - 1) / sr), 256>>>(out, sr, plane); }), bytes7);
        snprintf(nm, 64, "P9 read-only 7 planes dword sr=%d", sr); rep(nm, timeit([&] { p9_ronly<<<dim3(N / 256, (N + sr - 1) / sr), 256>>>(out, in, sr, plane); }), bytes7);
        snprintf(nm, 64, "P2 strip dword NT sr=%d", sr); rep(nm, timeit([&] { p2_strip<true><<<dim3(N / 256, (N + sr - 1) / sr), 256>>>(in, out, sr, plane); }), bytes8);
    }
    {
        float* big; CK(hipMalloc(&big, plane * 4 * 20));
        for (int sr : {10, 19, 64}) {
            char nm[80];
#define P10(NPL, R, C) \
            snprintf(nm, 80, "P10 %2d planes separate %dx%d sr=%d", NPL, R, C, sr); rep(nm, timeit([&] { p10_planes<NPL, false><<<dim3((C / 64 + 3) / 4, (R + sr - 1) / sr), 256>>>(in, big, sr, R, C); }), (double)R * C * 4 * (NPL + 1)); \
            snprintf(nm, 80, "P10 %2d planes blocked  %dx%d sr=%d", NPL, R, C, sr); rep(nm, timeit([&] { p10_planes<NPL, true><<<dim3((C / 64 + 3) / 4, (R + sr - 1) / sr), 256>>>(in, big, sr, R, C); }), (double)R * C * 4 * (NPL + 1));
            P10(7, 4096, 4096) P10(12, 4096, 4096) P10(20, 4096, 4096) P10(12, 4320, 1920) P10(20, 4320, 1920)
        }
        CK(hipFree(big));
    }
    {   // rotating inputs: 8 distinct 64 MiB images so the input read cannot stay in the Infinity Cache
        float* ins; CK(hipMalloc(&ins, plane * 4 * 8));
        CK(hipMemset(ins, 0, plane * 4 * 8));
        for (int sr : {8, 16, 32, 64}) {
            char nm[80]; int it = 0;
            snprintf(nm, 80, "P2rot strip dword NT sr=%d (8 rotating inputs)", sr);
            rep(nm, timeit([&] { p2_strip<true><<<dim3(N / 256, (N + sr - 1) / sr), 256>>>(ins + (size_t)((it++) & 7) * plane, out, sr, plane); }, 24), bytes8);
            snprintf(nm, 80, "P3rot strip float4 NT sr=%d (8 rotating inputs)", sr);
            rep(nm, timeit([&] { p3_strip_vec<4, true><<<dim3(N / 1024, (N + sr - 1) / sr), 256>>>(ins + (size_t)((it++) & 7) * plane, out, sr, plane); }, 24), bytes8);
        }
        CK(hipFree(ins));
    }
    for (int sr : {16, 64, 256}) {
        char nm[64];
        snprintf(nm, 64, "P4 vertical-block dword sr=%d", sr); rep(nm, timeit([&] { p4_vert<<<dim3(N / 64, (N + 4 * sr - 1) / (4 * sr)), 256>>>(in, out, sr, plane); }), bytes8);
        snprintf(nm, 64, "P5 write-only strip dword sr=%d", sr); rep(nm, timeit([&] { p5_wonly<<<dim3(N / 256, (N + sr - 1) / sr), 256>>>(out, sr, plane); }), bytes7);
    }
    return 0;
}
