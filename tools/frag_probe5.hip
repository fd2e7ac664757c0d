// frag_probe5.hip -- hypothesis from frag_probe3: the 9-plane streaming write is slow (5.7 TB/s) when all nine planes
// lie in one physically contiguous run and fast (7.2 TB/s) when they are split between two runs that lie far apart
// (windows straddling a jump of the allocator were fast).  Test: group A = the first 5 planes, then a spacer
// allocation of D bytes (kept alive), then group B = the other 4 planes.  Also three groups, and write-only.
// Build: hipcc --offload-arch=gfx950 -O3 tools/frag_probe5.hip -o tools/frag_probe5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int N = 4096, NPL = 9;
constexpr size_t PLANE_B = (size_t)N * N * 4;
struct Tab { float* p[NPL]; };

template <bool READ>
__global__ __launch_bounds__(256) void k_planes(const float* in, Tab t, int strip_rows)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = (blockIdx.x * 4 + wv) * 64 + lane;
    const int y0 = blockIdx.y * strip_rows;
    for (int y = y0; y < y0 + strip_rows && y < N; ++y) {
        const float v = READ ? in[(size_t)y * N + x] : (float)y;
#pragma unroll
        for (int p = 0; p < NPL; ++p) __builtin_nontemporal_store(v + p, t.p[p] + (size_t)y * N + x);
    }
}

template <bool READ>
static double run(const float* in, const Tab& t, int reps = 16)
{
    static hipEvent_t a = nullptr, b = nullptr;
    if (!a) { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    const int sr = 19;
    dim3 grid(N / 256, (N + sr - 1) / sr);
    for (int i = 0; i < 3; ++i) k_planes<READ><<<grid, 256>>>(in, t, sr);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) k_planes<READ><<<grid, 256>>>(in, t, sr);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return (double)N * N * 4.0 * (NPL + (READ ? 1 : 0)) / (ms / reps) / 1e6;
}

int main()
{
    float* in; CK(hipMalloc(&in, PLANE_B));
    CK(hipMemset(in, 0, PLANE_B));
    void* burn; CK(hipMalloc(&burn, 6ull << 30));  // leave the fragmented start of the heap behind
    auto report = [&](const char* what, const Tab& t) {
        printf("%-72s %7.1f GB/s   write-only %7.1f GB/s\n", what, run<true>(in, t), run<false>(in, t));
        fflush(stdout);
    };
    char nm[128];
    for (size_t d_mib : {0ull, 64ull, 1024ull, 4096ull, 16384ull, 65536ull}) {
        Tab t;
        std::vector<void*> keep;
        for (int p = 0; p < 5; ++p) CK(hipMalloc(&t.p[p], PLANE_B));
        void* sp = nullptr;
        if (d_mib) CK(hipMalloc(&sp, d_mib << 20));
        for (int p = 5; p < NPL; ++p) CK(hipMalloc(&t.p[p], PLANE_B));
        snprintf(nm, sizeof nm, "5 planes | spacer %6zu MiB | 4 planes  (separate hipMallocs)", (size_t)d_mib);
        report(nm, t);
        if (sp) { CK(hipFree(sp)); snprintf(nm, sizeof nm, "   ... spacer freed"); report(nm, t); }
        for (int p = 0; p < NPL; ++p) CK(hipFree(t.p[p]));
    }
    // three groups
    for (size_t d_mib : {4096ull, 32768ull}) {
        Tab t;
        void *s1, *s2;
        for (int p = 0; p < 3; ++p) CK(hipMalloc(&t.p[p], PLANE_B));
        CK(hipMalloc(&s1, d_mib << 20));
        for (int p = 3; p < 6; ++p) CK(hipMalloc(&t.p[p], PLANE_B));
        CK(hipMalloc(&s2, d_mib << 20));
        for (int p = 6; p < NPL; ++p) CK(hipMalloc(&t.p[p], PLANE_B));
        snprintf(nm, sizeof nm, "3 planes | %zu MiB | 3 planes | %zu MiB | 3 planes", (size_t)d_mib, (size_t)d_mib);
        report(nm, t);
        CK(hipFree(s1)); CK(hipFree(s2));
        for (int p = 0; p < NPL; ++p) CK(hipFree(t.p[p]));
    }
    // every plane isolated by 4 GiB
    {
        Tab t;
        std::vector<void*> sp;
        for (int p = 0; p < NPL; ++p) { CK(hipMalloc(&t.p[p], PLANE_B)); void* s; CK(hipMalloc(&s, 4096ull << 20)); sp.push_back(s); }
        report("every plane followed by a 4 GiB spacer", t);
        for (void* s : sp) CK(hipFree(s));
        for (int p = 0; p < NPL; ++p) CK(hipFree(t.p[p]));
    }
    // one block: planes 0-4 at the start, planes 5-8 at an offset of D inside a 70 GiB hipMalloc
    {
        float* big; CK(hipMalloc(&big, 70ull << 30));
        for (size_t d_mib : {320ull, 1024ull, 4096ull, 16384ull, 65536ull}) {
            Tab t;
            for (int p = 0; p < 5; ++p) t.p[p] = big + (size_t)p * PLANE_B / 4;
            for (int p = 5; p < NPL; ++p) t.p[p] = big + ((d_mib << 20) + (size_t)(p - 5) * PLANE_B) / 4;
            snprintf(nm, sizeof nm, "one 70 GiB hipMalloc: planes 0-4 at 0, planes 5-8 at +%zu MiB", (size_t)d_mib);
            report(nm, t);
        }
        // sliding both groups together through the block: is the block uniform?
        for (size_t base_gib : {0ull, 8ull, 16ull, 24ull, 32ull, 40ull, 48ull, 56ull, 64ull}) {
            Tab t;
            for (int p = 0; p < NPL; ++p) t.p[p] = big + ((base_gib << 30) + (size_t)p * PLANE_B) / 4;
            snprintf(nm, sizeof nm, "one 70 GiB hipMalloc: nine consecutive planes at +%zu GiB", (size_t)base_gib);
            report(nm, t);
        }
        CK(hipFree(big));
    }
    return 0;
}
