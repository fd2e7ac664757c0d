// frag_probe8.hip -- can the fast mode be had BY CONSTRUCTION?  frag_probe7: nine planes picked from two different
// "runs" of consecutively created physical pieces write at 7.2 TB/s, nine planes from one run at 5.7.  The VRAM
// allocator serves a request from the smallest free block that fits, so requests of very different sizes come from
// different places: here the even planes are sub-ranges of ONE large physical allocation (hipMemCreate of 1-4 GiB,
// mapped plane by plane with hipMemMap's offset) and the odd planes are individual 64 MiB allocations.
// Build: hipcc --offload-arch=gfx950 -O3 tools/frag_probe8.hip -o tools/frag_probe8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int N = 4096, MAXP = 20;
constexpr size_t PLANE_B = (size_t)N * N * 4, PE = PLANE_B / 4;
struct Tab { float* p[MAXP]; };
template <int NPL>
__global__ __launch_bounds__(256) void k_planes(const float* in, Tab t, int strip_rows)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = (blockIdx.x * 4 + wv) * 64 + lane;
    const int y0 = blockIdx.y * strip_rows;
    for (int y = y0; y < y0 + strip_rows && y < N; ++y) {
        const float v = in[(size_t)y * N + x];
#pragma unroll
        for (int p = 0; p < NPL; ++p) __builtin_nontemporal_store(v + p, t.p[p] + (size_t)y * N + x);
    }
}
static hipEvent_t ea, eb;
template <int NPL>
static double run(const float* in, const Tab& t, int reps = 12)
{
    const int sr = 19;
    dim3 grid(N / 256, (N + sr - 1) / sr);
    for (int i = 0; i < 2; ++i) k_planes<NPL><<<grid, 256>>>(in, t, sr);
    CK(hipEventRecord(ea));
    for (int i = 0; i < reps; ++i) k_planes<NPL><<<grid, 256>>>(in, t, sr);
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    CK(hipGetLastError());
    return (double)N * N * 4.0 * (NPL + 1) / (ms / reps) / 1e6;
}
static hipMemAllocationProp prop()
{
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    return p;
}
static void access(void* va, size_t bytes)
{
    hipMemAccessDesc acc = {};
    acc.location = prop().location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, bytes, &acc, 1));
}
int main(int argc, char** argv)
{
    const size_t burn_gib = argc > 1 ? atoi(argv[1]) : 0;
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    float* in; CK(hipMalloc(&in, PLANE_B));
    CK(hipMemset(in, 0, PLANE_B));
    void* burn = nullptr;
    if (burn_gib) CK(hipMalloc(&burn, burn_gib << 30));
    const hipMemAllocationProp pr = prop();
    auto report = [&](const char* what, const Tab& t) {
        printf("%-78s 9pl %6.0f  12pl %6.0f  20pl %6.0f GB/s\n", what, run<9>(in, t), run<12>(in, t), run<20>(in, t));
        fflush(stdout);
    };
    char nm[160];
    // hipMemMap does not take an offset on this runtime, so every plane is its own allocation.  Group A = 5 pieces, then a
    // temporary allocation of D bytes (kept / released), then group B = 5 pieces; planes alternate A, B, A, B ...
    for (int rep = 0; rep < 2; ++rep)
    for (size_t d_mib : {0ull, 64ull, 1024ull, 4096ull, 16384ull, 65536ull}) {
        for (int keep = 1; keep >= 0; --keep) {
            if (!d_mib && !keep) continue;
            std::vector<hipMemGenericAllocationHandle_t> A(10), B(10);
            hipMemGenericAllocationHandle_t T = 0;
            for (auto& h : A) CK(hipMemCreate(&h, PLANE_B, &pr, 0));
            if (d_mib) CK(hipMemCreate(&T, d_mib << 20, &pr, 0));
            if (d_mib && !keep) { CK(hipMemRelease(T)); T = 0; }
            for (auto& h : B) CK(hipMemCreate(&h, PLANE_B, &pr, 0));
            void* va; CK(hipMemAddressReserve(&va, MAXP * PLANE_B, 2ull << 20, nullptr, 0));
            Tab t;
            for (int p = 0; p < MAXP; ++p) {
                void* at = (char*)va + (size_t)p * PLANE_B;
                CK(hipMemMap(at, PLANE_B, 0, (p & 1) ? B[p / 2] : A[p / 2], 0));
                t.p[p] = (float*)at;
            }
            access(va, MAXP * PLANE_B);
            snprintf(nm, sizeof nm, "10 pieces | %5zu MiB temporary (%s) | 10 pieces; planes alternate between the groups", (size_t)d_mib,
                     keep ? "kept" : "released before group B");
            report(nm, t);
            Tab u = t;
            for (int p = 0; p < 10; ++p) { u.p[p] = t.p[2 * p]; u.p[10 + p] = t.p[2 * p + 1]; }
            report("   same memory, planes 0-9 = group A, 10-19 = group B", u);
            CK(hipDeviceSynchronize());
            CK(hipMemUnmap(va, MAXP * PLANE_B));
            CK(hipMemAddressFree(va, MAXP * PLANE_B));
            for (auto h : A) CK(hipMemRelease(h));
            for (auto h : B) CK(hipMemRelease(h));
            if (T) CK(hipMemRelease(T));
        }
    }
    // reference: hipMalloc
    {
        float* b; CK(hipMalloc(&b, MAXP * PLANE_B));
        Tab t; for (int p = 0; p < MAXP; ++p) t.p[p] = b + (size_t)p * PE;
        report("hipMalloc, 20 planes back to back", t);
    }
    return 0;
}
