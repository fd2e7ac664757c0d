// frag_probe9.hip -- the library's placement search found arrangements that were fast when probed through the pool
// mapping and slow after the same physical pieces had been mapped into the final block.  So the virtual side matters
// too.  Here: a pool of 36 pieces (64 MiB) mapped once; find a fast and a slow 12-plane arrangement through pointer
// tables; then map EXACTLY those pieces, in that plane order, into fresh virtual ranges reserved with different
// alignments, several times each, and probe again.  The pieces stay mapped in the pool as well (two mappings of one
// allocation) in variant A and are unmapped from the pool first in variant B.
// Build: hipcc --offload-arch=gfx950 -O3 tools/frag_probe9.hip -o tools/frag_probe9
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int N = 4096, NPL = 12;
constexpr size_t PLANE_B = (size_t)N * N * 4;
struct Tab { float* p[NPL]; };
__global__ __launch_bounds__(256) void k_planes(Tab t, int strip_rows)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = (blockIdx.x * 4 + wv) * 64 + lane;
    const int y0 = blockIdx.y * strip_rows;
    for (int y = y0; y < y0 + strip_rows && y < N; ++y) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) __builtin_nontemporal_store((float)(y + p), t.p[p] + (size_t)y * N + x);
    }
}
static hipEvent_t ea, eb;
static double run(const Tab& t, int reps = 6)
{
    const int sr = 19;
    dim3 grid(N / 256, (N + sr - 1) / sr);
    for (int i = 0; i < 2; ++i) k_planes<<<grid, 256>>>(t, sr);
    CK(hipEventRecord(ea));
    for (int i = 0; i < reps; ++i) k_planes<<<grid, 256>>>(t, sr);
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    CK(hipGetLastError());
    return (double)N * N * 4.0 * NPL / (ms / reps) / 1e6;
}
static void access(void* va, size_t bytes)
{
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, bytes, &acc, 1));
}
int main()
{
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const int pool_n = 36;
    hipMemAllocationProp pr = {};
    pr.type = hipMemAllocationTypePinned;
    pr.location.type = hipMemLocationTypeDevice;
    std::vector<hipMemGenericAllocationHandle_t> pool(pool_n);
    for (auto& h : pool) CK(hipMemCreate(&h, PLANE_B, &pr, 0));
    void* pva; CK(hipMemAddressReserve(&pva, pool_n * PLANE_B, 2ull << 20, nullptr, 0));
    for (int i = 0; i < pool_n; ++i) CK(hipMemMap((char*)pva + (size_t)i * PLANE_B, PLANE_B, 0, pool[i], 0));
    access(pva, pool_n * PLANE_B);
    printf("pool va %p\n", pva);
    auto deal = [&](int b, std::vector<int>& out) {
        int lo = b - 1, hi = b;
        out.resize(NPL);
        for (int p = 0; p < NPL; ++p) { const bool from_hi = ((p & 1) && hi < pool_n) || lo < 0; out[p] = from_hi ? hi++ : lo--; }
    };
    auto tab_pool = [&](const std::vector<int>& arr) { Tab t; for (int p = 0; p < NPL; ++p) t.p[p] = (float*)((char*)pva + (size_t)arr[p] * PLANE_B); return t; };
    std::vector<std::vector<int>> cands;
    std::vector<double> sp;
    { std::vector<int> id(NPL); for (int p = 0; p < NPL; ++p) id[p] = p; cands.push_back(id); }
    for (int b = 6; b <= pool_n - 6; b += 2) { std::vector<int> a; deal(b, a); cands.push_back(a); }
    printf("through the pool mapping (GB/s):");
    for (auto& a : cands) { sp.push_back(run(tab_pool(a))); printf(" %.0f", sp.back()); }
    printf("\n");
    const int ifast = (int)(std::max_element(sp.begin(), sp.end()) - sp.begin()), islow = (int)(std::min_element(sp.begin(), sp.end()) - sp.begin());
    for (int which : {ifast, islow}) {
        const std::vector<int>& arr = cands[which];
        printf("\narrangement %d (%s, %.0f GB/s through the pool): pieces", which, which == ifast ? "fastest" : "slowest", sp[which]);
        for (int p : arr) printf(" %d", p);
        printf("\n");
        for (size_t align_mib : {2ull, 64ull, 1024ull})
            for (int rep = 0; rep < 3; ++rep) {
                void* va; CK(hipMemAddressReserve(&va, NPL * PLANE_B, align_mib << 20, nullptr, 0));
                // a second mapping of pieces that are still mapped in the pool may be refused: then skip
                bool ok = true;
                for (int p = 0; p < NPL && ok; ++p) ok = hipMemMap((char*)va + (size_t)p * PLANE_B, PLANE_B, 0, pool[arr[p]], 0) == hipSuccess;
                if (!ok) { printf("   second mapping refused\n"); (void)hipGetLastError(); CK(hipMemAddressFree(va, NPL * PLANE_B)); goto unmapped; }
                access(va, NPL * PLANE_B);
                Tab t; for (int p = 0; p < NPL; ++p) t.p[p] = (float*)((char*)va + (size_t)p * PLANE_B);
                printf("   mapped again at %p (alignment %4zu MiB, va %% 64 MiB = %2zu MiB): %.0f GB/s   [pool mapping now: %.0f]\n", va, (size_t)align_mib,
                       (size_t)(((size_t)va >> 20) & 63), run(t), run(tab_pool(arr)));
                CK(hipDeviceSynchronize());
                CK(hipMemUnmap(va, NPL * PLANE_B));
                CK(hipMemAddressFree(va, NPL * PLANE_B));
            }
    }
unmapped:
    // variant B: pool unmapped first, then the fastest arrangement mapped alone
    {
        const std::vector<int> arr = cands[ifast];
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(pva, pool_n * PLANE_B));
        CK(hipMemAddressFree(pva, pool_n * PLANE_B));
        for (size_t align_mib : {2ull, 64ull, 1024ull})
            for (int rep = 0; rep < 3; ++rep) {
                void* va; CK(hipMemAddressReserve(&va, NPL * PLANE_B, align_mib << 20, nullptr, 0));
                for (int p = 0; p < NPL; ++p) CK(hipMemMap((char*)va + (size_t)p * PLANE_B, PLANE_B, 0, pool[arr[p]], 0));
                access(va, NPL * PLANE_B);
                Tab t; for (int p = 0; p < NPL; ++p) t.p[p] = (float*)((char*)va + (size_t)p * PLANE_B);
                printf("pool unmapped; fastest arrangement at %p (alignment %4zu MiB, va %% 64 MiB = %2zu MiB): %.0f GB/s\n", va, (size_t)align_mib,
                       (size_t)(((size_t)va >> 20) & 63), run(t));
                CK(hipDeviceSynchronize());
                CK(hipMemUnmap(va, NPL * PLANE_B));
                CK(hipMemAddressFree(va, NPL * PLANE_B));
            }
        // ... and with the unused pieces released (what the library does)
        std::vector<char> used(pool_n, 0);
        for (int p : arr) used[p] = 1;
        for (int i = 0; i < pool_n; ++i) if (!used[i]) CK(hipMemRelease(pool[i]));
        for (int rep = 0; rep < 3; ++rep) {
            void* va; CK(hipMemAddressReserve(&va, NPL * PLANE_B, 2ull << 20, nullptr, 0));
            for (int p = 0; p < NPL; ++p) CK(hipMemMap((char*)va + (size_t)p * PLANE_B, PLANE_B, 0, pool[arr[p]], 0));
            access(va, NPL * PLANE_B);
            Tab t; for (int p = 0; p < NPL; ++p) t.p[p] = (float*)((char*)va + (size_t)p * PLANE_B);
            printf("unused pieces released; fastest arrangement at %p: %.0f GB/s\n", va, run(t));
            CK(hipDeviceSynchronize());
            CK(hipMemUnmap(va, NPL * PLANE_B));
            CK(hipMemAddressFree(va, NPL * PLANE_B));
        }
    }
    return 0;
}
