// frag_probe7.hip -- nine planes chosen from a pool of consecutively created 64 MiB physical pieces: consecutive,
// regularly spaced, two groups, one outlier, and RANDOM subsets.  Which choices give the fast mode?
// Build: hipcc --offload-arch=gfx950 -O3 tools/frag_probe7.hip -o tools/frag_probe7
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int N = 4096, NPL = 9;
constexpr size_t PLANE_B = (size_t)N * N * 4, PE = PLANE_B / 4;
struct Tab { float* p[NPL]; };
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
static double run(const float* in, const Tab& t, int reps = 12)
{
    const int sr = 19;
    dim3 grid(N / 256, (N + sr - 1) / sr);
    for (int i = 0; i < 2; ++i) k_planes<<<grid, 256>>>(in, t, sr);
    CK(hipEventRecord(ea));
    for (int i = 0; i < reps; ++i) k_planes<<<grid, 256>>>(in, t, sr);
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    CK(hipGetLastError());
    return (double)N * N * 4.0 * (NPL + 1) / (ms / reps) / 1e6;
}
int main(int argc, char** argv)
{
    const int slots = argc > 1 ? atoi(argv[1]) : 128;
    const size_t burn_gib = argc > 2 ? atoi(argv[2]) : 6;
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    float* in; CK(hipMalloc(&in, PLANE_B));
    CK(hipMemset(in, 0, PLANE_B));
    void* burn = nullptr;
    if (burn_gib) CK(hipMalloc(&burn, burn_gib << 30));
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, (size_t)slots * PLANE_B, 2ull << 20, nullptr, 0));
    for (int i = 0; i < slots; ++i) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, PLANE_B, &p, 0));
        CK(hipMemMap((char*)va + (size_t)i * PLANE_B, PLANE_B, 0, h, 0));
    }
    hipMemAccessDesc acc = {};
    acc.location = p.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, (size_t)slots * PLANE_B, &acc, 1));
    float* v = (float*)va;
    auto test = [&](const char* what, const std::vector<int>& sl) {
        Tab t;
        for (int i = 0; i < NPL; ++i) t.p[i] = v + (size_t)sl[i] * PE;
        printf("%-34s [", what);
        for (int i = 0; i < NPL; ++i) printf("%3d%s", sl[i], i + 1 < NPL ? " " : "");
        printf("]  %6.0f GB/s\n", run(in, t));
        fflush(stdout);
    };
    // scan for run boundaries first (windows of 9, step 1) so that the rest can be read against them
    printf("window scan (9 consecutive, start k):");
    for (int k = 0; k + 9 <= slots; ++k) {
        Tab t; for (int i = 0; i < NPL; ++i) t.p[i] = v + (size_t)(k + i) * PE;
        printf("%s%3d:%4.0f", k % 12 ? " " : "\n", k, run(in, t, 6) / 10);
    }
    printf("\n(values in 10 GB/s)\n\n");
    const int b = slots / 2;
    test("consecutive", {b, b + 1, b + 2, b + 3, b + 4, b + 5, b + 6, b + 7, b + 8});
    test("stride 3", {b, b + 3, b + 6, b + 9, b + 12, b + 15, b + 18, b + 21, b + 24});
    test("stride 5", {b, b + 5, b + 10, b + 15, b + 20, b + 25, b + 30, b + 35, b + 40});
    test("two groups, 30 apart", {b, b + 1, b + 2, b + 3, b + 4, b + 34, b + 35, b + 36, b + 37});
    test("one outlier 40 away", {b, b + 1, b + 2, b + 3, b + 4, b + 5, b + 6, b + 7, b + 47});
    test("irregular gaps", {b, b + 1, b + 3, b + 6, b + 10, b + 15, b + 21, b + 28, b + 36});
    test("fibonacci", {b, b + 1, b + 2, b + 3, b + 5, b + 8, b + 13, b + 21, b + 34});
    std::mt19937 rng(7);
    std::vector<int> all(slots);
    for (int i = 0; i < slots; ++i) all[i] = i;
    for (int trial = 0; trial < 16; ++trial) {
        std::shuffle(all.begin(), all.end(), rng);
        std::vector<int> sl(all.begin(), all.begin() + NPL);
        test("random subset (unsorted)", sl);
        std::sort(sl.begin(), sl.end());
        test("   same subset, sorted", sl);
    }
    return 0;
}
