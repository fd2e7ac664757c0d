// tlb_probe.hip -- the fused kernels slow down once the planes they touch add up to more than ~4 GB (tools/size_trend.py:
// M5 on 67 Mpix 64 % against 78 % on 33 Mpix).  4 GiB is what 2048 translations of 2 MiB cover.  Test: 20 planes of
// P MiB each written by the streaming-store kernel, planes in (a) one hipMalloc block, (b) one physical allocation per
// plane mapped at a 2 MiB-aligned virtual address, (c) the same pieces at a virtual address aligned to 1 GiB (so that
// virtual and physical alignment agree up to the piece size and the page tables can use large fragments).
// Build: hipcc --offload-arch=gfx950 -O3 tools/tlb_probe.hip -o tools/tlb_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int NPL = 20;
struct Tab { float* p[NPL]; };
__global__ __launch_bounds__(256) void k_planes(Tab t, int rows, int cols, int strip_rows)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = (blockIdx.x * 4 + wv) * 64 + lane;
    if (x >= cols) return;
    const int y0 = blockIdx.y * strip_rows;
    for (int y = y0; y < y0 + strip_rows && y < rows; ++y) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) __builtin_nontemporal_store((float)(y + p), t.p[p] + (size_t)y * cols + x);
    }
}
static hipEvent_t ea, eb;
static double run(const Tab& t, int rows, int cols, int reps = 5)
{
    const int sr = 19;
    dim3 grid((cols + 255) / 256, (rows + sr - 1) / sr);
    for (int i = 0; i < 2; ++i) k_planes<<<grid, 256>>>(t, rows, cols, sr);
    CK(hipEventRecord(ea));
    for (int i = 0; i < reps; ++i) k_planes<<<grid, 256>>>(t, rows, cols, sr);
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    CK(hipGetLastError());
    return (double)rows * cols * 4.0 * NPL / (ms / reps) / 1e6;
}
int main()
{
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    hipMemAllocationProp pr = {};
    pr.type = hipMemAllocationTypePinned;
    pr.location.type = hipMemLocationTypeDevice;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int n : {4096, 5792, 8192, 11584}) {
        const size_t plane_b = ((size_t)n * n * 4 + (2u << 20) - 1) / (2u << 20) * (2u << 20);
        printf("%5d^2, 20 planes of %zu MiB = %.1f GB:", n, plane_b >> 20, 20.0 * plane_b / 1e9);
        {
            float* b; CK(hipMalloc(&b, NPL * plane_b));
            Tab t; for (int p = 0; p < NPL; ++p) t.p[p] = (float*)((char*)b + p * plane_b);
            printf("  hipMalloc %5.0f", run(t, n, n));
            CK(hipFree(b));
        }
        std::vector<hipMemGenericAllocationHandle_t> pc(NPL);
        for (auto& h : pc) CK(hipMemCreate(&h, plane_b, &pr, 0));
        size_t pow2 = 1; while (pow2 < plane_b) pow2 <<= 1;
        for (size_t align : {(size_t)2 << 20, pow2, (size_t)1 << 30}) {
            // each plane in a slot of `slot` bytes so that every plane starts at a multiple of the alignment under test
            const size_t slot = (plane_b + align - 1) / align * align;
            void* va; CK(hipMemAddressReserve(&va, NPL * slot, align, nullptr, 0));
            for (int p = 0; p < NPL; ++p) { CK(hipMemMap((char*)va + p * slot, plane_b, 0, pc[p], 0)); CK(hipMemSetAccess((char*)va + p * slot, plane_b, &acc, 1)); }
            Tab t; for (int p = 0; p < NPL; ++p) t.p[p] = (float*)((char*)va + p * slot);
            printf("  | pieces, planes at multiples of %4zu MiB: %5.0f", align >> 20, run(t, n, n));
            CK(hipDeviceSynchronize());
            for (int p = 0; p < NPL; ++p) CK(hipMemUnmap((char*)va + p * slot, plane_b));
            CK(hipMemAddressFree(va, NPL * slot));
            // fresh pieces for the next mapping: a piece is never mapped twice (tools/vmm_remap_check.hip)
            for (auto& h : pc) { CK(hipMemRelease(h)); CK(hipMemCreate(&h, plane_b, &pr, 0)); }
        }
        for (auto h : pc) CK(hipMemRelease(h));
        printf(" GB/s\n");
        fflush(stdout);
    }
    return 0;
}
