// frag_probe10.hip -- frag_probe9: the SAME physical pieces stream at 5.3 TB/s through one virtual mapping and at
// 6.85 TB/s through another.  So: twelve pieces (64 MiB each), one large reserved virtual range, and the 12-plane
// block mapped at offset j x 64 MiB for j = 0, 1, 2, ... -- speed as a function of the virtual address alone.
// Build: hipcc --offload-arch=gfx950 -O3 tools/frag_probe10.hip -o tools/frag_probe10
#include <hip/hip_runtime.h>
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
static double run(const Tab& t, int reps = 5)
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
int main(int argc, char** argv)
{
    const int nslots = argc > 1 ? atoi(argv[1]) : 160;
    const size_t step_mib = argc > 2 ? atoi(argv[2]) : 64;
    const size_t align_gib = argc > 3 ? atoi(argv[3]) : 16;
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    hipMemAllocationProp pr = {};
    pr.type = hipMemAllocationTypePinned;
    pr.location.type = hipMemLocationTypeDevice;
    std::vector<hipMemGenericAllocationHandle_t> pc(NPL);
    for (auto& h : pc) CK(hipMemCreate(&h, PLANE_B, &pr, 0));
    const size_t span = (size_t)nslots * (step_mib << 20) + NPL * PLANE_B;
    void* base; CK(hipMemAddressReserve(&base, span, align_gib << 30, nullptr, 0));
    printf("reserved %zu MiB of virtual range at %p (alignment %zu GiB); 12 planes mapped at base + j x %zu MiB; GB/s per j:\n", span >> 20, base, align_gib, step_mib);
    void* alias; CK(hipMemAddressReserve(&alias, NPL * PLANE_B, 2ull << 20, nullptr, 0));
    for (int p = 0; p < NPL; ++p) CK(hipMemMap((char*)alias + (size_t)p * PLANE_B, PLANE_B, 0, pc[p], 0));
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(alias, NPL * PLANE_B, &acc, 1));
    CK(hipMemset(alias, 0xff, NPL * PLANE_B));
    for (int j = 0; j < nslots; ++j) {
        char* va = (char*)base + (size_t)j * (step_mib << 20);
        for (int p = 0; p < NPL; ++p) CK(hipMemMap(va + (size_t)p * PLANE_B, PLANE_B, 0, pc[p], 0));
        CK(hipMemSetAccess(va, NPL * PLANE_B, &acc, 1));
        Tab t; for (int p = 0; p < NPL; ++p) t.p[p] = (float*)(va + (size_t)p * PLANE_B);
        const double gbs = run(t);
        // did the stores land?  the kernel writes (float)(y + p) at row y of plane p: check a few elements of every
        // plane through this mapping and through a second, permanent mapping of the same pieces
        CK(hipDeviceSynchronize());
        int bad_here = 0, bad_alias = 0;
        for (int p = 0; p < NPL; ++p)
            for (int y : {0, 1777, 4095}) {
                float v[2] = {-1.f, -1.f};
                CK(hipMemcpy(&v[0], (float*)(va + (size_t)p * PLANE_B) + (size_t)y * N + 123, 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(&v[1], (float*)((char*)alias + (size_t)p * PLANE_B) + (size_t)y * N + 123, 4, hipMemcpyDeviceToHost));
                bad_here += v[0] != (float)(y + p);
                bad_alias += v[1] != (float)(y + p);
            }
        // scrub through the permanent mapping so that the next j cannot pass on old data
        CK(hipMemset(alias, 0xff, NPL * PLANE_B));
        printf("%s%3d:%5.0f%s", j % 8 ? " " : "\n", j, gbs, bad_here || bad_alias ? (bad_alias ? "(LOST)" : "(stale?)") : "");
        fflush(stdout);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(va, NPL * PLANE_B));
    }
    printf("\n");
    return 0;
}
