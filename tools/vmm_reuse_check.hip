// vmm_reuse_check.hip -- hipMemAddressFree halves the host-link copy rate of the process (tools/d2h_probe.hip), so a
// reserved virtual range should be kept and reused.  Is a range that held pieces before safe for NEW pieces?  12 slots
// of 64 MiB: map fresh pieces, write by kernel, verify by hipMemcpy, unmap, release -- 6 generations at the same
// addresses, also with a different slot order and with only part of the range remapped; copy rate at the end.
// Build: hipcc --offload-arch=gfx950 -O3 tools/vmm_reuse_check.hip -o tools/vmm_reuse_check
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int N = 4096, NPL = 12;
constexpr size_t PLANE_B = (size_t)N * N * 4;
struct Tab { float* p[NPL]; };
__global__ __launch_bounds__(256) void k_planes(Tab t, int strip_rows, float tag)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = (blockIdx.x * 4 + wv) * 64 + lane;
    const int y0 = blockIdx.y * strip_rows;
    for (int y = y0; y < y0 + strip_rows && y < N; ++y) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) __builtin_nontemporal_store(tag + (float)(y + p), t.p[p] + (size_t)y * N + x);
    }
}
int main(int argc, char**)
{
    hipMemAllocationProp pr = {}; pr.type = hipMemAllocationTypePinned; pr.location.type = hipMemLocationTypeDevice;
    hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.flags = hipMemAccessFlagsProtReadWrite;
    const bool fresh_range = argc > 1;   // any argument: every generation reserves a NEW range and no range is ever freed
    void* base; CK(hipMemAddressReserve(&base, 16 * PLANE_B, 2ull << 20, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int gen = 0; gen < 6; ++gen) {
        const int first = gen == 3 ? 2 : gen == 4 ? 4 : 0;          // generations 3, 4: shifted windows of the same range
        std::vector<hipMemGenericAllocationHandle_t> pc(NPL);
        for (auto& h : pc) CK(hipMemCreate(&h, PLANE_B, &pr, 0));
        if (fresh_range && gen) CK(hipMemAddressReserve(&base, 16 * PLANE_B, 2ull << 20, nullptr, 0));
        char* va = (char*)base + (size_t)first * PLANE_B;
        for (int p = 0; p < NPL; ++p) { CK(hipMemMap(va + (size_t)p * PLANE_B, PLANE_B, 0, pc[p], 0)); }
        CK(hipMemSetAccess(va, NPL * PLANE_B, &acc, 1));
        Tab t; for (int p = 0; p < NPL; ++p) t.p[p] = (float*)(va + (size_t)p * PLANE_B);
        const float tag = 1000.f * (gen + 1);
        k_planes<<<dim3(N / 256, (N + 18) / 19), 256>>>(t, 19, tag);   // first touch
        CK(hipEventRecord(a));
        for (int r = 0; r < 5; ++r) k_planes<<<dim3(N / 256, (N + 18) / 19), 256>>>(t, 19, tag);
        CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        int bad = 0;
        for (int p = 0; p < NPL; ++p)
            for (int y : {0, 9, 1777, 4095})
                for (int x : {0, 123, 4095}) {
                    float v = -1.f;
                    CK(hipMemcpy(&v, (float*)(va + (size_t)p * PLANE_B) + (size_t)y * N + x, 4, hipMemcpyDeviceToHost));
                    bad += v != tag + (float)(y + p);
                }
        printf("generation %d (range %p): fresh pieces at slots %d..%d: %d of 144 checked values wrong, %.0f GB/s\n", gen, base, first, first + NPL - 1, bad,
               NPL * PLANE_B / (ms / 5) / 1e6);
        for (int p = 0; p < NPL; ++p) { CK(hipMemUnmap(va + (size_t)p * PLANE_B, PLANE_B)); CK(hipMemRelease(pc[p])); }
    }
    // copy rate with the range still reserved
    char *d, *h; CK(hipMalloc(&d, 128ull << 20)); CK(hipHostMalloc(&h, 128ull << 20));
    for (int phase = 0; phase < 2; ++phase) {
        double best = 1e9;
        for (int r = 0; r < 4; ++r) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipMemcpy(h, d, 128ull << 20, hipMemcpyDeviceToHost));
            best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        printf("device-to-host copy rate %s: %.1f GB/s\n", phase ? "after hipMemAddressFree" : "with the range still reserved", (128ull << 20) / best / 1e6);
        if (!phase && !fresh_range) CK(hipMemAddressFree(base, 16 * PLANE_B));
        if (!phase && fresh_range) break;
    }
    return 0;
}
