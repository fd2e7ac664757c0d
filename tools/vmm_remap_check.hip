// vmm_remap_check.hip -- do stores land after hipMemUnmap + hipMemMap?  12 pieces of 64 MiB; a 12-plane block is
// mapped, written by a kernel, read back with hipMemcpy and verified, unmapped; then mapped again
//   A: at the SAME virtual address,   B: 64 MiB further (overlapping the previous range),   C: in a range never used,
// each with and without a permanent second mapping (alias) of the same pieces existing.
// Build: hipcc --offload-arch=gfx950 -O3 tools/vmm_remap_check.hip -o tools/vmm_remap_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <unistd.h>
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
int main(int argc, char** argv)
{
    const int wait_mode = argc > 1 ? atoi(argv[1]) : 0;
    printf("wait mode %d (0 none, 1 device sync after SetAccess, 2 + 20 ms sleep, 3 = 2 + unmap piece by piece)\n", wait_mode);
    hipMemAllocationProp pr = {};
    pr.type = hipMemAllocationTypePinned;
    pr.location.type = hipMemLocationTypeDevice;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int with_alias = 0; with_alias < 2; ++with_alias) {
        std::vector<hipMemGenericAllocationHandle_t> pc(NPL);
        for (auto& h : pc) CK(hipMemCreate(&h, PLANE_B, &pr, 0));
        void* alias = nullptr;
        if (with_alias) {
            CK(hipMemAddressReserve(&alias, NPL * PLANE_B, 2ull << 20, nullptr, 0));
            for (int p = 0; p < NPL; ++p) CK(hipMemMap((char*)alias + (size_t)p * PLANE_B, PLANE_B, 0, pc[p], 0));
            CK(hipMemSetAccess(alias, NPL * PLANE_B, &acc, 1));
        }
        void* base; CK(hipMemAddressReserve(&base, 64 * PLANE_B, 2ull << 20, nullptr, 0));
        const char* names[3] = {"same address", "64 MiB further (overlapping)", "fresh range"};
        for (int mode = 0; mode < 3; ++mode) {
            int bad_total = 0;
            for (int it = 0; it < 6; ++it) {
                const size_t slot = mode == 0 ? 0 : mode == 1 ? (size_t)(1 + it) : (size_t)(16 + 14 * it > 50 ? 50 : 16 + 14 * it);
                char* va = (char*)base + slot * PLANE_B;
                for (int p = 0; p < NPL; ++p) CK(hipMemMap(va + (size_t)p * PLANE_B, PLANE_B, 0, pc[p], 0));
                CK(hipMemSetAccess(va, NPL * PLANE_B, &acc, 1));
                if (wait_mode >= 1) CK(hipDeviceSynchronize());
                if (wait_mode >= 2) usleep(20000);
                Tab t; for (int p = 0; p < NPL; ++p) t.p[p] = (float*)(va + (size_t)p * PLANE_B);
                const float tag = 1000.f * (mode * 10 + it + 1);
                k_planes<<<dim3(N / 256, (N + 18) / 19), 256>>>(t, 19, tag);
                CK(hipDeviceSynchronize());
                int bad = 0;
                for (int p = 0; p < NPL; ++p)
                    for (int y : {0, 1777, 4095}) {
                        float v = -1.f;
                        CK(hipMemcpy(&v, (float*)(va + (size_t)p * PLANE_B) + (size_t)y * N + 123, 4, hipMemcpyDeviceToHost));
                        bad += v != tag + (float)(y + p);
                    }
                bad_total += bad;
                if (wait_mode == 3) { for (int p = 0; p < NPL; ++p) CK(hipMemUnmap(va + (size_t)p * PLANE_B, PLANE_B)); }
                else CK(hipMemUnmap(va, NPL * PLANE_B));
            }
            printf("%s, remapped at the %-30s: %d of 216 checked values wrong\n", with_alias ? "with a second mapping   " : "without second mapping  ", names[mode], bad_total);
        }
        for (auto h : pc) CK(hipMemRelease(h));
    }
    return 0;
}
