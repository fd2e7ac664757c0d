// d2h_probe.hip -- which virtual-memory operation makes the device-to-host copy rate of a process drop from 56 to ~30 GB/s
// (tools/pcie_probe.hip), and what brings it back?  128 MiB down from a buffer allocated at program start, pinned host
// memory, after each step of: create 48 pieces of 64 MiB / map them / write them / unmap half / unmap all / release /
// free the range / hipMalloc+hipFree of 1..8 GiB.
// Build: hipcc --offload-arch=gfx950 -O3 tools/d2h_probe.hip -o tools/d2h_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
using clk = std::chrono::steady_clock;
static char *d_out, *h_out, *d_in, *h_in;
static hipStream_t sd;
static void rate(const char* what)
{
    const size_t P = 128ull << 20;
    double best = 1e9, bestu = 1e9;
    for (int r = 0; r < 4; ++r) {
        auto t0 = clk::now();
        CK(hipMemcpyAsync(h_out, d_out, P, hipMemcpyDeviceToHost, sd)); CK(hipStreamSynchronize(sd));
        best = std::min(best, std::chrono::duration<double, std::milli>(clk::now() - t0).count());
        t0 = clk::now();
        CK(hipMemcpyAsync(d_in, h_in, P / 2, hipMemcpyHostToDevice, sd)); CK(hipStreamSynchronize(sd));
        bestu = std::min(bestu, std::chrono::duration<double, std::milli>(clk::now() - t0).count());
    }
    printf("%-64s down %5.1f GB/s   up %5.1f GB/s\n", what, P / best / 1e6, P / 2 / bestu / 1e6);
    fflush(stdout);
}
int main()
{
    const size_t P = 128ull << 20;
    CK(hipMalloc(&d_out, P)); CK(hipMalloc(&d_in, P)); CK(hipHostMalloc(&h_out, P)); CK(hipHostMalloc(&h_in, P));
    CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
    rate("start");
    hipMemAllocationProp pr = {}; pr.type = hipMemAllocationTypePinned; pr.location.type = hipMemLocationTypeDevice;
    hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t pb = 64ull << 20; const int np = 48;
    std::vector<hipMemGenericAllocationHandle_t> hs(np);
    for (auto& h : hs) CK(hipMemCreate(&h, pb, &pr, 0));
    rate("48 pieces of 64 MiB created");
    void* va; CK(hipMemAddressReserve(&va, np * pb, 2ull << 20, nullptr, 0));
    rate("virtual range reserved");
    for (int i = 0; i < np; ++i) CK(hipMemMap((char*)va + i * pb, pb, 0, hs[i], 0));
    rate("pieces mapped");
    CK(hipMemSetAccess(va, np * pb, &acc, 1));
    rate("access set");
    CK(hipMemset(va, 1, np * pb)); CK(hipDeviceSynchronize());
    rate("pieces written (hipMemset)");
    for (int i = 0; i < np / 2; ++i) CK(hipMemUnmap((char*)va + i * pb, pb));
    rate("first half unmapped");
    for (int i = 0; i < np / 2; ++i) CK(hipMemRelease(hs[i]));
    rate("first half released");
    for (int i = np / 2; i < np; ++i) CK(hipMemUnmap((char*)va + i * pb, pb));
    rate("second half unmapped");
    for (int i = np / 2; i < np; ++i) CK(hipMemRelease(hs[i]));
    rate("second half released");
    CK(hipMemAddressFree(va, np * pb));
    rate("virtual range freed");
    for (size_t gib : {1ull, 3ull, 8ull}) {
        void* t; CK(hipMalloc(&t, gib << 30)); CK(hipMemset(t, 0, gib << 30)); CK(hipDeviceSynchronize());
        char nm[64]; snprintf(nm, sizeof nm, "hipMalloc of %zu GiB (written)", (size_t)gib); rate(nm);
        CK(hipFree(t));
        snprintf(nm, sizeof nm, "... and freed"); rate(nm);
    }
    return 0;
}
