// pcie_probe.hip -- what the host link gives for the host-plane path: 64 MiB up (image) and 128 MiB down (two result
// planes), pageable vs pinned host memory, one after the other vs up and down at the same time from two host threads,
// whole planes vs row bands.  Build: hipcc --offload-arch=gfx950 -O3 tools/pcie_probe.hip -o tools/pcie_probe -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

int main()
{
    const size_t P = 64ull << 20;
    float *d_in, *d_out;
    CK(hipMalloc(&d_in, P)); CK(hipMalloc(&d_out, 2 * P));
    hipStream_t su, sd;
    CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
    for (int phase = 0; phase < 3; ++phase) {
    if (phase == 1) {
        // a storm of virtual-memory operations, as the placement search of an 8192^2 state makes them: 36 physical pieces of
        // 256 MiB created, mapped into one range, unmapped and released
        hipMemAllocationProp pr = {}; pr.type = hipMemAllocationTypePinned; pr.location.type = hipMemLocationTypeDevice;
        hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.flags = hipMemAccessFlagsProtReadWrite;
        const size_t pb = 256ull << 20; const int np = 36;
        std::vector<hipMemGenericAllocationHandle_t> hs(np);
        void* va; CK(hipMemAddressReserve(&va, np * pb, 2ull << 20, nullptr, 0));
        for (int i = 0; i < np; ++i) { CK(hipMemCreate(&hs[i], pb, &pr, 0)); CK(hipMemMap((char*)va + i * pb, pb, 0, hs[i], 0)); }
        CK(hipMemSetAccess(va, np * pb, &acc, 1));
        CK(hipMemset(va, 1, np * pb)); CK(hipDeviceSynchronize());
        for (int i = 0; i < np; ++i) { CK(hipMemUnmap((char*)va + i * pb, pb)); CK(hipMemRelease(hs[i])); }
        CK(hipMemAddressFree(va, np * pb));
        printf("--- after 36 x 256 MiB of hipMemCreate / Map / Unmap / Release ---\n");
    }
    if (phase == 2) {
        void* t; CK(hipMalloc(&t, 3ull << 30)); CK(hipMemset(t, 0, 3ull << 30)); CK(hipDeviceSynchronize()); CK(hipFree(t));
        printf("--- after a 3 GiB hipMalloc / hipFree ---\n");
    }
    for (int pinned = 0; pinned < 2; ++pinned) {
        char *h_in, *h_out;
        if (pinned) { CK(hipHostMalloc(&h_in, P)); CK(hipHostMalloc(&h_out, 2 * P)); }
        else { h_in = (char*)malloc(P); h_out = (char*)malloc(2 * P); }
        memset(h_in, 1, P); memset(h_out, 2, 2 * P);
        const char* kind = pinned ? "pinned  " : "pageable";
        for (int bands : {1, 8}) {
            const size_t bu = P / bands, bd = 2 * P / bands;
            auto up = [&] { for (int b = 0; b < bands; ++b) CK(hipMemcpyAsync((char*)d_in + b * bu, h_in + b * bu, bu, hipMemcpyHostToDevice, su)); CK(hipStreamSynchronize(su)); };
            auto down = [&] { CK(hipSetDevice(0)); for (int b = 0; b < bands; ++b) CK(hipMemcpyAsync(h_out + b * bd, (char*)d_out + b * bd, bd, hipMemcpyDeviceToHost, sd)); CK(hipStreamSynchronize(sd)); };
            up(); down();
            double t_up = 1e9, t_down = 1e9, t_seq = 1e9, t_par = 1e9;
            for (int r = 0; r < 5; ++r) {
                auto t0 = clk::now(); up(); t_up = std::min(t_up, ms_since(t0));
                t0 = clk::now(); down(); t_down = std::min(t_down, ms_since(t0));
                t0 = clk::now(); up(); down(); t_seq = std::min(t_seq, ms_since(t0));
                t0 = clk::now(); { std::thread th(down); up(); th.join(); } t_par = std::min(t_par, ms_since(t0));
            }
            printf("%s %d band(s): up 64 MiB %.3f ms (%.1f GB/s)  down 128 MiB %.3f ms (%.1f GB/s)  one after the other %.3f ms  two threads at once %.3f ms\n",
                   kind, bands, t_up, P / t_up / 1e6, t_down, 2 * P / t_down / 1e6, t_seq, t_par);
        }
        // hipHostRegister of pageable memory (pin in place), then copy
        if (!pinned) {
            auto t0 = clk::now();
            CK(hipHostRegister(h_in, P, hipHostRegisterDefault));
            CK(hipHostRegister(h_out, 2 * P, hipHostRegisterDefault));
            const double t_reg = ms_since(t0);
            t0 = clk::now();
            CK(hipMemcpyAsync(d_in, h_in, P, hipMemcpyHostToDevice, su)); CK(hipMemcpyAsync(h_out, d_out, 2 * P, hipMemcpyDeviceToHost, sd));
            CK(hipStreamSynchronize(su)); CK(hipStreamSynchronize(sd));
            const double t_cp = ms_since(t0);
            t0 = clk::now();
            CK(hipHostUnregister(h_in)); CK(hipHostUnregister(h_out));
            printf("pageable, hipHostRegister in place: register %.3f ms, up+down concurrently %.3f ms, unregister %.3f ms\n", t_reg, t_cp, ms_since(t0));
        }
        // CPU memcpy rate into a pinned buffer (what a library-side staging ring would cost)
        if (pinned) {
            char* src = (char*)malloc(P); memset(src, 3, P);
            double best = 1e9;
            for (int r = 0; r < 3; ++r) { auto t0 = clk::now(); memcpy(h_in, src, P); best = std::min(best, ms_since(t0)); }
            printf("CPU memcpy of 64 MiB into pinned memory, one thread: %.3f ms (%.1f GB/s)\n", best, P / best / 1e6);
            free(src);
        }
    }
    }
    return 0;
}
