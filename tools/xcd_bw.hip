// tools/xcd_bw.hip -- do the 8 XCDs stream to / from HBM equally fast?  Every workgroup moves the same number of
// bytes (its own contiguous 1 MiB chunk) and stamps its XCD (HW_REG_XCC_ID) and its duration (100 MHz counter).
// build: hipcc --offload-arch=gfx950 -O3 tools/xcd_bw.hip -o tools/xcd_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>  // 0 = write (nontemporal), 1 = read, 2 = copy
__global__ __launch_bounds__(256) void k(f4* dst, const f4* src, unsigned long long* stamp, int per_wg)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    f4* d = dst + (size_t)blockIdx.x * per_wg;
    const f4* s = src + (size_t)blockIdx.x * per_wg;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < per_wg; i += 256) {
        if (MODE == 0) __builtin_nontemporal_store(f4{1.f, 2.f, 3.f, (float)i}, d + i);
        if (MODE == 1) acc += s[i];
        if (MODE == 2) __builtin_nontemporal_store(s[i], d + i);
    }
    if (MODE == 1 && acc.x == 123.456f) d[0] = acc;
    __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0) {
        stamp[blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime() - t0;
        stamp[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;
    }
}
int main()
{
    const int per_wg = 65536;            // 1 MiB of float4 per workgroup
    const int wgs = 8192;                // 8 GiB in all: far beyond the Infinity Cache
    f4 *a, *b;
    unsigned long long* st;
    hipMalloc(&a, (size_t)wgs * per_wg * sizeof(f4));
    hipMalloc(&b, (size_t)wgs * per_wg * sizeof(f4));
    hipMalloc(&st, wgs * 2 * sizeof(unsigned long long));
    hipMemset(a, 0, (size_t)wgs * per_wg * sizeof(f4));
    hipMemset(b, 0, (size_t)wgs * per_wg * sizeof(f4));
    std::vector<unsigned long long> h(wgs * 2);
    const char* names[3] = {"write (nt)", "read", "copy (nt)"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, a, b, st, per_wg);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, a, b, st, per_wg);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, a, b, st, per_wg);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), st, wgs * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sum[8] = {0}; int cnt[8] = {0};
        for (int w = 0; w < wgs; ++w) { const int x = (int)h[w * 2 + 1]; sum[x] += h[w * 2] * 0.01; cnt[x]++; }
        printf("%-10s mean workgroup time per XCD (us):", names[mode]);
        for (int x = 0; x < 8; ++x) printf(" %6.1f", cnt[x] ? sum[x] / cnt[x] : 0.0);
        printf("   (workgroups per XCD: %d)\n", cnt[0]);
    }
    return 0;
}
