// frag_probe3.hip -- speed of the 9-plane streaming write (basis-kernel access shape) as a function of WHERE the nine
// planes lie inside one large buffer: windows of 9 consecutive 64 MiB slots, slid through (a) a 16 GiB virtual range
// made of 64 MiB physical allocations (hipMemCreate), (b) a 16 GiB hipMalloc block, (c) the same slots but with the
// nine planes spread out (stride of several slots).
// Build: hipcc --offload-arch=gfx950 -O3 tools/frag_probe3.hip -o tools/frag_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int N = 4096;
constexpr size_t PLANE_B = (size_t)N * N * 4;

template <int NPL>
__global__ __launch_bounds__(256) void k_planes(const float* in, float* out, size_t plane_stride, int strip_rows)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = (blockIdx.x * 4 + wv) * 64 + lane;
    const int y0 = blockIdx.y * strip_rows;
    for (int y = y0; y < y0 + strip_rows && y < N; ++y) {
        const float v = in[(size_t)y * N + x];
#pragma unroll
        for (int p = 0; p < NPL; ++p) __builtin_nontemporal_store(v + p, out + p * plane_stride + (size_t)y * N + x);
    }
}

template <int NPL>
static double run(const float* in, float* out, size_t plane_stride_elems, int reps = 12)
{
    static hipEvent_t a = nullptr, b = nullptr;
    if (!a) { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    const int sr = 19;
    dim3 grid(N / 256, (N + sr - 1) / sr);
    for (int i = 0; i < 2; ++i) k_planes<NPL><<<grid, 256>>>(in, out, plane_stride_elems, sr);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) k_planes<NPL><<<grid, 256>>>(in, out, plane_stride_elems, sr);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return (double)N * N * 4.0 * (NPL + 1) / (ms / reps) / 1e6;
}

int main(int argc, char** argv)
{
    const int slots = argc > 1 ? atoi(argv[1]) : 256;
    float* in; CK(hipMalloc(&in, PLANE_B));
    CK(hipMemset(in, 0, PLANE_B));
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    p.location.id = 0;
    void* va = nullptr;
    const size_t total = (size_t)slots * PLANE_B;
    CK(hipMemAddressReserve(&va, total, 2ull << 20, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> hs(slots);
    for (int i = 0; i < slots; ++i) {
        CK(hipMemCreate(&hs[i], PLANE_B, &p, 0));
        CK(hipMemMap((char*)va + (size_t)i * PLANE_B, PLANE_B, 0, hs[i], 0));
    }
    hipMemAccessDesc acc = {};
    acc.location = p.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &acc, 1));
    float* v = (float*)va;
    const size_t PE = PLANE_B / 4;
    printf("va base %p\n", va);
    for (int pass = 0; pass < 1; ++pass) {
        printf("\nVMM, 64 MiB pieces, window of 9 consecutive slots starting at slot k (GB/s), pass %d:\n", pass);
        for (int k = 0; k + 9 <= slots; k += 3) { printf("%s%3d:%5.0f", (k / 3) % 8 ? "  " : "\n", k, run<9>(in, v + (size_t)k * PE, PE)); fflush(stdout); }
    }
    // Is the fast mode the Infinity Cache keeping part of the nine planes from one launch to the next (the timing
    // loop rewrites the same 576 MiB again and again)?  Alternate every launch with a write of nine OTHER planes
    // (a fixed "spoiler" window at the end of the range): whatever a launch leaves in the cache is evicted before
    // the same planes are written again.  pair = bytes of both launches / time of both.
    {
        const int spoil = slots - 10;
        printf("\n\nwindow k alternating with a spoiler window at slot %d: solo GB/s -> pair GB/s\n", spoil);
        const int sr = 19;
        dim3 grid(N / 256, (N + sr - 1) / sr);
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int k = 0; k + 9 <= spoil - 9; k += 3) {
            const double solo = run<9>(in, v + (size_t)k * PE, PE);
            for (int i = 0; i < 2; ++i) { k_planes<9><<<grid, 256>>>(in, v + (size_t)k * PE, PE, sr); k_planes<9><<<grid, 256>>>(in, v + (size_t)spoil * PE, PE, sr); }
            CK(hipEventRecord(a));
            for (int i = 0; i < 8; ++i) { k_planes<9><<<grid, 256>>>(in, v + (size_t)k * PE, PE, sr); k_planes<9><<<grid, 256>>>(in, v + (size_t)spoil * PE, PE, sr); }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("%s%3d:%5.0f->%5.0f", (k / 3) % 6 ? "  " : "\n", k, solo, 16.0 * N * N * 4.0 * 10 / ms / 1e6);
            fflush(stdout);
        }
    }
    printf("\n\nVMM, 9 planes with a stride of S slots, starting at slot 0 / 64 / 128:\n");
    for (int S : {1, 2, 3, 4, 5, 7, 8, 13, 16, 21})
        if (8 * S < slots - 128) printf("S=%2d: %5.0f %5.0f %5.0f\n", S, run<9>(in, v, PE * S), run<9>(in, v + 64 * PE, PE * S), run<9>(in, v + 128 * PE, PE * S));
    printf("\nVMM, single-plane write (1 plane, 64 MiB) per slot, every 8th slot:\n");
    for (int k = 0; k < slots; k += 8) printf("%s%3d:%5.0f", (k / 8) % 8 ? "  " : "\n", k, run<1>(in, v + (size_t)k * PE, PE));
    printf("\n");
    float* big = nullptr;
    if (hipMalloc(&big, total) == hipSuccess) {
        printf("\nhipMalloc block of the same size, window of 9 consecutive slots starting at slot k (GB/s):\n");
        for (int k = 0; k + 9 <= slots; k += 3) { printf("%s%3d:%5.0f", (k / 3) % 8 ? "  " : "\n", k, run<9>(in, big + (size_t)k * PE, PE)); fflush(stdout); }
        printf("\n\nhipMalloc, 9 planes with a stride of S slots, starting at slot 0 / 64 / 128:\n");
        for (int S : {1, 2, 3, 4, 5, 7, 8, 13, 16, 21})
            if (8 * S < slots - 128) printf("S=%2d: %5.0f %5.0f %5.0f\n", S, run<9>(in, big, PE * S), run<9>(in, big + 64 * PE, PE * S), run<9>(in, big + 128 * PE, PE * S));
    } else {
        printf("hipMalloc of %zu bytes failed\n", total);
    }
    return 0;
}
