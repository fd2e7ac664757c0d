// frag_probe.hip -- does the speed of a many-plane streaming write depend on how the buffer's virtual and physical
// addresses are aligned (page-table fragment size -> TLB reach)?  Writes NPL planes of 4096x4096 f32 in the basis
// kernel's access shape (wave = 64-column strip, 19 rows, nontemporal dword stores, one cached input plane) into
// buffers obtained in different ways, several instances each, and prints GB/s per instance.
// Build: hipcc --offload-arch=gfx950 -O3 tools/frag_probe.hip -o tools/frag_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int N = 4096;

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
static double run(const float* in, float* out, size_t plane_stride, int reps = 30)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int sr = 19;
    dim3 grid(N / 256, (N + sr - 1) / sr);
    for (int i = 0; i < 4; ++i) k_planes<NPL><<<grid, 256>>>(in, out, plane_stride, sr);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) k_planes<NPL><<<grid, 256>>>(in, out, plane_stride, sr);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return (double)N * N * 4.0 * (NPL + 1) / (ms / reps) / 1e6;
}

struct Vmm { void* va; size_t size; hipMemGenericAllocationHandle_t h; void* base; size_t reserved; };

static bool vmm_alloc(size_t bytes, size_t va_align, size_t va_skew, Vmm& v)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return false;
    const size_t size = (bytes + gran - 1) / gran * gran;
    v.reserved = size + va_skew;
    if (hipMemAddressReserve(&v.base, v.reserved, va_align, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    v.va = (char*)v.base + va_skew;
    if (hipMemCreate(&v.h, size, &prop, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipMemMap(v.va, size, 0, v.h, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(v.va, size, &acc, 1) != hipSuccess) { (void)hipGetLastError(); return false; }
    v.size = size;
    return true;
}

int main(int argc, char** argv)
{
    const int inst = argc > 1 ? atoi(argv[1]) : 5;
    const size_t plane = (size_t)N * N;
    float* in; CK(hipMalloc(&in, plane * 4));
    CK(hipMemset(in, 0, plane * 4));
    size_t gran = 0;
    {
        hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        size_t gmin = 0; CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
        printf("VMM granularity: recommended %zu, minimum %zu\n", gran, gmin);
    }
    const size_t b12 = plane * 4 * 12, b20 = plane * 4 * 20;
    // A: plain hipMalloc, several instances alive at once
    {
        std::vector<float*> bufs(inst);
        for (auto& p : bufs) CK(hipMalloc(&p, b20));
        for (int i = 0; i < inst; ++i)
            printf("hipMalloc            #%d va=%p (va %% 1GiB = %4zu MiB)  9pl %7.1f  12pl %7.1f  20pl %7.1f GB/s\n", i, (void*)bufs[i],
                   (size_t)(((size_t)bufs[i] & ((1ull << 30) - 1)) >> 20), run<9>(in, bufs[i], plane), run<12>(in, bufs[i], plane), run<20>(in, bufs[i], plane));
        for (auto p : bufs) CK(hipFree(p));
    }
    // B: VMM, one physical allocation, VA aligned to 2 MiB / 64 MiB / 1 GiB / 2 GiB, and 1 GiB + a 2 MiB skew
    const size_t aligns[] = {2ull << 20, 64ull << 20, 1ull << 30, 2ull << 30};
    for (size_t al : aligns)
        for (int i = 0; i < (inst + 1) / 2; ++i) {
            Vmm v{};
            if (!vmm_alloc(b20, al, 0, v)) { printf("VMM align %zu MiB: not available\n", al >> 20); break; }
            printf("VMM va-align %4zu MiB #%d va=%p  9pl %7.1f  12pl %7.1f  20pl %7.1f GB/s\n", al >> 20, i, v.va,
                   run<9>(in, (float*)v.va, plane), run<12>(in, (float*)v.va, plane), run<20>(in, (float*)v.va, plane));
            // deliberately leaked until exit: instances must not reuse each other's physical memory
        }
    for (int i = 0; i < 2; ++i) {
        Vmm v{};
        if (!vmm_alloc(b20, 1ull << 30, 2ull << 20, v)) { printf("VMM skewed: not available\n"); break; }
        printf("VMM va-align 1 GiB + 2 MiB skew #%d va=%p  9pl %7.1f  12pl %7.1f  20pl %7.1f GB/s\n", i, v.va,
               run<9>(in, (float*)v.va, plane), run<12>(in, (float*)v.va, plane), run<20>(in, (float*)v.va, plane));
    }
    // C: VMM, 1 GiB VA alignment, the physical memory in pieces of 2 MiB / 64 MiB / 256 MiB (each its own hipMemCreate)
    const size_t pieces[] = {2ull << 20, 64ull << 20, 256ull << 20};
    for (size_t pc : pieces) {
        hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice;
        const size_t size = (b20 + pc - 1) / pc * pc;
        void* base = nullptr;
        if (hipMemAddressReserve(&base, size, 1ull << 30, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); break; }
        bool ok = true;
        for (size_t off = 0; off < size && ok; off += pc) {
            hipMemGenericAllocationHandle_t h;
            ok = hipMemCreate(&h, pc, &prop, 0) == hipSuccess && hipMemMap((char*)base + off, pc, 0, h, 0) == hipSuccess;
        }
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        ok = ok && hipMemSetAccess(base, size, &acc, 1) == hipSuccess;
        if (!ok) { printf("VMM pieces of %zu MiB: failed\n", pc >> 20); (void)hipGetLastError(); continue; }
        printf("VMM 1 GiB va-align, physical pieces of %3zu MiB  9pl %7.1f  12pl %7.1f  20pl %7.1f GB/s\n", pc >> 20,
               run<9>(in, (float*)base, plane), run<12>(in, (float*)base, plane), run<20>(in, (float*)base, plane));
    }
    // D: plane stride padded inside a hipMalloc block (bank / channel aliasing between planes?)
    {
        float* big; CK(hipMalloc(&big, b20 + (64ull << 20)));
        const size_t pads[] = {0, 64, 1024, 4096 + 64, 65536 + 1024, (2u << 20) / 4 + 1024};
        for (size_t pad : pads)
            printf("hipMalloc, plane stride 64 MiB + %7zu B: 9pl %7.1f  12pl %7.1f  20pl %7.1f GB/s\n", pad * 4,
                   run<9>(in, big, plane + pad), run<12>(in, big, plane + pad), run<20>(in, big, plane + pad));
        CK(hipFree(big));
    }
    return 0;
}
