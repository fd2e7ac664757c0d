// frag_probe4.hip -- which property of the nine output planes' placement decides between the slow (5.7 TB/s) and the
// fast (7.2 TB/s) mode of the streaming write?  Planes are given as a table of nine pointers, so that any mixture of
// allocations can be tested: physical pieces (hipMemCreate) mapped in different virtual orders / at different virtual
// strides, separate hipMallocs, with and without gaps.
// Build: hipcc --offload-arch=gfx950 -O3 tools/frag_probe4.hip -o tools/frag_probe4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int N = 4096, NPL = 9;
constexpr size_t PLANE_B = (size_t)N * N * 4;
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

static double run(const float* in, const Tab& t, int reps = 16)
{
    static hipEvent_t a = nullptr, b = nullptr;
    if (!a) { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    const int sr = 19;
    dim3 grid(N / 256, (N + sr - 1) / sr);
    for (int i = 0; i < 3; ++i) k_planes<<<grid, 256>>>(in, t, sr);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) k_planes<<<grid, 256>>>(in, t, sr);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return (double)N * N * 4.0 * (NPL + 1) / (ms / reps) / 1e6;
}

static hipMemAllocationProp prop()
{
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    return p;
}

static hipMemGenericAllocationHandle_t piece(size_t bytes = PLANE_B)
{
    hipMemGenericAllocationHandle_t h;
    const hipMemAllocationProp p = prop();
    CK(hipMemCreate(&h, bytes, &p, 0));
    return h;
}

static void* map_at(void* va, hipMemGenericAllocationHandle_t h, size_t bytes = PLANE_B)
{
    CK(hipMemMap(va, bytes, 0, h, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop().location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, bytes, &acc, 1));
    return va;
}

int main()
{
    float* in; CK(hipMalloc(&in, PLANE_B));
    CK(hipMemset(in, 0, PLANE_B));
    // burn the fragmented start of the heap so that what follows comes from one large free block
    void* burn; CK(hipMalloc(&burn, 6ull << 30));

    // ---- T1: nine consecutive physical pieces, different virtual arrangements of the SAME pieces ----
    {
        std::vector<hipMemGenericAllocationHandle_t> h(NPL);
        for (auto& x : h) x = piece();
        auto test = [&](const char* what, const std::vector<int>& order, size_t va_stride) {
            void* va; CK(hipMemAddressReserve(&va, va_stride * NPL, 2ull << 20, nullptr, 0));
            Tab t;
            for (int p = 0; p < NPL; ++p) t.p[p] = (float*)map_at((char*)va + (size_t)p * va_stride, h[order[p]]);
            printf("T1 %-64s %7.1f GB/s\n", what, run(in, t));
            CK(hipDeviceSynchronize());
            for (int p = 0; p < NPL; ++p) CK(hipMemUnmap((char*)va + (size_t)p * va_stride, PLANE_B));
            CK(hipMemAddressFree(va, va_stride * NPL));
        };
        std::vector<int> id(NPL), rev(NPL), perm = {0, 2, 4, 6, 8, 1, 3, 5, 7};
        std::iota(id.begin(), id.end(), 0);
        for (int i = 0; i < NPL; ++i) rev[i] = NPL - 1 - i;
        test("9 consecutive pieces, plane p = piece p", id, PLANE_B);
        test("same pieces, plane p = piece 8-p", rev, PLANE_B);
        test("same pieces, planes = pieces 0,2,4,6,8,1,3,5,7", perm, PLANE_B);
        test("same pieces, plane p = piece p, virtual stride 128 MiB", id, 2 * PLANE_B);
        test("same pieces, plane p = piece p, virtual stride 64 MiB + 2 MiB", id, PLANE_B + (2ull << 20));
        test("same pieces, plane p = piece p (again)", id, PLANE_B);
        for (auto x : h) CK(hipMemRelease(x));
    }
    // ---- T2: 18 consecutive pieces, planes on the even ones; then the odd ones released; then planes on the odd-numbered slots ----
    {
        std::vector<hipMemGenericAllocationHandle_t> h(2 * NPL);
        for (auto& x : h) x = piece();
        void* va; CK(hipMemAddressReserve(&va, PLANE_B * NPL, 2ull << 20, nullptr, 0));
        Tab t;
        for (int p = 0; p < NPL; ++p) t.p[p] = (float*)map_at((char*)va + (size_t)p * PLANE_B, h[2 * p]);
        printf("T2 %-64s %7.1f GB/s\n", "18 pieces, planes on the even ones (odd ones allocated, unmapped)", run(in, t));
        for (int p = 0; p < NPL; ++p) CK(hipMemRelease(h[2 * p + 1]));
        printf("T2 %-64s %7.1f GB/s\n", "... after the odd pieces were released", run(in, t));
        void* other; CK(hipMalloc(&other, 9 * PLANE_B));  // something else takes the gaps (or not)
        printf("T2 %-64s %7.1f GB/s\n", "... after another 576 MiB hipMalloc", run(in, t));
        CK(hipDeviceSynchronize());
        CK(hipFree(other));
        for (int p = 0; p < NPL; ++p) { CK(hipMemUnmap((char*)va + (size_t)p * PLANE_B, PLANE_B)); CK(hipMemRelease(h[2 * p])); }
        CK(hipMemAddressFree(va, PLANE_B * NPL));
    }
    // ---- T3: every plane its own hipMalloc; T4: with a 64 MiB hipMalloc between them ----
    {
        Tab t;
        for (int p = 0; p < NPL; ++p) CK(hipMalloc(&t.p[p], PLANE_B));
        printf("T3 %-64s %7.1f GB/s\n", "nine separate hipMallocs of 64 MiB", run(in, t));
        for (int p = 0; p < NPL; ++p) CK(hipFree(t.p[p]));
        std::vector<void*> gaps;
        for (int p = 0; p < NPL; ++p) { CK(hipMalloc(&t.p[p], PLANE_B)); void* g; CK(hipMalloc(&g, PLANE_B)); gaps.push_back(g); }
        printf("T4 %-64s %7.1f GB/s\n", "nine hipMallocs of 64 MiB with a 64 MiB hipMalloc between them", run(in, t));
        for (void* g : gaps) CK(hipFree(g));
        printf("T4 %-64s %7.1f GB/s\n", "... after the gap allocations were freed", run(in, t));
        for (int p = 0; p < NPL; ++p) CK(hipFree(t.p[p]));
        // one hipMalloc block, planes at a stride of 128 MiB + k
        float* big; CK(hipMalloc(&big, 20 * PLANE_B));
        for (size_t stride_mib : {64, 128, 96, 80, 72, 66}) {
            for (int p = 0; p < NPL; ++p) t.p[p] = big + p * (stride_mib << 20) / 4;
            char nm[96]; snprintf(nm, sizeof nm, "one hipMalloc block, plane stride %zu MiB", stride_mib);
            printf("T5 %-64s %7.1f GB/s\n", nm, run(in, t));
        }
        CK(hipFree(big));
    }
    // ---- T6: ONE physical allocation of 1.25 GiB (hipMemCreate), planes at strides of 64 / 128 MiB inside it ----
    {
        hipMemGenericAllocationHandle_t h = piece(20 * PLANE_B);
        void* va; CK(hipMemAddressReserve(&va, 20 * PLANE_B, 2ull << 20, nullptr, 0));
        map_at(va, h, 20 * PLANE_B);
        Tab t;
        for (size_t stride_mib : {64, 128, 96}) {
            for (int p = 0; p < NPL; ++p) t.p[p] = (float*)va + p * (stride_mib << 20) / 4;
            char nm[96]; snprintf(nm, sizeof nm, "one hipMemCreate block of 1.25 GiB, plane stride %zu MiB", stride_mib);
            printf("T6 %-64s %7.1f GB/s\n", nm, run(in, t));
        }
    }
    return 0;
}
