// frag_probe2.hip -- follow-up of frag_probe: the many-plane streaming write runs at very different speeds
// depending on WHERE in physical memory the planes lie relative to each other (frag_probe: one mapping made of
// 256 MiB physical pieces wrote 12 / 20 planes at 7.2 TB/s, plain hipMalloc at 5.5 / 6.4).  Here every plane is
// its own physical allocation (hipMemCreate, 64 MiB), all mapped back to back into one virtual range, and what
// varies is the physical distance between consecutive planes: a spacer allocation of S bytes is created between
// them (and released after the measurement).  Also: pieces of other sizes, repeated, to see what repeats.
// Build: hipcc --offload-arch=gfx950 -O3 tools/frag_probe2.hip -o tools/frag_probe2
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
static double run(const float* in, float* out, int reps = 20)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int sr = 19;
    dim3 grid(N / 256, (N + sr - 1) / sr);
    for (int i = 0; i < 3; ++i) k_planes<NPL><<<grid, 256>>>(in, out, PLANE_B / 4, sr);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) k_planes<NPL><<<grid, 256>>>(in, out, PLANE_B / 4, sr);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return (double)N * N * 4.0 * (NPL + 1) / (ms / reps) / 1e6;
}

static hipMemAllocationProp prop()
{
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    p.location.id = 0;
    return p;
}

struct Mapping {
    void* va = nullptr;
    size_t size = 0;
    std::vector<hipMemGenericAllocationHandle_t> phys, spacers;
};

// planes = NP physical allocations of `piece` bytes each (piece divides or is a multiple of the plane size), with a
// spacer of `spacer` bytes created after each one
static bool build(Mapping& m, int np, size_t piece, size_t spacer, bool reverse)
{
    const hipMemAllocationProp p = prop();
    const size_t total = (size_t)np * PLANE_B;
    const size_t npieces = (total + piece - 1) / piece;
    m.size = npieces * piece;
    if (hipMemAddressReserve(&m.va, m.size, 2ull << 20, nullptr, 0) != hipSuccess) return false;
    m.phys.resize(npieces);
    for (size_t i = 0; i < npieces; ++i) {
        if (hipMemCreate(&m.phys[i], piece, &p, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (spacer) {
            hipMemGenericAllocationHandle_t s;
            if (hipMemCreate(&s, spacer, &p, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
            m.spacers.push_back(s);
        }
    }
    for (size_t i = 0; i < npieces; ++i) {
        const size_t slot = reverse ? npieces - 1 - i : i;
        if (hipMemMap((char*)m.va + slot * piece, piece, 0, m.phys[i], 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    }
    hipMemAccessDesc acc = {};
    acc.location = p.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    return hipMemSetAccess(m.va, m.size, &acc, 1) == hipSuccess;
}

static void destroy(Mapping& m)
{
    CK(hipDeviceSynchronize());
    (void)hipMemUnmap(m.va, m.size);
    for (auto h : m.phys) (void)hipMemRelease(h);
    for (auto h : m.spacers) (void)hipMemRelease(h);
    (void)hipMemAddressFree(m.va, m.size);
    m = Mapping();
}

static void report(const char* what, const float* in, float* out)
{
    printf("%-58s 9pl %7.1f  12pl %7.1f  20pl %7.1f GB/s\n", what, run<9>(in, out), run<12>(in, out), run<20>(in, out));
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    float* in; CK(hipMalloc(&in, PLANE_B));
    CK(hipMemset(in, 0, PLANE_B));
    char nm[128];
    if (mode == 0 || mode == 1) {
        // piece size sweep, each twice (fresh physical memory both times: the first instance is kept alive)
        const size_t pieces_mib[] = {2, 8, 32, 64, 128, 256, 512, 1024, 1280};
        for (size_t pm : pieces_mib) {
            Mapping a, b;
            const bool oa = build(a, 20, pm << 20, 0, false);
            if (oa) { snprintf(nm, sizeof nm, "pieces of %4zu MiB, first instance", pm); report(nm, in, (float*)a.va); }
            const bool ob = build(b, 20, pm << 20, 0, false);
            if (ob) { snprintf(nm, sizeof nm, "pieces of %4zu MiB, second instance", pm); report(nm, in, (float*)b.va); }
            if (!oa || !ob) printf("pieces of %zu MiB: failed\n", pm);
            destroy(a); destroy(b);
        }
    }
    if (mode == 0 || mode == 2) {
        // one plane per physical allocation, physical distance between consecutive planes = 64 MiB + spacer
        const size_t spacers_mib[] = {0, 64, 192, 448, 960, 1984, 4032, 8128};
        for (size_t sm : spacers_mib) {
            Mapping a;
            if (build(a, 20, PLANE_B, sm << 20, false)) {
                snprintf(nm, sizeof nm, "plane = own allocation, %5zu MiB apart", sm + 64);
                report(nm, in, (float*)a.va);
            } else printf("spacer %zu MiB: failed\n", sm);
            destroy(a);
        }
        // and mapped in reverse order (does the ORDER of the planes in physical memory matter?)
        Mapping r;
        if (build(r, 20, PLANE_B, 0, true)) report("plane = own allocation, 64 MiB apart, reversed", in, (float*)r.va);
        destroy(r);
    }
    if (mode == 0 || mode == 3) {
        // 256 MiB pieces with spacers between the pieces
        const size_t spacers_mib[] = {0, 256, 768, 1792, 3840};
        for (size_t sm : spacers_mib) {
            Mapping a;
            if (build(a, 20, 256ull << 20, sm << 20, false)) {
                snprintf(nm, sizeof nm, "pieces of 256 MiB, %5zu MiB apart", sm + 256);
                report(nm, in, (float*)a.va);
            }
            destroy(a);
        }
    }
    // reference: plain hipMalloc, three instances alive together
    {
        float* b[3];
        for (auto& p : b) CK(hipMalloc(&p, 20 * PLANE_B));
        for (int i = 0; i < 3; ++i) { snprintf(nm, sizeof nm, "hipMalloc instance %d", i); report(nm, in, b[i]); }
        for (auto p : b) CK(hipFree(p));
    }
    return 0;
}
