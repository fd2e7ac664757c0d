// frag_probe6.hip -- frag_probe3 showed: inside a range made of consecutively created 64 MiB physical pieces the
// 9-plane streaming write is slow (5.7 TB/s) except for windows that straddle certain piece boundaries (7.2 TB/s),
// at the same places on every pass.  Here: find such a boundary by scanning, then (a) how many of the nine planes
// must lie beyond it, (b) how the speed depends on the number of planes written (2..9) inside one run and across the
// boundary, (c) the same for read-only and for plain (non-nt) stores.
// Build: hipcc --offload-arch=gfx950 -O3 tools/frag_probe6.hip -o tools/frag_probe6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int N = 4096, MAXP = 12;
constexpr size_t PLANE_B = (size_t)N * N * 4, PE = PLANE_B / 4;
struct Tab { float* p[MAXP]; };

template <int NPL, int MODE>  // MODE 0 = nt stores, 1 = plain stores, 2 = read-only
__global__ __launch_bounds__(256) void k_planes(const float* in, Tab t, int strip_rows, float* sink)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = (blockIdx.x * 4 + wv) * 64 + lane;
    const int y0 = blockIdx.y * strip_rows;
    float acc = 0.f;
    for (int y = y0; y < y0 + strip_rows && y < N; ++y) {
        if (MODE == 2) {
#pragma unroll
            for (int p = 0; p < NPL; ++p) acc += t.p[p][(size_t)y * N + x];
        } else {
            const float v = in[(size_t)y * N + x];
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                if (MODE == 0) __builtin_nontemporal_store(v + p, t.p[p] + (size_t)y * N + x);
                else t.p[p][(size_t)y * N + x] = v + p;
            }
        }
    }
    if (MODE == 2 && acc == 12345.678f) sink[0] = acc;
}

static hipEvent_t ea, eb;
template <int NPL, int MODE>
static double run(const float* in, const Tab& t, int reps = 12)
{
    const int sr = 19;
    dim3 grid(N / 256, (N + sr - 1) / sr);
    for (int i = 0; i < 2; ++i) k_planes<NPL, MODE><<<grid, 256>>>(in, t, sr, (float*)in);
    CK(hipEventRecord(ea));
    for (int i = 0; i < reps; ++i) k_planes<NPL, MODE><<<grid, 256>>>(in, t, sr, (float*)in);
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    CK(hipGetLastError());
    return (double)N * N * 4.0 * (NPL + (MODE == 2 ? 0 : 1)) / (ms / reps) / 1e6;
}

template <int MODE>
static double run_n(int n, const float* in, const Tab& t)
{
    switch (n) {
        case 1: return run<1, MODE>(in, t);
        case 2: return run<2, MODE>(in, t);
        case 3: return run<3, MODE>(in, t);
        case 4: return run<4, MODE>(in, t);
        case 5: return run<5, MODE>(in, t);
        case 6: return run<6, MODE>(in, t);
        case 7: return run<7, MODE>(in, t);
        case 8: return run<8, MODE>(in, t);
        case 9: return run<9, MODE>(in, t);
        default: return run<12, MODE>(in, t);
    }
}

int main(int argc, char** argv)
{
    const int slots = argc > 1 ? atoi(argv[1]) : 160;
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    float* in; CK(hipMalloc(&in, PLANE_B));
    CK(hipMemset(in, 0, PLANE_B));
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    void* va = nullptr;
    const size_t total = (size_t)slots * PLANE_B;
    CK(hipMemAddressReserve(&va, total, 2ull << 20, nullptr, 0));
    for (int i = 0; i < slots; ++i) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, PLANE_B, &p, 0));
        CK(hipMemMap((char*)va + (size_t)i * PLANE_B, PLANE_B, 0, h, 0));
    }
    hipMemAccessDesc acc = {};
    acc.location = p.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &acc, 1));
    float* v = (float*)va;
    auto window = [&](int k, int n = 9) { Tab t{}; for (int i = 0; i < MAXP; ++i) t.p[i] = v + (size_t)(k + (i < n ? i : 0)) * PE; return t; };
    // 1. scan every window start
    std::vector<double> sp(slots, 0.0);
    printf("9-plane nt write, window start k = 0..%d:\n", slots - 9);
    for (int k = 0; k + 9 <= slots; ++k) { sp[k] = run<9, 0>(in, window(k)); printf("%s%3d:%5.0f", k % 10 ? " " : "\n", k, sp[k]); }
    printf("\n");
    // 2. boundaries: last k of each fast stretch + 1 .. (a window is fast while it still contains the far side)
    std::vector<int> bounds;
    for (int k = 1; k + 9 <= slots; ++k)
        if (sp[k - 1] > 6700 && sp[k] < 6100) bounds.push_back(k);   // window k no longer straddles: boundary between k-1 and k
    printf("boundaries (first slot of the far side that windows stop containing): ");
    for (int b : bounds) printf("%d ", b);
    printf("\n");
    for (int b : bounds) {
        if (b < 12 || b + 12 > slots) continue;
        printf("\n== boundary between slot %d and %d ==\n", b - 1, b);
        // j planes at slots b-1, b-2, ... (near side, going down) ; 9-j planes at b, b+1, ...
        // note: a window [k, k+8] fast until k = b-1 means slots >= b are the side entered LAST when sliding up
        for (int j = 0; j <= 9; ++j) {
            Tab t{};
            int n = 0;
            for (int i = 0; i < j; ++i) t.p[n++] = v + (size_t)(b - 1 - i) * PE;
            for (int i = 0; n < 9; ++i) t.p[n++] = v + (size_t)(b + i) * PE;
            for (int i = 9; i < MAXP; ++i) t.p[i] = t.p[0];
            printf("  %d planes below the boundary (slots %d..), %d above: %6.0f GB/s\n", j, b - 1, 9 - j, run<9, 0>(in, t));
        }
        printf("  planes written n = 1..9 (+12), all inside the run above the boundary (slots %d..):      ", b);
        for (int n : {1, 2, 3, 4, 5, 6, 7, 8, 9, 12}) printf(" %d:%5.0f", n, run_n<0>(n, in, window(b, n)));
        printf("\n  planes written n = 2..9 (+12), alternating sides (slot b-1, b, b-2, b+1, ...):               ");
        for (int n : {2, 3, 4, 5, 6, 7, 8, 9, 12}) {
            Tab t{};
            for (int i = 0; i < MAXP; ++i) t.p[i] = v + (size_t)((i & 1) ? b + i / 2 : b - 1 - i / 2) * PE;
            printf(" %d:%5.0f", n, run_n<0>(n, in, t));
        }
        printf("\n  plain stores, 9 planes: inside %6.0f  straddling %6.0f ;  read-only 9 planes: inside %6.0f  straddling %6.0f\n",
               run<9, 1>(in, window(b)), run<9, 1>(in, window(b - 4)), run<9, 2>(in, window(b)), run<9, 2>(in, window(b - 4)));
        break;  // one boundary in detail is enough
    }
    return 0;
}
