#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i4 make_rsrc(const void* p, unsigned bytes) {
    unsigned long long a = (unsigned long long)p;
    i4 r; r.x = (int)(unsigned)a; r.y = (int)((a >> 32) & 0xffff); r.z = (int)bytes; r.w = 0x00020000; return r;
}
__device__ __forceinline__ void dma(i4 rsrc, unsigned voff, unsigned soff, unsigned ldsaddr) {
    asm volatile("s_mov_b32 m0, %3\n\tbuffer_load_dword %0, %1, %2 offen lds" :: "v"(voff), "s"(rsrc), "s"(soff), "s"(ldsaddr) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__global__ void k(const float* in, float* out, int n)
{
    __shared__ float ring[4][128];
    const int lane = threadIdx.x;
    const i4 r = make_rsrc(in, n * 4);
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&ring[0][0]);
    // 4 rows prefetched, lane offsets: main = lane*4, halo = 12 lanes at 256+lane*4, others out of range
    const unsigned vm = lane * 4, vh = lane < 12 ? 256 + lane * 4 : 0x80000000u;
    for (int j = 0; j < 4; ++j) { dma(r, vm, j * 512, base + j * 512); dma(r, vh, j * 512, base + j * 512 + 256); }
    float acc = 0;
    wait_vm<6>(); acc += ring[0][lane] + ring[0][lane + 12];
    wait_vm<4>(); acc += ring[1][lane] + ring[1][lane + 12];
    wait_vm<2>(); acc += ring[2][lane] + ring[2][lane + 12];
    wait_vm<0>(); acc += ring[3][lane] + ring[3][lane + 12];
    out[lane] = acc;
    out[64 + lane] = ring[0][64 + lane];   // what did out-of-range lanes write?
}
int main() {
    float *in, *out; hipMalloc(&in, 4096 * 4); hipMalloc(&out, 1024);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i; hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    hipMemset(out, 0xff, 1024);
    k<<<1, 64>>>(in, out, 4096);
    float o[128]; hipMemcpy(o, out, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) { float e = 0; for (int j = 0; j < 4; ++j) e += (j * 128 + l) + (j * 128 + l + 12); if (o[l] != e) ++bad; }
    printf("bad %d; halo area lanes 0..15:", bad); for (int l = 0; l < 16; ++l) printf(" %g", o[64 + l]); printf("\n");
    return bad != 0;
}
