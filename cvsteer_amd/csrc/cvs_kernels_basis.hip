// cvs_kernels_basis.hip -- "K1": the separable G2/H2 (7-plane) and G4/H4 (11-plane) basis
// filter bank, fused, for gfx950 (MI355X, wave64).
//
// Replaces the 7 (11) cv::sepFilter2D calls of SteerableFiltersG2::setup (reference
// cvsteer/SteerableFiltersG2.cpp:62-68) / SteerableFiltersG4::setup (SteerableFiltersG4.cpp:69-80)
// and, in the fused epilogues, G2.cpp:70-99 (C1..C3, theta, strength), G2.cpp:137-145 /
// G4.cpp:114-122 (scalar steer) and the callers' whole sequence test/test.cpp:86-90 (F_PIPE).
//
// Design (HBM-bound: 4 B read + 28/44 B written per pixel; ~46 / ~109 vector instructions per output row and wave, most of them packed):
//  * one WAVE owns a strip 64 columns wide and `strip_rows` tall and marches down it; the four
//    waves of a workgroup own four adjacent strips and never synchronise with each other.
//  * per input row the wave issues one 256-B row load (+ one 2W-lane halo load) that lands DIRECTLY in the wave's ring of
//    2W+1 LDS lines (buffer_load ... lds: no prefetch registers, no LDS stores), and every lane reads its 2W+1
//    neighbours back: the row pass.  REFLECT_101 is folded into the load addresses, so the
//    border costs nothing in the loop.
//  * the 6 (10) distinct row-filtered values enter a (2W+1)-deep sliding window held in
//    VGPRs (the row loop is unrolled 2W+1 times so every window slot is a fixed register);
//    the column pass runs on that window, also in folded symmetric / antisymmetric form.
//  * both passes work on PAIRS -- a mirror and an anti-mirror kernel in the two halves of a register pair, v_pk_fma_f32 with
//    tap pairs from SGPR pairs ("Packed f32" below): a tap multiply with a scalar operand costs the SIMD as much as a packed one.
//  * the input is read once and every output plane is written once with 256-B row segments,
//    nontemporal once the planes outgrow the Infinity Cache.
//  * loads for the next 2W+1 rows are in flight while the current ones are filtered (the line
//    just consumed is refilled at once; the prefetch is branch-free; waits are counted by hand, see dma_row).
//  * every memory access is a buffer instruction: plane = resource (SGPRs), row = scalar offset,
//    column = one per-lane byte offset shared by all loads and stores; masked lanes use an
//    out-of-range offset that the hardware range check drops.
//  * the state planes lie row-interleaved in groups ([row][plane][column]: basis | orientation, for G4 G | H | orientation;
//    BasisArgs pitch / plane_stride per group, cvs_handle.cpp layout_state): the launch's write frontier is one linear
//    sweep per group; 8-bit images are read as bytes (template U8) and widened in registers.
//  * variants (templates): epilogue flags, streaming stores, batched launch (grid.z = frame),
//    single state resource; G4 runs as two half banks side by side in one launch (k_basis_pair).
//  * which tile a workgroup takes: pick_tile -- static orders from blockIdx, or the tail of the launch from per-XCD queues.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "cvs_device_math.h"
#include "cvs_internal.h"
#include "cvs_lit_taps.h"

namespace cvs {

// ---------------------------------------------------------------------------------------
// filter-bank descriptors: which distinct 1-D kernels exist and which (row, column) pair
// makes each basis plane.  Kernel ids: [0, NE) even (mirror), [NE, NE+NO) odd (anti-mirror).
// ---------------------------------------------------------------------------------------
// Packed f32.  Measured on this GPU (tools/valu_rate.hip, profiles/r05_valu_rate.txt; cycles a SIMD is taken per wave64 instruction): a
// vector instruction with a scalar-register operand -- every tap multiply of the plain form -- 4.3-4.7, a packed one (v_pk_fma_f32,
// v_pk_add_f32, v_pk_mul_f32: two f32 results per lane) 4.3-4.5, and only tap-free simple instructions on VGPRs / literals 2.3-2.5 (two of
// them from different waves go through together).  A tap multiply therefore costs what a packed one costs, and the strip kernels issue
// 70-100 % of what the SIMDs take.
// So the two passes work on PAIRS: an even (mirror) and an odd (anti-mirror) 1-D kernel side by side in the two halves of a
// register pair.
//   row pass     {s[W+i], s[W-i]} -> {sum_i, dif_i} in one v_pk_add_f32 (neg_hi on the second operand); row pair k accumulates
//                {even kernel te(k), odd kernel to(k)} with tap pairs from SGPR pairs -- one v_pk_fma_f32 per tap index;
//   window       f2 per row pair and slot;
//   column pass  a PLANE pair (lo, hi) whose row kernels are a row pair takes its column taps from a tap table as well (halves
//                swapped by op_sel where the lo plane's column kernel is the odd one): W v_pk_add_f32 (+ / - per half by
//                neg_lo / neg_hi), W packed multiply-adds, and ONE plain fma for the centre tap of the even half (the odd column
//                kernel has no centre term in the reference's folded column filter; the row filter has, +0.0).
// Per result the operations and their order are those of the plain form (an fma is an fma): results are bit-identical.
// Planes and kernels left over (G2: g2c; G4 G bank: g4c and its row kernel G45) stay on plain instructions.
typedef float f2 __attribute__((ext_vector_type(2)));
struct PairOp { int lo, hi, w, t; bool swap; };   // basis planes (bank-local) in the lo / hi half, window pair, tap table; swap: lo plane's column kernel = the table's odd one
struct SingleOp { int plane, w, t; };             // w >= 0: lo half of window pair w, else single window -1-w; t >= 0: even kernel of table t, else single taps -1-t

struct BankG2 {  // SteerableFiltersG2.cpp:62-68
    static constexpr int KIND = 2, W = 4, NE = 3, NO = 3, NB = 7;
    static constexpr int MIN_WAVES = 1;  // waves per SIMD the register allocator must leave room for (1 = unconstrained)
    // even: E0=G21 E1=G22(=H22) E2=H24 ; odd: O0=G23 O1=H21 O2=H23 (ids 3,4,5)
    __host__ __device__ static constexpr int rx(int p) { constexpr int t[NB] = {0, 3, 1, 4, 2, 5, 1}; return t[p]; }
    __host__ __device__ static constexpr int cy(int p) { constexpr int t[NB] = {1, 3, 0, 1, 5, 2, 4}; return t[p]; }
    static constexpr int even_member(int r) { constexpr int t[NE] = {0, 1, 6}; return t[r]; }
    static constexpr int odd_member(int r) { constexpr int t[NO] = {2, 3, 5}; return t[r]; }
    static constexpr int dup_a = 1, dup_b = 4;  // m_g2 == m_h2 bit for bit
    static constexpr int PLANE0 = 0, HALF = 0;  // first basis plane written; 0 = whole bank
    // plane offsets in per-lane registers (VOFF, see BankG4G) and the strength-reduced scalar bookkeeping of the row loop (SRED,
    // see basis_body): off for the bank as a whole -- the pipeline variants are short of SGPRs and spill with it --, switched
    // on per variant in basis_body for the basis / orientation / fused-steer launches (FLAGS 0..3)
    static constexpr bool VOFF = false;
    static constexpr bool SRED = false;
    // packed arithmetic (see "Packed f32" below): tap tables (even, odd), the first NRP of them are the row pairs = window pairs
    static constexpr int NTP = 4, NRP = 3, NS = 0, NPP = 3, NSP = 1;
    static constexpr int te(int m) { constexpr int t[NTP] = {0, 1, 2, 1}; return t[m]; }
    static constexpr int to(int m) { constexpr int t[NTP] = {0, 1, 2, 0}; return t[m]; }
    static constexpr int se(int) { return 0; }
    static constexpr PairOp pp(int q) { constexpr PairOp t[NPP] = {{6, 3, 1, 1, true}, {4, 5, 2, 2, true}, {0, 1, 0, 3, false}}; return t[q]; }
    static constexpr SingleOp sp(int q) { constexpr SingleOp t[NSP] = {{2, 1, 0}}; return t[q]; }
};

// The 11-plane G4 bank as ONE kernel needs a 10 x 13 register window (158 VGPRs with the LDS-DMA input path, three waves per
// SIMD) and runs 3-6 % behind this form at every strip height and order (profiles/r05_input_dma_ab.txt; removed in round 5): its
// G and H halves share no row-filtered plane, so they run as the two z-planes of one launch with 65- and 78-register windows
// (126 VGPRs, four waves per SIMD); the image is read twice (the second read is an L2 / Infinity-Cache hit).
struct BankG4G {  // planes g4a..g4e, SteerableFiltersG4.cpp:69-73
    static constexpr int KIND = 4, W = 6, NE = 3, NO = 2, NB = 5;
    static constexpr int MIN_WAVES = 1;
    // even: E0=G41 E1=G42 E2=G45 ; odd: O0=G43 O1=G44 (ids 3,4)
    __host__ __device__ static constexpr int rx(int p) { constexpr int t[NB] = {0, 3, 2, 4, 1}; return t[p]; }
    __host__ __device__ static constexpr int cy(int p) { constexpr int t[NB] = {1, 4, 2, 3, 0}; return t[p]; }
    static constexpr int even_member(int r) { constexpr int t[NE] = {0, 1, 4}; return t[r]; }
    static constexpr int odd_member(int r) { constexpr int t[NO] = {2, 3}; return t[r]; }
    static constexpr int dup_a = 1, dup_b = 6;
    static constexpr int PLANE0 = 0, HALF = 1;
    // single-resource form: the plane's byte offset sits in one per-lane register per plane (fixed for the strip) instead of
    // being added to the scalar row offset before every store -- 5 / 6 scalar instructions per row less; the G4 half banks
    // have the registers to spare (131 -> 137 VGPRs, still three waves per SIMD) and are sensitive to the scalar unit
    static constexpr bool VOFF = true;
    static constexpr bool SRED = true;
    static constexpr int NTP = 2, NRP = 2, NS = 1, NPP = 2, NSP = 1;
    static constexpr int te(int m) { constexpr int t[NTP] = {0, 1}; return t[m]; }
    static constexpr int to(int m) { constexpr int t[NTP] = {0, 1}; return t[m]; }
    static constexpr int se(int) { return 2; }   // G45: row and column kernel of g4c, in no pair
    static constexpr PairOp pp(int q) { constexpr PairOp t[NPP] = {{0, 1, 0, 1, false}, {4, 3, 1, 0, false}}; return t[q]; }
    static constexpr SingleOp sp(int q) { constexpr SingleOp t[NSP] = {{2, -1, -1}}; return t[q]; }
};

struct BankG4H {  // planes h4a..h4f, SteerableFiltersG4.cpp:75-80
    static constexpr int KIND = 4, W = 6, NE = 3, NO = 3, NB = 6;
    static constexpr int MIN_WAVES = 1;
    // even: E0=H42 E1=H43 E2=H46 ; odd: O0=H41 O1=H44 O2=H45 (ids 3,4,5)
    __host__ __device__ static constexpr int rx(int p) { constexpr int t[NB] = {3, 1, 5, 2, 4, 0}; return t[p]; }
    __host__ __device__ static constexpr int cy(int p) { constexpr int t[NB] = {0, 4, 2, 5, 1, 3}; return t[p]; }
    static constexpr int even_member(int r) { constexpr int t[NE] = {6, 7, 10}; return t[r]; }
    static constexpr int odd_member(int r) { constexpr int t[NO] = {5, 8, 9}; return t[r]; }
    static constexpr int dup_a = 1, dup_b = 6;
    static constexpr int PLANE0 = 5, HALF = 2;
    static constexpr bool VOFF = true;
    static constexpr bool SRED = true;
    static constexpr int NTP = 3, NRP = 3, NS = 0, NPP = 3, NSP = 0;
    static constexpr int te(int m) { return m; }
    static constexpr int to(int m) { return m; }
    static constexpr int se(int) { return 0; }
    static constexpr PairOp pp(int q) { constexpr PairOp t[NPP] = {{5, 0, 0, 0, true}, {1, 4, 1, 1, true}, {3, 2, 2, 2, true}}; return t[q]; }
    static constexpr SingleOp sp(int) { return SingleOp{0, 0, 0}; }
};

// the tables above against rx / cy: every plane exactly once, row kernels = the window pair's, column kernels = the tap table's
template <class B>
constexpr bool bank_tables_ok()
{
    int seen[B::NB] = {};
    for (int q = 0; q < B::NPP; ++q) {
        const PairOp o = B::pp(q);
        if (o.w >= B::NRP || o.t >= B::NTP) return false;
        if (B::rx(o.lo) != B::te(o.w) || B::rx(o.hi) != B::NE + B::to(o.w)) return false;
        if (B::cy(o.lo) != (o.swap ? B::NE + B::to(o.t) : B::te(o.t)) || B::cy(o.hi) != (o.swap ? B::te(o.t) : B::NE + B::to(o.t))) return false;
        ++seen[o.lo]; ++seen[o.hi];
    }
    for (int q = 0; q < B::NSP; ++q) {
        const SingleOp o = B::sp(q);
        if (B::rx(o.plane) != (o.w >= 0 ? B::te(o.w) : B::se(-1 - o.w)) || B::cy(o.plane) != (o.t >= 0 ? B::te(o.t) : B::se(-1 - o.t))) return false;
        ++seen[o.plane];
    }
    for (int p = 0; p < B::NB; ++p) if (seen[p] != 1) return false;
    for (int k = 0; k < B::NRP; ++k) if (B::te(k) >= B::NE || B::to(k) >= B::NO) return false;
    return true;
}
static_assert(bank_tables_ok<BankG2>() && bank_tables_ok<BankG4G>() && bank_tables_ok<BankG4H>(), "pair tables do not match rx / cy");

// folded taps in pairs: tp[m][i] = {even kernel te(m) at offset +/-i, odd kernel to(m) at offset +i}, i = 0..W; the odd kernel's centre
// tap tp[m][0].y is +0.0 (what the row filter multiplies the centre sample with).  ts[q][i]: the even kernels no table holds.
template <class B>
struct Folded {
    f2 tp[B::NTP][B::W + 1];
    float ts[B::NS ? B::NS : 1][B::W + 1];
};

// Where a launch's taps come from.  TapsArg: the kernel arguments (scalar registers; any taps the handle was made with).  TapsLitG2: the
// reference's DEFAULT G2 / H2 taps as compile-time constants -- they end up as 32-bit literals in the instructions.  Why: a vector instruction
// with a scalar-register operand takes its SIMD for 4.3-4.7 cycles, the same instruction with a literal 2.3-2.5 and goes through beside
// another wave's (profiles/r05_valu_rate.txt), and the pipeline variants -- plain arithmetic, bound by instruction issue -- spend a quarter of
// their vector instructions on tap multiplies.  Same operations on the same values in the same order: bit-identical.  The launcher picks
// the literal instance only when the handle's folded taps equal the table bit for bit (launch_fast_impl).
template <class B>
struct TapsArg {
    const Folded<B>& t;
    __device__ __forceinline__ f2 tp(int m, int i) const { return t.tp[m][i]; }
    __device__ __forceinline__ float ts(int q, int i) const { return t.ts[q][i]; }
};
struct TapsLitG2 {
    __device__ __forceinline__ f2 tp(int m, int i) const
    {
        return f2{__builtin_bit_cast(float, kLitEvenG2[BankG2::te(m)][i]), __builtin_bit_cast(float, kLitOddG2[BankG2::to(m)][i])};
    }
    __device__ __forceinline__ float ts(int, int) const { return 0.f; }
};

// PK = false: the same results from plain instructions on the two halves (same operations, same order).
// {a.x + a.y, a.x - a.y}
template <bool PK>
__device__ __forceinline__ f2 pk_sumdif(f2 a)
{
    if constexpr (!PK) return f2{a.x + a.y, a.x - a.y};
    f2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}
// {a.x +/- b.x, a.y +/- b.y}: minus where the half's column kernel is odd
template <bool PK, bool ODD_LO, bool ODD_HI>
__device__ __forceinline__ f2 pk_addsub(f2 a, f2 b)
{
    if constexpr (!PK) return f2{ODD_LO ? a.x - b.x : a.x + b.x, ODD_HI ? a.y - b.y : a.y + b.y};
    f2 r;
    if constexpr (ODD_LO && ODD_HI) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    else if constexpr (ODD_LO) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    else if constexpr (ODD_HI) asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// t (a pair of SGPRs; halves exchanged when SWAP) * x
template <bool PK, bool SWAP>
__device__ __forceinline__ f2 pk_mul(f2 t, f2 x)
{
    if constexpr (!PK) return f2{(SWAP ? t.y : t.x) * x.x, (SWAP ? t.x : t.y) * x.y};
    f2 r;
    if constexpr (SWAP) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "s"(t), "v"(x));
    else asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "s"(t), "v"(x));
    return r;
}
// {word W + i, word W - i} of this lane's window into a ring line (addr = LDS byte address of word 0); asynchronous: the caller waits
// lgkmcnt(0) before it uses the value
template <int W, int I>
__device__ __forceinline__ f2 lds_pair(unsigned addr)
{
    f2 r;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(addr), "n"(W + I), "n"(W - I) : "memory");
    return r;
}
template <int W, int... I>
__device__ __forceinline__ void lds_pairs(unsigned addr, f2* P, std::integer_sequence<int, I...>)
{
    ((P[I] = lds_pair<W, I>(addr)), ...);
}
// The launch's BasisArgs as they lie in the kernel-argument segment (first argument of every strip kernel), through a pointer the
// compiler must treat as new at every call: what is read through it is loaded where it is used (s_load, scalar cache) and not kept
// in scalar registers across the row loop.
typedef const __attribute__((address_space(4))) BasisArgs* kernarg_ptr_t;
__device__ __forceinline__ kernarg_ptr_t kernarg_fresh()
{
    kernarg_ptr_t ka = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return ka;
}
template <bool PK, bool SWAP>
__device__ __forceinline__ f2 pk_fma(f2 t, f2 x, f2 acc)
{
    if constexpr (!PK) return f2{fmaf(SWAP ? t.y : t.x, x.x, acc.x), fmaf(SWAP ? t.x : t.y, x.y, acc.y)};
    if constexpr (SWAP) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(t), "v"(x));
    else asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "s"(t), "v"(x));
    return acc;
}

enum { F_ORIENT = 1, F_STEER = 2, F_PIPE = 4, F_NOSTATE = 8, F_PYR = 16, F_PYRONLY = 32, F_FEAT3 = 64 };  // F_PIPE implies F_ORIENT; F_NOSTATE: outputs only;
                                                                            // F_FEAT3 (with F_NOSTATE): exactly the three feature maps, find*(magnitude, phase), fastAtan2 --
                                                                            // what example/steer.cpp keeps -- decided at COMPILE time: no per-output tests and branches in the row
                                                                            // loop, no second arctangent path in the instruction cache (+3.5 % on 32 x 1080p; same values);
                                                                            // F_PYR: also emit cv::pyrDown(image) (next pyramid level);
                                                                            // F_PYRONLY (with F_PYR): nothing but that -- cvs_pyr_down as a strip march

// Addressing idiom: buffer instructions.  A plane is a raw buffer resource (4 SGPRs, built from
// wave-uniform values only), the row is the scalar offset (one SGPR, `soffset`), the lane's column
// is a per-lane BYTE offset that is fixed for the whole strip (one VGPR shared by every load and
// store of the kernel).  No per-access 64-bit VALU address arithmetic, no address VGPR pairs.
// Planes are addressed with 32-bit offsets: a plane may be up to 4 GiB (32768 x 32768 f32).
typedef __amdgpu_buffer_rsrc_t rsrc_t;

// A lane offset at or above every legal num_records (planes are < 2 GiB on this path): the
// hardware range check turns such a load into 0 and drops such a store, with no exec-mask branch
// and -- whether or not the scalar row offset takes part in the check -- no 32-bit wrap.
constexpr unsigned kLaneOff = 0x80000000u;
constexpr size_t kMaxPlaneBytes = 0x7ffffff0ull;

// BORDER_REFLECT_101 when at most one reflection is needed (-len < p < 2*len - 1): two scalar
// compare/selects.  The launcher sends images too small for this to the generic path.
__device__ __forceinline__ int reflect1(int p, int len)
{
    p = p < 0 ? -p : p;
    return p >= len ? 2 * len - 2 - p : p;
}

__device__ __forceinline__ rsrc_t plane_rsrc(const float* base, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0,
                                             (int)(bytes > kMaxPlaneBytes ? kMaxPlaneBytes : bytes), 0x00020000);
}
// cache policy of the streaming stores (aux bits of the buffer store on gfx94x / gfx950: 1 = sc0, 2 = nt, 16 = sc1).  nt alone is the
// product's (the other policies were measured in round 4: profiles/HISTORY_round_4.md); -DCVS_STREAM_AUX=n builds a twin with another one
#ifndef CVS_STREAM_AUX
#define CVS_STREAM_AUX 2
#endif
template <bool STREAM>
__device__ __forceinline__ void bst(rsrc_t r, unsigned lane_off, unsigned row_off, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, lane_off, row_off, STREAM ? CVS_STREAM_AUX : 0);
}

// ---------------------------------------------------------------------------------------
// Input rows straight into LDS (round 5): `buffer_load_dword ... lds` writes lane i's dword to LDS address M0 + 4 i without
// passing through a VGPR.  Every wave owns a ring of 2W+1 lines of kRingLine floats; line j holds input row (i mod 2W+1 = j):
// words [0, 64) = columns x0-W .. x0-W+63 (one load, all lanes), words [64, 64+2W) = the next 2W columns (a second load whose
// other lanes carry an out-of-range offset: the hardware returns 0 for them, which lands in the line's padding).  Lane l then
// reads its 2W+1 neighbours as line[l .. l+2W] -- the same values the staged line of rounds 1-4 held, without the 2 x (2W+1)
// prefetch VGPRs and the two ds_write per row.  The compiler knows nothing about these LDS writes (its own LDS-DMA intrinsic
// makes every later LDS read wait for the LAST such load -- no prefetching), so the loads are inline assembly and the waits are
// counted by hand: vector-memory operations retire in issue order and `s_waitcnt vmcnt(N)` waits until at most N are
// outstanding, so a row has landed once N = number of loads and stores issued after its halo load (see the row loop).
// ---------------------------------------------------------------------------------------
constexpr int kRingLine = 128;   // floats per ring line (512 B): 64 + 2W used, the rest takes the zeros of the halo load's idle lanes
typedef int i4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_float;

__device__ __forceinline__ i4_t raw_rsrc(const void* base, size_t bytes)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    i4_t r;
    r.x = (int)(unsigned)a;
    r.y = (int)((a >> 32) & 0xffffu);
    r.z = (int)(bytes > kMaxPlaneBytes ? kMaxPlaneBytes : bytes);
    r.w = 0x00020000;
    return r;
}
// (M0 is written here behind the compiler's back.  It cannot be declared: M0 is a reserved register for this target and clang ignores
// -- and warns about -- a clobber of it.  The compiler itself writes M0 only right in front of an instruction that reads it (LDS-DMA
// intrinsics, s_movrel, sendmsg, GWS), none of which these kernels contain; tests/test_isa_cpu.py disassembles the built object and
// fails if any instruction other than the three below touches M0.)
// both loads of one input row into the ring line at LDS byte address `lds_line` (s_nop: one wait state between an SALU write of
// M0 and the LDS-DMA that reads it, which the assembler does not insert inside inline assembly).  8-bit images: the byte load
// writes the zero-extended sample as a dword, and the lanes convert what they read back (cv::Mat1f(const Mat&), unscaled).
template <bool U8>
__device__ __forceinline__ void dma_row(i4_t rsrc, unsigned vmain, unsigned vhalo, unsigned soff, unsigned lds_line)
{
    if constexpr (U8)
        asm volatile("s_mov_b32 m0, %3\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_ubyte %0, %2, %4 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x100\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_ubyte %1, %2, %4 offen lds"
                     :
                     : "v"(vmain), "v"(vhalo), "s"(rsrc), "s"(lds_line), "s"(soff)
                     : "memory", "scc");
    else
        asm volatile("s_mov_b32 m0, %3\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dword %0, %2, %4 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x100\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dword %1, %2, %4 offen lds"
                     :
                     : "v"(vmain), "v"(vhalo), "s"(rsrc), "s"(lds_line), "s"(soff)
                     : "memory", "scc");
}
// one load whose only purpose is to bring a row segment into this XCD's L2 ahead of the workgroup that will need it: it lands in a
// spare LDS line nobody reads (no VGPR whose late write-back could hit a register the compiler has given to something else)
template <bool U8>
__device__ __forceinline__ void dma_warm(i4_t rsrc, unsigned voff, unsigned soff, unsigned lds_line)
{
    if constexpr (U8) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_ubyte %0, %1, %3 offen lds" : : "v"(voff), "s"(rsrc), "s"(lds_line), "s"(soff) : "memory");
    else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %3 offen lds" : : "v"(voff), "s"(rsrc), "s"(lds_line), "s"(soff) : "memory");
}
// s_waitcnt vmcnt(n) for an n that is a constant once the row loop is unrolled (the instruction takes an immediate; the counter
// has 6 bits, and a smaller n only waits longer)
#ifdef CVS_DIAG_CANARY
// Canary twin (make canary; tests/test_gpu_canary.py): a DIRECT check of the hand-counted waits.  Every ring line is filled with a
// pattern no image contains before the load that refills it is issued; a lane that still reads the pattern behind the
// s_waitcnt vmcnt(N) that is supposed to cover the row has caught a count that is too high (or an LDS-DMA load that retires from
// the counter before its data is in the LDS).  The twin also tallies the vector-memory stores of every output row against S_ROW,
// the compile-time lower bound the counts are built from.  g_canary: [0] stale words read, [1] output rows with fewer stores
// than S_ROW, [2] row reads checked, [3] output rows tallied.  -DCVS_DIAG_CANARY_SLACK=k builds the twin's twin whose counts
// are k too high: it MUST trip the canary.
__device__ unsigned long long g_canary[4];
constexpr unsigned kCanary = 0x7fc0dead;   // a quiet NaN with a payload; as an integer it is no 8-bit sample either
#ifndef CVS_DIAG_CANARY_SLACK
#define CVS_DIAG_CANARY_SLACK 0
#endif
#endif
#define CVS_VMW(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
#define CVS_VMW8(a, b, c, d, e, f, g, h) CVS_VMW(a) CVS_VMW(b) CVS_VMW(c) CVS_VMW(d) CVS_VMW(e) CVS_VMW(f) CVS_VMW(g) CVS_VMW(h)
__device__ __forceinline__ void wait_vmcnt(int n)
{
#ifdef CVS_DIAG_NOWAIT   // diagnostic twin only (WRONG results: rows are read before they have landed): what the launches would run at if a wave never had to wait for its
    n = 63;              // older stores in order to see its row -- an upper bound for any scheme that decouples the input stream from vmcnt
#endif
#ifdef CVS_DIAG_CANARY
    n += CVS_DIAG_CANARY_SLACK;
#endif
    switch (n < 63 ? n : 63) {
        CVS_VMW8(0, 1, 2, 3, 4, 5, 6, 7) CVS_VMW8(8, 9, 10, 11, 12, 13, 14, 15) CVS_VMW8(16, 17, 18, 19, 20, 21, 22, 23)
        CVS_VMW8(24, 25, 26, 27, 28, 29, 30, 31) CVS_VMW8(32, 33, 34, 35, 36, 37, 38, 39) CVS_VMW8(40, 41, 42, 43, 44, 45, 46, 47)
        CVS_VMW8(48, 49, 50, 51, 52, 53, 54, 55) CVS_VMW8(56, 57, 58, 59, 60, 61, 62, 63)
    }
}
#undef CVS_VMW8
#undef CVS_VMW

// a 64-bit value that is the same in every lane, moved to SGPRs (the compiler cannot prove that a
// value loaded from the per-frame table is wave-uniform; without this every use becomes a VGPR
// address + a readfirstlane "waterfall" loop around each buffer instruction)
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// ---------------------------------------------------------------------------------------
// Which tile does this workgroup filter?  STATIC orders: computed from blockIdx (static_tile).  DYNAMIC order (block_order =
// kOrderDynamic, round 4): the first 90 % of the tiles are dealt exactly like the plain order (tile = workgroup index); the
// TAIL of the launch is handed out on demand -- eight queues in device memory, one per XCD; queue q holds the tail tiles q,
// q + 8, q + 16, ... (what the dispatcher's round-robin would have given XCD q anyway); a late workgroup takes the next tile of
// its own XCD's queue and, when that is empty, of the queues of the others.  The grid has a quarter more tail workgroups than
// tail tiles: the dispatcher deals every XCD the same number of workgroups, so an XCD that is ahead (on this box, with this
// kernel, with this placement of the planes) works off its own queue early and spends its surplus workgroups on the tiles of
// the others, whose surplus workgroups find nothing and leave.  Nothing about the speed of an XCD is assumed (rounds 2-4
// also dealt fixed even : odd shares, which helped in some processes and hurt in others; removed in round 5).  Cost: one returning atomic
// (~1 us) and one barrier at the START of a tail workgroup; the body is the same code as for the static orders.  Handing out
// ALL tiles this way was measured too: 20-40 % slower (every workgroup then starts with that microsecond).
// Measured (profiles/r04_order_probe.txt, six processes on one box): level with the plain order where that is at its best,
// +3 % for the 12- and 20-plane launches in some processes, +2 % for the G4 pair launch.
// Two sets of queues per handle, used alternately: a launch takes its tickets from one set and zeroes the OTHER one (which
// the previous launch of the handle used; launches of a handle are ordered on its stream), so every launch finds its set at
// zero without a host-side step, a reset kernel or a "last one out" protocol inside the launch.  (Under stream capture -- a graph
// replays the same set every time -- and for a queue slot that has served another state block the launcher brackets the launch with
// k_reset_queues launches instead: dynamic_queues below.  Kernels, not memset nodes: a fill's stores sit in an XCD's L2 while the queue
// atomics execute at the memory side.)
// ---------------------------------------------------------------------------------------
constexpr int kQueueStride = 16;    // unsigned ints between two queue heads (64 B); head q at [q * 16], "all queues empty" flag at [8 * 16]
constexpr int kQueueSetUints = 256; // one set of queues: 1 KiB; a handle's slot holds two

__device__ __forceinline__ int take_tile(const BasisArgs& a, int q0, int ntiles)   // ntiles = tiles in the queues (the tail)
{
    unsigned k = __hip_atomic_fetch_add(a.tile_ctr + q0 * kQueueStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k < (1u << 27) && (int)k * 8 + q0 < ntiles) return (int)k * 8 + q0;
    // own queue empty.  Has somebody already found ALL of them empty?  (one load instead of seven more atomics for most
    // of the surplus workgroups)
    if (__hip_atomic_load(a.tile_ctr + 8 * kQueueStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return -1;
    for (int i = 1; i < 8; ++i) {   // help the others
        const int q = (q0 + i) & 7;
        k = __hip_atomic_fetch_add(a.tile_ctr + q * kQueueStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k < (1u << 27) && (int)k * 8 + q < ntiles) return (int)k * 8 + q;
    }
    __hip_atomic_store(a.tile_ctr + 8 * kQueueStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return -1;
}

// static orders: the tile of this workgroup; false = none (padding workgroups of the XCD-column grid)
__device__ __forceinline__ bool static_tile(const BasisArgs& a, int& bx, int& by)
{
    // workgroup -> (column block bx, row band by).  block_order = 0: plain row-major grid (the default).
    bx = blockIdx.x;
    by = blockIdx.y;
    if (a.block_order == kOrderXcdColumns) {
        // every XCD owns a contiguous range of column blocks and walks it row-major, so horizontally AND vertically adjacent
        // tiles share an L2 and the line at a tile's left / right edge is not fetched from HBM by two XCDs: on a stream of
        // fresh images the L2 fetch drops from 1.24 x to 1.11 x the image (rocprofv3 FETCH_SIZE) at the same launch time.
        // (workgroup b runs on XCD b % 8 on every launch observed; only speed depends on it)
        const int cpx = (a.grid_x + 7) >> 3, xcd = blockIdx.x & 7;
        const int k = blockIdx.x >> 3;
        by = k / cpx;
        bx = xcd * cpx + (k - by * cpx);
        if (bx >= a.grid_x || by >= a.grid_y) return false;
    }
    return true;
}

// the tile of this workgroup: (bx, by, z); false = none.  s_tile: one int of LDS shared by the workgroup.
__device__ __forceinline__ bool pick_tile(const BasisArgs& a, int* s_tile, int& bx, int& by, unsigned& z)
{
    z = blockIdx.z;
    if (a.block_order != kOrderDynamic) return static_tile(a, bx, by);
    const int per_z = a.grid_x * a.grid_y, ntiles = per_z * a.dyn_nz;
    if (blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 64 + 9)   // the set the NEXT launch of this handle will use
        __hip_atomic_store(a.tile_ctr_next + (threadIdx.x - 64) * kQueueStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the first dyn_static tiles are dealt like the plain order (tile = workgroup index, no ticket, no barrier): only the TAIL
    // of the launch -- where an XCD that is ahead would otherwise go idle -- is handed out through the queues
    int tl = (int)blockIdx.x;
    if (tl >= a.dyn_static) {
        if (threadIdx.x == 0) {
            const int q0 = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u);   // HW_REG_XCC_ID: the XCD this workgroup runs on
            const int t = take_tile(a, q0, ntiles - a.dyn_static);
            *s_tile = t < 0 ? t : t + a.dyn_static;
        }
        __syncthreads();
        tl = __builtin_amdgcn_readfirstlane(*s_tile);   // the same in every lane: keep it (and bx, by, z) in scalar registers
    }
    if (tl < 0) return false;
    z = (unsigned)(tl / per_z);
    const int r = tl - (int)z * per_z;
    by = r / a.grid_x;
    bx = r - by * a.grid_x;
    return true;
}

// the kernel body: one wave filters one strip -- addressing, the hand-counted waits of the input ring and the stores; the arithmetic on
// values in registers is row_pass / column_pass / pyr_* / pipe_values above.  `line` = this wave's LDS ring ((2W+1) x kRingLine floats),
// `zframe` = frame index of a batched launch.
// WPB = waves (adjacent 64-column strips of one row band) per workgroup.  Always 4: 8-wave workgroups were built
// and measured on one handle (tools/ab_same.py) -- no gain for any variant, -1..-6 % for the 12/20-plane ones; one- and
// two-wave workgroups fill the wave slots better (no slot waits for the slowest of four) but lose 3-12 % on the G2 legs and
// 7 % on fresh images (profiles/r03_wpb_probe.txt): the four strips of a workgroup write 1 KiB of every plane row from one CU.
#ifdef CVS_DIAG_CANARY
#define CVS_BST(ST, ...) (++st_tally, bst<ST>(__VA_ARGS__))
#else
#define CVS_BST(ST, ...) bst<ST>(__VA_ARGS__)
#endif
// Fused scalar steer (G2.cpp:137-145 / G4.cpp:114-122) of one output row: g and / or h from the bank's planes b and the host-computed weights,
// stored to the caller's two planes -- whose pointer and pitch are read from the kernel arguments per row (kernarg_fresh: kept in scalar
// registers across the row loop they were spilled to vector lanes).  `st_tally`: the canary twin's store count (dead in the product).
template <class B, bool STREAM>
__device__ __forceinline__ void store_steered(const BasisArgs& a, const float (&b)[B::NB], const unsigned xbr, const unsigned yo, [[maybe_unused]] unsigned& st_tally)
{
    constexpr int NB = B::NB;
    const kernarg_ptr_t ks = kernarg_fresh();
    const rsrc_t rg = plane_rsrc(ks->steer_g, kMaxPlaneBytes);
    const rsrc_t rh = plane_rsrc(ks->steer_h, kMaxPlaneBytes);
    const unsigned og = yo * (unsigned)(ks->steer_g_pitch * sizeof(float));
    const unsigned oh = yo * (unsigned)(ks->steer_h_pitch * sizeof(float));
    if constexpr (B::HALF == 0) {
        static_assert(B::KIND == 2, "the whole-bank form exists for G2 only");
        float gq, hq;
        g2_steer_weights(b, a.steer_w, gq, hq);
        CVS_BST(STREAM, rg, xbr, og, gq);
        CVS_BST(STREAM, rh, xbr, oh, hq);
    } else if constexpr (B::HALF == 1) {  // the G sum of G2.cpp:143 / G4.cpp:120, left to right
        float gq = __fadd_rn(__fmul_rn(a.steer_w[0], b[0]), __fmul_rn(a.steer_w[1], b[1]));
#pragma unroll
        for (int p = 2; p < NB; ++p) gq = __fadd_rn(gq, __fmul_rn(a.steer_w[p], b[p]));
        CVS_BST(STREAM, rg, xbr, og, gq);
    } else {  // the H sum of G2.cpp:144 / G4.cpp:121
        float hq = __fadd_rn(__fmul_rn(a.steer_w[B::PLANE0], b[0]), __fmul_rn(a.steer_w[B::PLANE0 + 1], b[1]));
#pragma unroll
        for (int p = 2; p < NB; ++p) hq = __fadd_rn(hq, __fmul_rn(a.steer_w[B::PLANE0 + p], b[p]));
        CVS_BST(STREAM, rh, xbr, oh, hq);
    }
}

// The pipeline's outputs of one row to the caller's own planes (one image, or a frame table).  One image (FROM_KERNARG): the eight planes are
// pointer + pitch each -- 32 SGPRs that the row loop has not got; kept across it they are spilled to vector-register lanes, and every
// v_readlane_b32 costs the SIMD as much as a packed multiply-add (180 of them per output row were 38 % of this variant's vector time).  They are
// read from the kernel-argument segment again in every row instead (scalar cache); the record count of the resource is the constant maximum
// -- the range check is only there to drop the lanes right of the image.
template <bool STREAM, bool FEAT3, bool FROM_KERNARG>
__device__ __forceinline__ void store_pipe_planes(const PlaneRef (&pipe_out)[8], const float (&q)[8], const unsigned xbr, const unsigned yo, [[maybe_unused]] unsigned& st_tally)
{
#pragma unroll
    for (int k = FEAT3 ? 5 : 0; k < 8; ++k) {
        PlaneRef po = pipe_out[k];
        if constexpr (FROM_KERNARG) {
            const kernarg_ptr_t ka = kernarg_fresh();
            po.p = ka->pipe_out[k].p;
            po.pitch = ka->pipe_out[k].pitch;
        }
        if (FEAT3 || po.p) CVS_BST(STREAM, plane_rsrc(po.p, kMaxPlaneBytes), xbr, yo * (unsigned)(po.pitch * sizeof(float)), q[k]);
    }
}

// ---------------------------------------------------------------------------------------
// The pieces of a row step that work on values in registers only (no addressing, no stores); basis_body below strings them together.
// All are inlined into the unrolled row loop: `j` is a constant there and every window slot a fixed register.
// ---------------------------------------------------------------------------------------
// Row pass: the 2W+1 samples of one input row (P[i] = {sample at +i, sample at -i}) -> window slot j of every row kernel.
template <class B, bool PK, class T>
__device__ __forceinline__ void row_pass(const T& t, const f2 (&P)[B::W + 1], const int j, f2 (&win2)[B::NRP][2 * B::W + 1],
                                         [[maybe_unused]] float (&win1)[B::NS ? B::NS : 1][2 * B::W + 1])
{
    constexpr int W = B::W;
    f2 SD[W + 1];   // {sum_i, dif_i}
#pragma unroll
    for (int i = 1; i <= W; ++i) SD[i] = pk_sumdif<PK>(P[i]);
#pragma unroll
    for (int k = 0; k < B::NRP; ++k) {
        f2 acc = pk_mul<PK, false>(t.tp(k, W), SD[W]);
#pragma unroll
        for (int i = W - 1; i >= 1; --i) acc = pk_fma<PK, false>(t.tp(k, i), SD[i], acc);
        // the centre tap of an odd kernel is +0.0 (tp[k][0].y): the CPU row filter still multiplies it in, which
        // matters only for non-finite pixels (0 * Inf = NaN) -- keep that footprint identical
        acc = pk_fma<PK, false>(t.tp(k, 0), P[0], acc);
        // (pinned here: the compiler would otherwise sink the row pass of the priming steps into the conditional column-pass blocks
        // that use it, and keep the thirteen samples of every such step alive instead of its results)
        asm volatile("" : "+v"(acc));
        win2[k][j] = acc;
    }
#pragma unroll
    for (int q = 0; q < B::NS; ++q) {
        float acc = t.ts(q, W) * SD[W].x;
#pragma unroll
        for (int i = W - 1; i >= 1; --i) acc = fmaf(t.ts(q, i), SD[i].x, acc);
        acc = fmaf(t.ts(q, 0), P[0].x, acc);
        asm volatile("" : "+v"(acc));
        win1[q][j] = acc;
    }
}

// Column pass on the window: the newest row is slot j, the centre row W back = slot (j + 1 + W) % NT.  b[p] = basis plane p of the bank at
// the centre row.
template <class B, bool PK, class T>
__device__ __forceinline__ void column_pass(const T& t, const f2 (&win2)[B::NRP][2 * B::W + 1],
                                            [[maybe_unused]] const float (&win1)[B::NS ? B::NS : 1][2 * B::W + 1], const int j, float (&b)[B::NB])
{
    constexpr int W = B::W, NT = 2 * W + 1;
#pragma unroll
    for (int q = 0; q < B::NPP; ++q) {
        constexpr auto slot = [](int jj, int d) constexpr { return (jj + 1 + W + d + NT) % NT; };
        const PairOp o = B::pp(q);
        f2 acc;
        // lo half: column kernel odd when o.swap, hi half: odd when not
        if (o.swap) {
            acc = pk_mul<PK, true>(t.tp(o.t, W), pk_addsub<PK, true, false>(win2[o.w][slot(j, W)], win2[o.w][slot(j, -W)]));
#pragma unroll
            for (int i = W - 1; i >= 1; --i)
                acc = pk_fma<PK, true>(t.tp(o.t, i), pk_addsub<PK, true, false>(win2[o.w][slot(j, i)], win2[o.w][slot(j, -i)]), acc);
            acc.y = fmaf(t.tp(o.t, 0).x, win2[o.w][slot(j, 0)].y, acc.y);
        } else {
            acc = pk_mul<PK, false>(t.tp(o.t, W), pk_addsub<PK, false, true>(win2[o.w][slot(j, W)], win2[o.w][slot(j, -W)]));
#pragma unroll
            for (int i = W - 1; i >= 1; --i)
                acc = pk_fma<PK, false>(t.tp(o.t, i), pk_addsub<PK, false, true>(win2[o.w][slot(j, i)], win2[o.w][slot(j, -i)]), acc);
            acc.x = fmaf(t.tp(o.t, 0).x, win2[o.w][slot(j, 0)].x, acc.x);
        }
        b[o.lo] = acc.x;
        b[o.hi] = acc.y;
    }
#pragma unroll
    for (int q = 0; q < B::NSP; ++q) {   // planes outside the pairs: an even row kernel, an even column kernel, plain instructions
        constexpr auto slot = [](int jj, int d) constexpr { return (jj + 1 + W + d + NT) % NT; };
        const SingleOp o = B::sp(q);
        auto wv = [&](int sl) { return o.w >= 0 ? win2[o.w >= 0 ? o.w : 0][sl].x : win1[o.w >= 0 ? 0 : -1 - o.w][sl]; };
        auto tap = [&](int i) { return o.t >= 0 ? t.tp(o.t >= 0 ? o.t : 0, i).x : t.ts(o.t >= 0 ? 0 : -1 - o.t, i); };
        float acc = tap(W) * (wv(slot(j, W)) + wv(slot(j, -W)));
#pragma unroll
        for (int i = W - 1; i >= 1; --i) acc = fmaf(tap(i), wv(slot(j, i)) + wv(slot(j, -i)), acc);
        b[o.plane] = fmaf(tap(0), wv(slot(j, 0)), acc);
    }
}

// The callers' sequence (test/test.cpp:86-90) on values still in registers: steer at theta_dom, oriented energy, magnitude / phase, the
// three feature maps -> q = {g2, h2, e, magnitude, phase, edges, dark lines, bright lines}.  need_e: evaluate the energy (else q[2] = 0);
// on_e: find*(e, phase) instead of find*(magnitude, phase).
template <bool FEAT3>
__device__ __forceinline__ void pipe_values(const float (&b)[7], float th, float c1, float c2, float c3, bool need_e, int amode, bool on_e, float (&q)[8])
{
    // theta_dom in (-pi/2, pi/2]: the bounded cos/sin path, no library call
    g2_steer_angle<true>(b, th, q[0], q[1]);
    q[2] = 0.f;
    if (need_e) {
        float s2, cc2;
        sincos_small(__fmul_rn(th, 2.0f), s2, cc2);
        q[2] = __fadd_rn(__fadd_rn(c1, __fmul_rn(c2, cc2)), __fmul_rn(c3, s2));
    }
    mag_phase(q[0], q[1], amode, q[3], q[4]);
    const float en = on_e ? q[2] : q[3];
    float le, ld, lb;
    phase_lambda3<true>(q[4], le, ld, lb);   // one cos / sin pair for the three maps
    q[5] = __fmul_rn(en, le);
    q[6] = __fmul_rn(en, ld);
    q[7] = __fmul_rn(en, lb);
}

// cv::pyrDown on the rows the wave has staged anyway (launch_pyr_down's arithmetic, op for op): hw = the last five horizontally blurred rows
// ([1 4 6 4 1] at this lane's column), newest last; pyr_column = the vertical blur of the five, before the 1/256.
template <int WP1>
__device__ __forceinline__ void pyr_shift_in(float (&hw)[5], const f2 (&P)[WP1])
{
    hw[0] = hw[1]; hw[1] = hw[2]; hw[2] = hw[3]; hw[3] = hw[4];
    hw[4] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(P[0].x, 6.0f), __fmul_rn(__fadd_rn(P[1].y, P[1].x), 4.0f)), P[2].y), P[2].x);
}
__device__ __forceinline__ float pyr_column(const float (&hw)[5])
{
    return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(hw[2], 6.0f), __fmul_rn(__fadd_rn(hw[1], hw[3]), 4.0f)), hw[0]), hw[4]);
}

template <class B, int FLAGS, bool STREAM, int BATCH, bool ONE, int WPB, bool U8 = false, bool LIT = false>
__device__ __forceinline__ void basis_body(const BasisArgs& a, const Folded<B>& t, float* line, int zframe, const int bx, const int by)
{
    constexpr unsigned EB = U8 ? 1u : 4u;   // bytes per input sample
    constexpr int W = B::W, NT = 2 * W + 1, NB = B::NB;
    // packed arithmetic everywhere but in the pipeline variants (F_PIPE): a packed instruction takes the whole SIMD, a plain one with a scalar tap
    // leaves its second half to another wave's tap-free instructions, and the pipeline's long epilogue is made of those (see "Packed f32").
    // Measured: the three-maps-only batch 9 % slower packed; the single-image pipeline level on two boxes and 11 % slower packed on a third,
    // the eight-outputs-only batch 4 % slower packed (profiles/r05_packed_ab.txt, table 7).  The orientation-only epilogue (full setup) is
    // level to 1 % ahead packed and stays packed.
    constexpr bool PK = (FLAGS & F_PIPE) == 0;
    // LIT: the taps are the reference's defaults, compiled in as literals (TapsLitG2) -- the plain-arithmetic pipeline variants of the G2 bank only
    static_assert(!LIT || (!PK && B::KIND == 2 && B::HALF == 0), "literal taps exist for the pipeline variants of the whole G2 bank");
    using Taps = std::conditional_t<LIT, TapsLitG2, TapsArg<B>>;
    const Taps taps = [&]() -> Taps { if constexpr (LIT) return TapsLitG2{}; else return TapsArg<B>{t}; }();
    // vector-memory instructions per output row that EVERY launch of this variant issues (state planes, fused steer, the three
    // maps of FEAT3; outputs selected at run time are not counted): a lower bound is all the hand-counted waits need
    constexpr int S_ROW = (((FLAGS & F_NOSTATE) == 0 && (FLAGS & F_PYRONLY) == 0) ? NB : 0) +
                          (((FLAGS & F_ORIENT) != 0 && B::KIND == 2 && (FLAGS & F_NOSTATE) == 0 && (FLAGS & F_PYRONLY) == 0) ? 5 : 0) +
                          (((FLAGS & F_STEER) != 0 && (FLAGS & F_PYRONLY) == 0) ? (B::HALF == 0 ? 2 : 1) : 0) + ((FLAGS & F_FEAT3) != 0 ? 3 : 0);
    constexpr int VM_ROWS = 2 * (NT - 1);   // loads of the NT - 1 rows issued after a row's own

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int x0 = (bx * WPB + wv) * 64;
    if (x0 >= a.cols) return;  // wave-uniform; between two tiles the waves of a workgroup do not wait for each other
#ifdef CVS_DIAG_STAMPS
    unsigned long long* stamp = a.diag ? a.diag + ((size_t)(by * a.grid_x + bx) * WPB + wv) * 4 : nullptr;
    [[maybe_unused]] bool stamped_first = false;
    if (stamp && lane == 0) {
        stamp[0] = __builtin_amdgcn_s_memrealtime();
        stamp[3] = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;  // HW_REG_XCC_ID[3:0]: the XCD this wave runs on
    }
#ifdef CVS_DIAG_CLOCK
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime();  // shader-clock ticks; stamp[1] then holds the ticks this wave lived
#endif
#endif
    // Planes of 2 GiB and more are filtered in row bands, one launch per band: the host shifts every plane
    // pointer down by row_base rows, so that the 32-bit buffer offsets of the band (halo included) stay
    // below 2 GiB, and the launch covers output rows [row_lo, row_hi).  Such planes never fit the
    // single-resource (ONE) form, which therefore keeps the plain arithmetic.
    const int rbase = ONE ? 0 : a.row_base;
    const int y0 = (ONE ? 0 : a.row_lo) + by * a.strip_rows;
    const int yend = min(y0 + a.strip_rows, ONE ? a.rows : a.row_hi);
    const int x = x0 + lane;
    const bool xin = x < a.cols;
    // per-lane byte offsets, fixed for the whole strip; kLaneOff = "this lane does not take part" (hardware range check)
    const unsigned xb = xin ? (unsigned)x * 4u : kLaneOff;
    // input columns (REFLECT_101, bytes of the image's own type): lane l fetches column x0 - W + l into word l of the line, lanes
    // < 2W also column x0 - W + 64 + l into word 64 + l.  Columns beyond cols + W feed no valid output; they are clamped into the row.
    const unsigned dmb = (unsigned)max(0, min(reflect1(x0 - W + lane, a.cols), a.cols - 1)) * EB;
    const unsigned dhb = lane < 2 * W ? (unsigned)max(0, min(reflect1(x0 - W + 64 + lane, a.cols), a.cols - 1)) * EB : kLaneOff;
    // per-frame pointers: kernel arguments, or (batched launch) entry blockIdx.z of the frame table
    const float* in_p = a.in;
    size_t in_pitch = a.in_pitch;
    float* basis_p = a.basis;
    [[maybe_unused]] float* basis2_p = a.basis2;
    float* orient_p = a.orient;
    PlaneRef pipe_out[8];
    [[maybe_unused]] rsrc_t r_out = plane_rsrc(nullptr, 0);
    if constexpr (BATCH == 2) {  // regular frames, one output resource per frame: nothing but three 64-bit bases depends on the frame
        basis_p += (size_t)zframe * a.frame_stride;
        orient_p += (size_t)zframe * a.frame_stride;
        in_p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(in_p) + (size_t)zframe * a.in_frame_stride * EB);
        r_out = plane_rsrc(a.out_base + (size_t)zframe * a.out_frame_stride, a.out_bytes);
    } else if constexpr (BATCH == 1) {
        basis_p += (size_t)zframe * a.frame_stride;
        orient_p += (size_t)zframe * a.frame_stride;
        if (a.frames) {  // per-frame pointers from the device table
            const BatchFrame* fr = a.frames + zframe;
            in_p = reinterpret_cast<const float*>(uniform64(reinterpret_cast<unsigned long long>(fr->in)));
            in_pitch = uniform64(fr->in_pitch);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                pipe_out[k].p = reinterpret_cast<float*>(uniform64(reinterpret_cast<unsigned long long>(fr->out[k].p)));
                pipe_out[k].pitch = uniform64(fr->out[k].pitch);
            }
        } else {  // regularly strided frames (one [n, H, W] block in, one [n, K, H, W] block out): scalar arithmetic only
            in_p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(in_p) + (size_t)zframe * a.in_frame_stride * EB);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                pipe_out[k].p = a.pipe_out[k].p ? a.pipe_out[k].p + (size_t)zframe * a.out_frame_stride : nullptr;
                pipe_out[k].pitch = a.pipe_out[k].pitch;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) pipe_out[k] = PlaneRef{nullptr, 0};   // one image: read per row (kernarg_fresh)
    }
    // buffer resources (wave-uniform): input plane, state planes
    const size_t plane_bytes = (size_t)(a.rows - rbase) * a.pitch * sizeof(float);
    const unsigned pitch_b = (unsigned)(a.pitch * sizeof(float));
    const i4_t r_in_dma = raw_rsrc(in_p, (size_t)(a.rows - rbase) * in_pitch * EB);
    const unsigned ring_base = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(line));   // LDS byte address of this wave's ring
    const unsigned lds_lane = (unsigned)reinterpret_cast<uintptr_t>(line) + (unsigned)lane * 4u;            // ... of this lane's first word in line 0
    // ONE: the frame's whole state block (basis + orientation planes, one allocation) is a single
    // resource and the plane is part of the scalar offset -- 4 SGPRs instead of 4 per plane, which is
    // what keeps the 20-plane pipeline variant from spilling SGPRs.  Needs the block to be < 2 GiB;
    // larger images use one resource per plane.
    const rsrc_t r_state = plane_rsrc(basis_p, a.state_bytes);
    const unsigned pstride_b = (unsigned)(a.plane_stride * sizeof(float));
    // G4: the planes h4a..h4f (state planes 5..10) are a second group with a row pitch and plane stride of its own
    constexpr int SPLIT = B::KIND == 4 ? 5 : 1 << 20;                       // first plane of the second group
    constexpr bool ANY_A = B::PLANE0 < SPLIT, ANY_B = B::PLANE0 + NB > SPLIT;
    [[maybe_unused]] const unsigned pitch2_b = (unsigned)(a.pitch2 * sizeof(float));
    [[maybe_unused]] const unsigned pstride2_b = (unsigned)(a.plane_stride2 * sizeof(float));
    [[maybe_unused]] const unsigned off2_b = ANY_B ? (unsigned)((size_t)(basis2_p - basis_p) * sizeof(float)) : 0u;   // single-resource form only
    [[maybe_unused]] const size_t plane2_bytes = (size_t)(a.rows - rbase) * a.pitch2 * sizeof(float);
    // the orientation planes are a group of their own (own row pitch, own plane stride: layout_state in cvs_handle.cpp)
    [[maybe_unused]] const unsigned opitch_b = (unsigned)(a.orient_pitch * sizeof(float));
    [[maybe_unused]] const unsigned ostride_b = (unsigned)(a.orient_stride * sizeof(float));
    [[maybe_unused]] const unsigned ooff_b = (unsigned)((size_t)(orient_p - basis_p) * sizeof(float));   // single-resource form only (block < 2 GiB)
    [[maybe_unused]] const size_t oplane_bytes = (size_t)(a.rows - rbase) * a.orient_pitch * sizeof(float);
    const unsigned in_pitch_b = (unsigned)(in_pitch * EB);

    // sliding window of row-filtered values, slot = input row mod NT: one register pair per row pair, one register per single row kernel
    f2 win2[B::NRP][NT];
    [[maybe_unused]] float win1[B::NS ? B::NS : 1][NT];
    // F_PYR: the last five horizontally blurred rows ([1 4 6 4 1] at this lane's column); even lanes of even centre
    // rows make one pixel of the next pyramid level each (launch_pyr_down's arithmetic, op for op)
    [[maybe_unused]] float hw[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] rsrc_t r_pyr = plane_rsrc(nullptr, 0);
    [[maybe_unused]] unsigned xpb = kLaneOff, pyr_pitch_b = 0;
    if constexpr ((FLAGS & F_PYR) != 0) {
        r_pyr = plane_rsrc(a.pyr_out, (size_t)((a.rows + 1) / 2) * a.pyr_pitch * sizeof(float));
        pyr_pitch_b = (unsigned)(a.pyr_pitch * sizeof(float));
        xpb = (xin && (x & 1) == 0) ? (unsigned)(x >> 1) * 4u : kLaneOff;
    }

    const int nrows_in = (yend - y0) + 2 * W;
    const int ngroups = (nrows_in + NT - 1) / NT;
    [[maybe_unused]] unsigned st_tally = 0;   // (the canary twin's store count; dead in the product)
#ifdef CVS_DIAG_CANARY
    unsigned cn_stale = 0, cn_short = 0, cn_reads = 0, cn_rows = 0;
    // a ring line filled with the pattern (before the load that refills it is issued; the pattern must be in the LDS before that load can land)
    // (ds_write by hand: a store through the generic `line` pointer could become a FLAT store, which counts in vmcnt as well and would
    // change the very counts under test)
    auto poison = [&](int jl) {
        const unsigned ad = lds_lane + (unsigned)(jl * kRingLine * 4), pat = kCanary;
        asm volatile("ds_write_b32 %0, %1" : : "v"(ad), "v"(pat) : "memory");
        const unsigned ad2 = lane < 2 * W ? ad + 256u : ad;   // (no branch: lanes without a halo word write their own word twice)
        asm volatile("ds_write_b32 %0, %1" : : "v"(ad2), "v"(pat) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
#pragma unroll
    for (int j = 0; j < NT; ++j) poison(j);
#endif
    // G2: the plain basis / orientation / fused-steer variants (FLAGS 0..3) take the strength-reduced scalar bookkeeping and
    // the per-lane plane offsets as well (late round 3; ISA per nine rows, single-resource form: M1 995 -> 975 vector and
    // 655 -> 515 scalar instructions, no SGPR spills left; the headline's fused steer 1181 -> 1113 / 734 -> 583, spill moves
    // 102 -> 28, 97 VGPRs = four waves per SIMD instead of five; headline loop +1-3 % on a plain
    // block, `profiles/r03_g2_sred_probe.txt`).  The pipeline variants keep the old bookkeeping: with the new one they spill.
    constexpr bool SRED = B::SRED || (B::KIND == 2 && FLAGS < 4);
    constexpr bool VOFF = B::VOFF || (B::KIND == 2 && FLAGS < 4);
    // Scalar bookkeeping of the row loop, strength-reduced for the banks that ask for it (B::SRED; round 3: the scalar unit
    // is shared by the CU's SIMDs and the G4 kernels feel every scalar instruction -- the pair kernel went from 1574 to 1213
    // scalar and from 3461 to 3347 vector instructions per 13 rows, its SGPR spills from 170 to 46 lane moves): the prefetch row offset and the output row offset advance by one pitch per
    // row instead of being rebuilt from the row number (reflect + multiply: 7 scalar instructions), and "is this an output
    // row" is one unsigned compare.  In the loop the prefetched row y0 - W + (g + 1) NT + j is never above the image
    // (>= y0 + W + 1), so only the lower border reflects: rows >= a.rows mirror to 2 rows - 2 - row, i.e. ro_mir - ro_lin.
    [[maybe_unused]] unsigned ro_lin = (unsigned)(y0 - W + NT - rbase) * in_pitch_b;
    // 32-bit offsets: every row a launch really reads lies below 2 GiB from the (band-shifted) plane pointer, so ro_lin and a
    // legitimate mirror offset fit; the image's END need not (a band high up in a plane of many GiB): the limit saturates,
    // "never reflect", and ro_mir is only ever used modulo 2^32 where the true value fits
    const unsigned long long ro_lim64 = (unsigned long long)(unsigned)(a.rows - rbase) * in_pitch_b;
    [[maybe_unused]] const unsigned ro_lim = ro_lim64 > 0xffffffffull ? 0xffffffffu : (unsigned)ro_lim64;
    [[maybe_unused]] const unsigned ro_mir = (unsigned)(2 * a.rows - 2 - 2 * rbase) * in_pitch_b;
    [[maybe_unused]] unsigned oi_run = 0u - (unsigned)(2 * W);                   // output row relative to y0 (wraps below 0)
    [[maybe_unused]] unsigned orow_run = ((unsigned)(y0 - rbase) - (unsigned)(2 * W)) * pitch_b;  // its byte offset in a state plane
    [[maybe_unused]] unsigned orow_o_run = ((unsigned)(y0 - rbase) - (unsigned)(2 * W)) * opitch_b;  // ... and in an orientation plane
    const unsigned nout = (unsigned)(yend - y0);
    [[maybe_unused]] unsigned xbp[NB];
    if constexpr (ONE && VOFF) {
#pragma unroll
        for (int p = 0; p < NB; ++p)   // kLaneOff + offset stays out of range
            xbp[p] = xb + (B::PLANE0 + p >= SPLIT ? off2_b + (unsigned)(B::PLANE0 + p - SPLIT) * pstride2_b : (unsigned)(B::PLANE0 + p) * pstride_b);
    }
    [[maybe_unused]] unsigned orow2_run = ((unsigned)(y0 - rbase) - (unsigned)(2 * W)) * pitch2_b;   // row offset in the second group

#pragma unroll
    for (int j = 0; j < NT; ++j)   // the first 2W+1 rows, one per ring line
        dma_row<U8>(r_in_dma, dmb, dhb, (unsigned)(reflect1(y0 - W + j, a.rows) - rbase) * in_pitch_b, ring_base + (unsigned)(j * kRingLine * 4));
    // New images (BasisArgs::warm_k > 0): the waves of the first warm_bands row bands of the launch also touch, 64 columns each, the
    // rows of warm_k bands further down -- band warm_bands + by * warm_k + k -- behind their own first rows: the rest of the image is
    // requested from HBM while the launch is young and comes out of the Infinity Cache when its tiles run, so that most of the
    // launch streams its writes without reads mixed in (profiles/r05_fresh_warm.txt).  Results do not depend on it.
    int nwarm = 0;   // wave-uniform
    if (a.warm_k > 0 && by < a.warm_bands) {
        const int rows_end = ONE ? a.rows : a.row_hi;
        for (int k = 0; k < a.warm_k; ++k) {
            const int wy = (ONE ? 0 : a.row_lo) + (a.warm_bands + by * a.warm_k + k) * a.strip_rows;
            for (int j = 0; j < a.strip_rows && wy + j < rows_end; ++j, ++nwarm)
                dma_warm<U8>(r_in_dma, (unsigned)min(x, a.cols - 1) * EB, (unsigned)(wy + j - rbase) * in_pitch_b, ring_base + (unsigned)(NT * kRingLine * 4));
        }
    }

    // One row step: input row i = g NT + j is read back from its line (slot j), the line is refilled with row i + NT, the row pass
    // feeds window slot j, and -- from step 2W on -- the column pass writes output row i - 2W.  PHASE 0 = the first group (window
    // priming: no stores before its last step, the loads it waits for were issued by the prologue), PHASE 1 = every later group;
    // the two differ in their hand-counted waits, and the priming steps carry no column-pass code at all.
    auto row_step = [&](auto phase, const int g, const int j, const bool more) __attribute__((always_inline)) {
        constexpr int PHASE = decltype(phase)::value;
        {
            f2 P[W + 1];
            {
                // Has row i = g NT + j landed in line j?  Operations issued after its halo load: the loads of the NT - 1 rows
                // that followed it, and the stores of every OUTPUT row among the NT row steps since.  Steps 2W.. are output rows;
                // so none in group 0, j + 1 of them in group 1, NT from group 2 on.  (Steps past the strip's last row wait for a
                // row nobody uses; their count may be short, which only lets them read a line that is still being written.)
                if constexpr (PHASE == 0) {
                    // window priming: behind row j's halo load lie the loads of the NT - 1 - j first rows that followed it, the nwarm
                    // read-ahead loads and the j refills issued since: VM_ROWS + nwarm.  The wait uses VM_ROWS + min(nwarm, NT) -- two
                    // immediates instead of a run-time switch; the exact count was measured in round 6 and buys nothing (M2 +0.0 %, M1 -1.1 %,
                    // 8192^2 -0.2 %: profiles/r06_fresh_exact_wait.txt), the read-ahead loads are L2 / HBM requests that have long been
                    // overtaken by the rows' own loads when the first row is needed
                    if (nwarm >= NT) wait_vmcnt(VM_ROWS + NT);
                    else wait_vmcnt(VM_ROWS);
                }
                else if (VM_ROWS + S_ROW * (j + 1) >= 63) wait_vmcnt(63);   // (a constant once the loop is unrolled)
                else if (g == 1) wait_vmcnt(VM_ROWS + S_ROW * (j + 1));
                else wait_vmcnt(VM_ROWS + S_ROW * NT);
                // this lane's window into line j: the address is made by an opaque instruction INSIDE the step -- written as plain
                // pointer arithmetic the compiler pairs the reads into ds_read2_b32 (offsets of at most 1 KiB), builds one base
                // register per pair and line, and hoists all NT x (W + 1) of them out of the loop (+50 VGPRs for G4)
                unsigned lj;
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(lj) : "v"(lds_lane), "s"((unsigned)(j * kRingLine * 4)));
                // P[i] = {sample at column offset +i, at -i} in one ds_read2_b32 each (the pass works on these pairs); P[0] = the centre twice
                lds_pairs<W>(lj, P, std::make_integer_sequence<int, W + 1>{});
                // the reads above must have left the LDS before the line is handed to the next row (the load lands hundreds of
                // cycles later, but nothing else orders it behind them)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i <= W; ++i) asm volatile("" : "+v"(P[i]));   // (the compiler does not know that the reads are asynchronous: nothing that uses them may move above the wait)
#ifdef CVS_DIAG_CANARY
                if (g * NT + j < nrows_in) {   // a row the strip really uses (steps past the last one read a line nobody needs)
                    unsigned st = 0;
#pragma unroll
                    for (int i = 0; i <= W; ++i) st += (__float_as_uint(P[i].x) == kCanary ? 1u : 0u) + ((i && __float_as_uint(P[i].y) == kCanary) ? 1u : 0u);
                    cn_stale += st;
                    ++cn_reads;
                }
                poison(j);
#endif
                if constexpr (U8) {
#pragma unroll
                    for (int i = 0; i <= W; ++i) {   // the line holds the samples as integers 0..255
                        P[i].x = (float)__float_as_uint(P[i].x);
                        P[i].y = i ? (float)__float_as_uint(P[i].y) : P[i].x;
                    }
                }
                {
                    unsigned ro;
                    if constexpr (SRED) {
                        ro = ro_lin >= ro_lim ? ro_mir - ro_lin : ro_lin;
                        ro_lin += in_pitch_b;
                    } else {
                        ro = (unsigned)(reflect1(y0 - W + (g + 1) * NT + j, a.rows) - rbase) * in_pitch_b;
                    }
                    // The load for the NEXT group is issued unconditionally (straight-line code); in the last group the lane offsets are
                    // kLaneOff: the loads are dropped by the range check without touching memory (and are still counted)
                    dma_row<U8>(r_in_dma, more ? dmb : kLaneOff, more ? dhb : kLaneOff, ro, ring_base + (unsigned)(j * kRingLine * 4));
                }
            }

            if constexpr ((FLAGS & F_PYR) != 0) {
                static_assert(W >= 2, "the pyramid level needs two columns / rows of halo");
                pyr_shift_in(hw, P);
                // newest staged row = y0 - W + i (reflected like the image rows themselves); it completes the 5-row window of
                // centre row c = y0 + i - W - 2.  This strip owns the even centre rows in [y0, yend).
                const int ci = g * NT + j - W - 2;  // centre row relative to y0, wave-uniform
                if (ci >= 0 && ci < yend - y0 && ((y0 + ci) & 1) == 0) {
                    const float v = pyr_column(hw);
                    // (cached stores: the next level's launch finds its image in the Infinity Cache.  Streaming stores + read-ahead make THIS launch
                    // 7 % faster and the five-level call 2-3 % slower: profiles/r06_c3_variants.txt, r06_c3_chain_localisation.txt)
                    CVS_BST(false, r_pyr, xpb, (unsigned)((y0 + ci) >> 1) * pyr_pitch_b, __fmul_rn(v, 1.0f / 256.0f));
                }
            }
            if constexpr ((FLAGS & F_PYRONLY) == 0) {
            row_pass<B, PK>(taps, P, j, win2, win1);

            // ---- column pass on the window; newest row is slot j, centre is W rows back ----
            // which output row this is, and whether the strip owns it (wave-uniform)
            unsigned oi = 0, orow_s = 0;
            [[maybe_unused]] unsigned orow_o_s = 0, orow2_s = 0;
            bool row_ok;
            [[maybe_unused]] int yout = 0;
            if constexpr (SRED) {
                oi = oi_run;
                orow_s = orow_run;
                ++oi_run;
                if constexpr (ANY_A) orow_run += pitch_b;
                if constexpr (ANY_B) {
                    orow2_s = orow2_run;
                    orow2_run += pitch2_b;
                }
                if constexpr ((FLAGS & F_ORIENT) != 0 && (FLAGS & F_NOSTATE) == 0) {
                    orow_o_s = orow_o_run;
                    orow_o_run += opitch_b;
                }
                row_ok = oi < nout;  // y0 <= y0 + oi < yend
            } else {
                yout = y0 + g * NT + j - 2 * W;
                row_ok = yout >= y0 && yout < yend;
            }
            if (PHASE == 0 && j < 2 * W) row_ok = false;   // window priming (what the run-time test says anyway): no column-pass code in these steps
            const unsigned xbr = xb;
            if (row_ok) {
#ifdef CVS_DIAG_CANARY
                const unsigned st_row0 = st_tally;
#endif
#ifdef CVS_DIAG_STAMPS
#ifndef CVS_DIAG_CLOCK
                if (stamp && !stamped_first) { stamped_first = true; if (lane == 0) stamp[1] = __builtin_amdgcn_s_memrealtime(); }
#endif
#endif
                // outputs-only batch: pitch, mask and plane offsets of the outputs are read from the kernel arguments HERE, a column pass and an
                // epilogue ahead of the stores that use them (read at the stores, every row waited for the scalar cache), and not kept across rows
                [[maybe_unused]] unsigned b2_pitch_b = 0, b2_mask = 0, b2_off[8] = {};
                if constexpr (BATCH == 2 && (FLAGS & F_NOSTATE) != 0 && (FLAGS & F_PIPE) != 0) {
                    const kernarg_ptr_t ka = kernarg_fresh();
                    b2_pitch_b = (unsigned)(ka->out_pitch * sizeof(float));
                    b2_mask = (FLAGS & F_FEAT3) != 0 ? 0xE0u : ka->out_mask;
#pragma unroll
                    for (int k = (FLAGS & F_FEAT3) != 0 ? 5 : 0; k < 8; ++k) b2_off[k] = ka->out_off[k];
                }
                float b[NB];
                column_pass<B, PK>(taps, win2, win1, j, b);
                // lanes right of the image carry kLaneOff in xb: their stores are dropped by the range check
                // output row relative to the plane pointers, and its byte offset in a state plane
                unsigned yo, orow;
                [[maybe_unused]] unsigned orow_o = 0, orow2 = 0;
                if constexpr (SRED) {
                    yo = (unsigned)(y0 - rbase) + oi;
                    orow = orow_s;
                    orow_o = orow_o_s;
                    orow2 = orow2_s;
                } else {
                    yo = row_ok ? (unsigned)(yout - rbase) : 0u;
                    orow = yo * pitch_b;
                    if constexpr (ANY_B) orow2 = yo * pitch2_b;
                    if constexpr ((FLAGS & F_ORIENT) != 0 && (FLAGS & F_NOSTATE) == 0) orow_o = yo * opitch_b;
                }
                // (the state-plane stores stay here: taken out into functions over a struct of the strip's addressing they compile to the same code
                // for every single-resource instance and to 3-5 % more spill moves for the per-plane ones -- round 6, tools/isa_summary.py)
                if constexpr ((FLAGS & F_NOSTATE) == 0) {
#pragma unroll
                    for (int p = 0; p < NB; ++p) {
                        const bool second = B::PLANE0 + p >= SPLIT;   // compile-time per plane
                        const unsigned orw = second ? orow2 : orow;
                        if constexpr (ONE && VOFF) CVS_BST(STREAM, r_state, xbp[p], orw, b[p]);
                        else if constexpr (ONE)
                            CVS_BST(STREAM, r_state, xbr, orw + (second ? off2_b + (unsigned)(B::PLANE0 + p - SPLIT) * pstride2_b : (unsigned)(B::PLANE0 + p) * pstride_b), b[p]);
                        else if (second) CVS_BST(STREAM, plane_rsrc(basis2_p + (size_t)(B::PLANE0 + p - SPLIT) * a.plane_stride2, plane2_bytes), xbr, orw, b[p]);
                        else CVS_BST(STREAM, plane_rsrc(basis_p + (size_t)(B::PLANE0 + p) * a.plane_stride, plane_bytes), xbr, orw, b[p]);
                    }
                }
                if constexpr ((FLAGS & F_ORIENT) != 0 && B::KIND == 2) {
                    // stateless pipeline: the oriented energy (and with it C1) is evaluated only when asked for
                    constexpr bool FEAT3 = (FLAGS & F_FEAT3) != 0;
                    const bool want_e = FEAT3 ? false : (BATCH == 2 ? (a.out_mask & 4u) != 0 : BATCH == 0 ? a.pipe_out[2].p != nullptr : pipe_out[2].p != nullptr);
                    const bool need_e = FEAT3 ? false : ((FLAGS & F_NOSTATE) == 0 || want_e || a.find_on_e != 0);  // wave-uniform
                    const int amode = FEAT3 ? 0 : a.atan_mode;   // FEAT3: a compile-time constant, the atan2f path is not even compiled in
                    float c1, c2, c3, th, st;
                    g2_orientation(b, amode, c1, c2, c3, th, st, need_e);
                    if constexpr ((FLAGS & F_NOSTATE) == 0) {
                        const float ov[5] = {c1, c2, c3, th, st};
#pragma unroll
                        for (int k = 0; k < 5; ++k)
                            if constexpr (ONE) CVS_BST(STREAM, r_state, xbr, orow_o + ooff_b + (unsigned)k * ostride_b, ov[k]);
                            else CVS_BST(STREAM, plane_rsrc(orient_p + (size_t)k * a.orient_stride, oplane_bytes), xbr, orow_o, ov[k]);
                    }
                    if constexpr ((FLAGS & F_PIPE) != 0) {
                        float q[8];
                        pipe_values<FEAT3>(b, th, c1, c2, c3, need_e, amode, !FEAT3 && a.find_on_e, q);
                        if constexpr (BATCH == 2) {
                            if constexpr ((FLAGS & F_NOSTATE) != 0) {   // outputs only: pitch and plane offsets were read at the top of this row's block (not kept across
                                                                        // rows; with the state planes written too that costs more scalar work than it saves)
                                const unsigned orow_out = yo * b2_pitch_b;
#pragma unroll
                                for (int k = FEAT3 ? 5 : 0; k < 8; ++k)
                                    if (FEAT3 || (b2_mask & (1u << k))) CVS_BST(STREAM, r_out, xbr, orow_out + b2_off[k], q[k]);
                            } else {
                                const unsigned orow_out = yo * (unsigned)(a.out_pitch * sizeof(float));
#pragma unroll
                                for (int k = 0; k < 8; ++k)
                                    if (a.out_mask & (1u << k)) CVS_BST(STREAM, r_out, xbr, orow_out + a.out_off[k], q[k]);
                            }
                        } else {
                            store_pipe_planes<STREAM, FEAT3, BATCH == 0>(pipe_out, q, xbr, yo, st_tally);
                        }
                    }
                }
                if constexpr ((FLAGS & F_STEER) != 0) store_steered<B, STREAM>(a, b, xbr, yo, st_tally);
#ifdef CVS_DIAG_CANARY
                ++cn_rows;
                if (st_tally - st_row0 < (unsigned)S_ROW) ++cn_short;   // S_ROW must be a LOWER bound of the stores of every output row
#endif
            }
            }  // !F_PYRONLY
        }
    };
    {
        const bool more = 1 < ngroups;  // wave-uniform
#pragma unroll
        for (int j = 0; j < NT; ++j) row_step(std::integral_constant<int, 0>{}, 0, j, more);
    }
    for (int g = 1; g < ngroups; ++g) {
        const bool more = g + 1 < ngroups;
#pragma unroll
        for (int j = 0; j < NT; ++j) row_step(std::integral_constant<int, 1>{}, g, j, more);
    }
#ifdef CVS_DIAG_CANARY
    if (cn_stale) atomicAdd(&g_canary[0], (unsigned long long)cn_stale);
    if (lane == 0) {
        if (cn_short) atomicAdd(&g_canary[1], (unsigned long long)cn_short);
        atomicAdd(&g_canary[2], (unsigned long long)cn_reads);
        atomicAdd(&g_canary[3], (unsigned long long)cn_rows);
    }
#endif
#ifdef CVS_DIAG_STAMPS
    if (stamp && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        stamp[2] = __builtin_amdgcn_s_memrealtime();
#ifdef CVS_DIAG_CLOCK
        stamp[1] = __builtin_amdgcn_s_memtime() - clk0;
#endif
    }
#endif
}

#undef CVS_BST
// (BasisArgs must stay the FIRST parameter of both strip kernels: kernarg_fresh() reads it at offset 0 of the kernel-argument segment)
template <class B, int FLAGS, bool STREAM, int BATCH = 0, bool ONE = false, int WPB = 4, bool U8 = false>
__global__ __launch_bounds__(64 * WPB, B::MIN_WAVES) void k_basis(const BasisArgs a, const Folded<B> t)
{
    __shared__ float lds[WPB][(2 * B::W + 2) * kRingLine];   // per wave: a ring of 2W+1 lines + one spare line (dma_warm)
    __shared__ int s_tile;
    // Frame batches with state kept: the frames are dispatched dealt from z_ways equal parts of the batch in turn (0, n/2, 1,
    // n/2 + 1, ... for two), so that the frames in flight together -- about ten of 1080p -- have their state planes, inputs and
    // outputs in DISTANT parts of the batch's blocks: planes written together stream faster when they come from two runs of the
    // VRAM allocator than from one, and a 3 GB batch block spans more than one run.  profiles/r03_batch_ways_probe.txt, 32 x 1080p, same handles and buffers: a "slow" block 0.677 -> 0.752 of the HBM
    // roofline, a "fast" one 0.717 -> 0.725; four / eight / sixteen parts give less (0.72 / 0.71 / 0.70).
    int bx = 0, by = 0;
    unsigned z = 0;
    if (!pick_tile(a, &s_tile, bx, by, z)) return;
    if constexpr (BATCH != 0) {
        if (a.z_ways > 1) {
            const unsigned per = ((unsigned)a.batch + a.z_ways - 1) / a.z_ways;
            z = (z % a.z_ways) * per + z / a.z_ways;
            if (z >= (unsigned)a.batch) return;
        }
    }
    basis_body<B, FLAGS, STREAM, BATCH, ONE, WPB, U8>(a, t, lds[threadIdx.x >> 6], z, bx, by);
}

// The pipeline variants with the reference's default taps compiled in (TapsLitG2): single-resource form.  `t` is still passed (same
// argument layout as k_basis: kernarg_fresh) and not read.  Same handle, alternating, sustained, three processes (profiles/r06_literal_taps.txt):
// 32 x 1080p with state +2.9 ... +3.0 %, three maps only +1.6 ... +2.0 % (single 4096^2 image +2.8 ... +4.8 %), single-image pipeline +0.5 ... +0.9 %.
template <class B, int FLAGS, bool STREAM, int BATCH, bool U8 = false>
__global__ __launch_bounds__(256, B::MIN_WAVES) void k_basis_lit(const BasisArgs a, const Folded<B> t)
{
    __shared__ float lds[4][(2 * B::W + 2) * kRingLine];
    __shared__ int s_tile;
    int bx = 0, by = 0;
    unsigned z = 0;
    if (!pick_tile(a, &s_tile, bx, by, z)) return;
    if constexpr (BATCH != 0) {   // (frames dealt from z_ways parts of the batch: see k_basis)
        if (a.z_ways > 1) {
            const unsigned per = ((unsigned)a.batch + a.z_ways - 1) / a.z_ways;
            z = (z % a.z_ways) * per + z / a.z_ways;
            if (z >= (unsigned)a.batch) return;
        }
    }
    basis_body<B, FLAGS, STREAM, BATCH, true, 4, U8, true>(a, t, lds[threadIdx.x >> 6], z, bx, by);
}

// G + H half banks in ONE launch: blockIdx.z picks the half bank (a wave-uniform branch), so both halves share
// one launch start-up / tail and read the same image rows at about the same time (L2 hits), while each
// keeps its half-size register window.
template <class BG, class BH, int FLAGS, bool STREAM, bool ONE, bool U8 = false>
__global__ __launch_bounds__(256) void k_basis_pair(const BasisArgs a, const Folded<BG> tg, const Folded<BH> th)
{
    __shared__ float lds[4][(2 * BG::W + 2) * kRingLine];
    __shared__ int s_tile;
    int bx = 0, by = 0;
    unsigned z = 0;
    if (!pick_tile(a, &s_tile, bx, by, z)) return;   // z = half bank (dynamic order: tiles of both halves come from one set of queues)
    if (z == 0) basis_body<BG, FLAGS, STREAM, false, ONE, 4, U8>(a, tg, lds[threadIdx.x >> 6], 0, bx, by);
    else basis_body<BH, FLAGS, STREAM, false, ONE, 4, U8>(a, th, lds[threadIdx.x >> 6], 0, bx, by);
}

// ---------------------------------------------------------------------------------------
// generic-width fallback (any width <= kMaxWidth, any taps): two plain passes through a
// scratch plane, one basis plane at a time.  Same structure as the CPU reference's generic RowFilter and its column
// filters (folded for mirror / anti-mirror kernels, plain otherwise).  Not tuned: the reference's defaults (G2 w=4, G4 w=6)
// never take this path.
// ---------------------------------------------------------------------------------------
struct TapVec {
    float k[kMaxTaps];
};

__global__ __launch_bounds__(256) void k_rowpass_generic(const float* in, size_t in_pitch, int rows, int cols,
                                                          float* out, size_t out_pitch, TapVec kx, int w)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= cols || y >= rows) return;
    const float* rp = in + (size_t)y * in_pitch;
    float acc = kx.k[0] * rp[reflect101(x - w, cols)];
    for (int i = 1; i <= 2 * w; ++i) acc = fmaf(kx.k[i], rp[reflect101(x - w + i, cols)], acc);
    out[(size_t)y * out_pitch + x] = acc;
}

// sym = +1 / -1: the column kernel is a mirror / anti-mirror image of itself and the CPU reference takes its folded column
// filter (SymmColumnFilter: k0 c + sum k_i (S[+i] +/- S[-i]); the anti-mirror form never touches the centre row -- which is
// what decides where a non-finite pixel shows up); 0: the plain sum over all taps
__global__ __launch_bounds__(256) void k_colpass_generic(const float* in, size_t in_pitch, int rows, int cols,
                                                          float* out, size_t out_pitch, TapVec ky, int w, int sym)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= cols || y >= rows) return;
    float acc;
    if (sym != 0) {
        acc = sym > 0 ? ky.k[w] * in[(size_t)y * in_pitch + x] : 0.0f;
        for (int j = 1; j <= w; ++j) {
            const float hi = in[(size_t)reflect101(y + j, rows) * in_pitch + x], lo = in[(size_t)reflect101(y - j, rows) * in_pitch + x];
            acc = fmaf(ky.k[w + j], sym > 0 ? hi + lo : hi - lo, acc);
        }
    } else {
        acc = ky.k[0] * in[(size_t)reflect101(y - w, rows) * in_pitch + x];
        for (int j = 1; j <= 2 * w; ++j) acc = fmaf(ky.k[j], in[(size_t)reflect101(y - w + j, rows) * in_pitch + x], acc);
    }
    out[(size_t)y * out_pitch + x] = acc;
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
template <class B>
static bool fold_taps(const float (*taps)[kMaxTaps], Folded<B>& f)
{
    constexpr int W = B::W;
    float ev[B::NE][W + 1];   // ev[r][i] = tap at offset +/-i (i = 0..W)
    float od[B::NO][W + 1];   // od[r][i] = tap at offset +i (i = 1..W); od[r][0] = +0.0f, the centre
    for (int r = 0; r < B::NE; ++r) {
        const float* k = taps[B::even_member(r)];
        for (int i = 0; i <= W; ++i) {
            if (k[W + i] != k[W - i]) return false;
            ev[r][i] = k[W + i];
        }
    }
    for (int r = 0; r < B::NO; ++r) {
        const float* k = taps[B::odd_member(r)];
        if (k[W] != 0.0f) return false;
        od[r][0] = 0.0f;
        for (int i = 1; i <= W; ++i) {
            if (k[W + i] != -k[W - i]) return false;
            od[r][i] = k[W + i];
        }
    }
    for (int i = 0; i < 2 * W + 1; ++i)
        if (taps[B::dup_a][i] != taps[B::dup_b][i]) return false;
    for (int m = 0; m < B::NTP; ++m)
        for (int i = 0; i <= W; ++i) f.tp[m][i] = f2{ev[B::te(m)][i], od[B::to(m)][i]};
    for (int q = 0; q < (B::NS ? B::NS : 1); ++q)
        for (int i = 0; i <= W; ++i) f.ts[q][i] = B::NS ? ev[B::se(q)][i] : 0.0f;
    return true;
}

bool basis_fast_path(int kind, int width, const float (*taps)[kMaxTaps])
{
    if (kind == 2 && width == BankG2::W) { Folded<BankG2> f; return fold_taps<BankG2>(taps, f); }
    if (kind == 4 && width == BankG4G::W) {
        Folded<BankG4G> fg;
        Folded<BankG4H> fh;
        return fold_taps<BankG4G>(taps, fg) && fold_taps<BankG4H>(taps, fh);
    }
    return false;
}

size_t basis_scratch_elems(int kind, int width, int rows, size_t pitch)
{
    (void)kind; (void)width;
    return (size_t)rows * pitch;  // one plane; only touched on the generic path
}

// conservative: true whenever launch_basis may take the generic path for this geometry
bool basis_may_need_scratch(int kind, int width, const float (*taps)[kMaxTaps], int rows, int cols, size_t max_pitch)
{
    if (!basis_fast_path(kind, width, taps)) return true;
    if (rows < 3 * width + 1 || cols < width + 1) return true;
    // huge planes go through the fast kernel in row bands unless not even one strip fits 2 GiB (see band_rows)
    return kMaxPlaneBytes / (max_pitch * sizeof(float)) < (size_t)(2 * width + 2);
}

// dynamic order: the last tenth of the tiles goes through the queues (the rest: tile = workgroup index) with a quarter more
// workgroups than queued tiles (see pick_tile; both figures from the sweep in profiles/r04_order_probe.txt), a multiple of 8 so
// that every XCD gets the same number
static unsigned dynamic_blocks(size_t ntiles, int* dyn_static)
{
    constexpr int pct = 25;
    int tail_pct = 10;
#ifdef CVS_DIAG_STAMPS   // diagnostic twin only (tools/k1_timeline.py): how long a tail balances the XCDs, and what that is worth
    if (const char* e = std::getenv("CVS_DIAG_TAIL_PCT")) tail_pct = std::max(1, std::min(100, std::atoi(e)));
#endif
    const size_t tail = std::max<size_t>(8, ntiles * tail_pct / 100);
    const size_t stat = ntiles > tail ? (ntiles - tail) / 8 * 8 : 0;
    *dyn_static = (int)stat;
    const size_t queued = ntiles - stat;
    return (unsigned)(stat + (queued + queued * pct / 100 + 7) / 8 * 8);
}

// one set of queues back to zero, with the same agent-scope stores the strip kernels use on them.  (NOT hipMemsetAsync: a memset
// node's fill writes through an XCD's L2 like any plain store, while the queue atomics execute at the memory side; inside a graph
// the kernel that follows the fill can find the old values there -- seen as unwritten tail tiles in one process history out of
// several, round 5.)
__global__ void k_reset_queues(unsigned* set)
{
    if (threadIdx.x < 9) __hip_atomic_store(set + threadIdx.x * kQueueStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// dynamic order: which of the handle's two sets of queues this launch uses (and which it zeroes for the next one).  Under
// stream capture the set is baked into the graph, so the launch is bracketed by two reset launches -- the set is made zero in
// front of the kernel and again behind it -- and the host-side parity is left alone: whatever mix of replays and eager launches
// follows, every launch finds its set at zero (round-4 advisor: replay, eager, replay, eager used to hand the second eager
// launch a set the replay had left exhausted, and the tail tiles were not written).
static hipError_t dynamic_queues(BasisArgs& a, hipStream_t s, bool* captured)
{
    *captured = false;
    if (a.block_order != kOrderDynamic || !a.tile_ctr) return hipSuccess;
    const int p = a.tile_parity ? (*a.tile_parity & 1) : 0;
    unsigned* base = a.tile_ctr;
    a.tile_ctr = base + p * kQueueSetUints;
    a.tile_ctr_next = base + (1 - p) * kQueueSetUints;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
        *captured = true;
        hipLaunchKernelGGL(k_reset_queues, dim3(1), dim3(64), 0, s, a.tile_ctr);
        return hipGetLastError();
    }
    if (a.tile_parity && (*a.tile_parity & 2)) {   // a recycled queue slot: its sets are whatever the previous owner left (see tile_ctr_alloc)
        hipLaunchKernelGGL(k_reset_queues, dim3(1), dim3(64), 0, s, a.tile_ctr);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        *a.tile_parity &= 1;
    }
    if (a.tile_parity) *a.tile_parity ^= 1;
    return hipSuccess;
}
static hipError_t dynamic_queues_done(const BasisArgs& a, hipStream_t s, bool captured, hipError_t launched)
{
    if (launched != hipSuccess || !captured) return launched;
    hipLaunchKernelGGL(k_reset_queues, dim3(1), dim3(64), 0, s, a.tile_ctr);
    return hipGetLastError();
}

template <class B>
static hipError_t launch_fast_impl(BasisArgs& a, const Folded<B>& f, hipStream_t s)
{
    const int strips_x = (a.cols + 63) / 64;
    const bool orient_v = a.orient != nullptr && B::KIND == 2;
    constexpr int wpb = 4;
    dim3 grid((strips_x + wpb - 1) / wpb, (a.row_hi - a.row_lo + a.strip_rows - 1) / a.strip_rows);
    a.grid_x = grid.x;
    a.grid_y = grid.y;
    a.dyn_nz = 1;
    a.warm_bands = a.warm_k > 0 ? (a.grid_y + a.warm_k) / (a.warm_k + 1) : 0;
    if (a.grid_y < 10 || a.frames) a.warm_k = 0;   // (frame batches: warm_k is set by the API layer only where it pays)
    if (a.block_order == kOrderDynamic && !a.tile_ctr) a.block_order = 0;   // no queue slot for this handle: the plain order
    if (a.block_order != kOrderDynamic && a.block_order != kOrderXcdColumns) a.block_order = 0;
    const bool dyn = a.block_order == kOrderDynamic;
    // dynamic order: the grid is set at the launch itself (CVS_LAUNCH_K): it covers the frames of a batch as well
    if (a.block_order == kOrderXcdColumns) grid = dim3(8u * (unsigned)(((a.grid_x + 7) / 8) * a.grid_y), 1);
    dim3 block(64 * wpb);
    const bool orient = orient_v;
    const bool steer = a.steer_g != nullptr && a.steer_h != nullptr;
    // the single-resource form addresses the whole image from row 0: a banded launch (a caller plane of 2 GiB or
    // more beside a small state block, e.g. a narrow column view of a huge image) must use the per-plane form,
    // which honours row_lo / row_hi / row_base
    const bool banded = a.row_lo != 0 || a.row_hi != a.rows || a.row_base != 0;
    const bool one = !banded && a.state_bytes > 0 && a.state_bytes <= kMaxPlaneBytes;
    // the handle's taps are the reference's defaults, bit for bit: the pipeline variants run the instance with the taps compiled in
    bool lit = false;
    if constexpr (B::KIND == 2 && B::HALF == 0) {
        lit = a.lit_taps != 0;
        for (int m = 0; m < B::NTP && lit; ++m)
            for (int i = 0; i <= B::W && lit; ++i)
            {   // (through plain floats: __builtin_bit_cast on an element of the vector type reads element 0 whatever the index -- seen with this compiler)
                const float ex = f.tp[m][i].x, oy = f.tp[m][i].y;
                unsigned ue, uo;
                std::memcpy(&ue, &ex, sizeof ue);
                std::memcpy(&uo, &oy, sizeof uo);
                lit = ue == kLitEvenG2[B::te(m)][i] && uo == kLitOddG2[B::to(m)][i];
            }
    }
    if (a.lit_used) *a.lit_used = 0;   // set where such an instance is launched (CVS_LAUNCH_U)
    // BasisArgs::wg_per_cu: the launch asks for more LDS than it uses, so that at most that many workgroups share a CU (see cvs_tune.cpp)
    unsigned lds_pad = 0;
    if (a.wg_per_cu > 0 && a.wg_per_cu < 8) {
        constexpr unsigned kLds = 160u << 10, kStatic = (unsigned)(wpb * (2 * B::W + 2) * kRingLine * sizeof(float) + 64);
        const unsigned n = (unsigned)a.wg_per_cu, want = (kLds / n + kLds / (n + 1)) / 2;   // N fit, N + 1 do not: the middle of that interval
        lds_pad = want > kStatic ? want - kStatic : 0;
        // never more than a workgroup may have (one or two per CU would ask for 120 / 67 KiB): the cap is then weaker than asked for
        static const unsigned max_lds = [] {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0) return (unsigned)v;
            (void)hipGetLastError();
            return 64u << 10;
        }();
        if (kStatic + lds_pad > max_lds) lds_pad = max_lds > kStatic ? max_lds - kStatic : 0;
    }
#define CVS_LAUNCH_K(...)                                                                  \
    do {                                                                                   \
        if (dyn) {                                                                         \
            a.dyn_nz = (int)grid.z;                                                        \
            grid = dim3(dynamic_blocks((size_t)a.grid_x * a.grid_y * grid.z, &a.dyn_static), 1, 1); \
        }                                                                                  \
        hipLaunchKernelGGL((__VA_ARGS__), grid, block, lds_pad, s, a, f);         \
    } while (0)
#define CVS_LAUNCH_U(FL, BATCHED, WP, U)                                                   \
    do {                                                                                   \
        if constexpr (B::KIND == 2 && B::HALF == 0 && ((FL) & F_PIPE) != 0 && (BATCHED) != 1) {   \
            if (lit && one) {                                                              \
                if (a.lit_used) *a.lit_used = 1;                                           \
                if (a.nt_stores) CVS_LAUNCH_K(k_basis_lit<B, FL, true, BATCHED, U>);       \
                else CVS_LAUNCH_K(k_basis_lit<B, FL, false, BATCHED, U>);                  \
                break;                                                                     \
            }                                                                              \
        }                                                                                  \
        if (one) {                                                                         \
            if (a.nt_stores) CVS_LAUNCH_K(k_basis<B, FL, true, BATCHED, true, WP, U>);     \
            else CVS_LAUNCH_K(k_basis<B, FL, false, BATCHED, true, WP, U>);                \
        } else {                                                                           \
            if (a.nt_stores) CVS_LAUNCH_K(k_basis<B, FL, true, BATCHED, false, WP, U>);    \
            else CVS_LAUNCH_K(k_basis<B, FL, false, BATCHED, false, WP, U>);               \
        }                                                                                  \
    } while (0)
#define CVS_LAUNCH_W(FL, BATCHED, WP)                       \
    do {                                                    \
        if (a.in_u8) CVS_LAUNCH_U(FL, BATCHED, WP, true);   \
        else CVS_LAUNCH_U(FL, BATCHED, WP, false);          \
    } while (0)
#define CVS_LAUNCH_B(FL, BATCHED) CVS_LAUNCH_W(FL, BATCHED, 4)
#define CVS_LAUNCH(FL) CVS_LAUNCH_B(FL, 0)
    if (a.frames || a.batch_regular) {  // batched caller pipeline: one launch, grid.z = frames (G2 only)
        if constexpr (B::KIND == 2 && B::HALF == 0) {
            grid.z = a.batch;
            if (a.z_ways < 1 || a.z_ways > a.batch) a.z_ways = 1;
            if (a.z_ways > 1) grid.z = (unsigned)((a.batch + a.z_ways - 1) / a.z_ways * a.z_ways);   // slots past the batch leave at once
            if (a.batch_regular && a.out_one) {
                // exactly the three feature maps of find*(magnitude, phase) with the compatible arctangent: the specialised instance
                const bool feat3 = a.no_state && a.out_mask == 0xE0u && !a.find_on_e && a.atan_mode == 0;
                if (feat3) CVS_LAUNCH_B(F_ORIENT | F_PIPE | F_NOSTATE | F_FEAT3, 2);
                else if (a.no_state) CVS_LAUNCH_B(F_ORIENT | F_PIPE | F_NOSTATE, 2);
                else CVS_LAUNCH_B(F_ORIENT | F_PIPE, 2);
            } else {
                if (a.no_state) CVS_LAUNCH_B(F_ORIENT | F_PIPE | F_NOSTATE, 1);
                else CVS_LAUNCH_B(F_ORIENT | F_PIPE, 1);
            }
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    const bool pipe = orient && a.pipe;
    const int flags = pipe ? (F_ORIENT | F_PIPE) : ((orient ? F_ORIENT : 0) | (steer ? F_STEER : 0));
    if (a.pyr_out) {  // filter this pyramid level and write the next one (launch_basis has checked that the launch qualifies)
        if constexpr (B::KIND == 2 && B::HALF == 0) {
            if (banded || pipe || steer) return hipErrorInvalidValue;
            if (orient) CVS_LAUNCH(F_ORIENT | F_PYR);
            else CVS_LAUNCH(F_PYR);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    if constexpr (B::KIND == 2) {
        switch (flags) {
            case 0: CVS_LAUNCH(0); break;
            case F_ORIENT: CVS_LAUNCH(F_ORIENT); break;
            case F_STEER: CVS_LAUNCH(F_STEER); break;
            case F_ORIENT | F_STEER: CVS_LAUNCH(F_ORIENT | F_STEER); break;
            default: {
                bool feat3 = a.no_state && !a.find_on_e && a.atan_mode == 0;
                for (int k = 0; k < 8; ++k) feat3 = feat3 && ((a.pipe_out[k].p != nullptr) == (k >= 5));
                if (feat3) CVS_LAUNCH(F_ORIENT | F_PIPE | F_NOSTATE | F_FEAT3);
                else if (a.no_state) CVS_LAUNCH(F_ORIENT | F_PIPE | F_NOSTATE);
                else CVS_LAUNCH(F_ORIENT | F_PIPE);
                break;
            }
        }
    } else {
        return hipErrorInvalidValue;   // G4 runs as the pair launch (launch_pair)
    }
#undef CVS_LAUNCH_B
#undef CVS_LAUNCH_W
#undef CVS_LAUNCH_U
#undef CVS_LAUNCH_K
#undef CVS_LAUNCH
    return hipGetLastError();
}

template <class B>
static hipError_t launch_fast(const BasisArgs& a_in, const Folded<B>& f, hipStream_t s)
{
    BasisArgs a = a_in;
    bool captured = false;
    const hipError_t qe = dynamic_queues(a, s, &captured);
    if (qe != hipSuccess) return qe;
    return dynamic_queues_done(a, s, captured, launch_fast_impl<B>(a, f, s));
}

template <class BG, class BH>
static hipError_t launch_pair_impl(BasisArgs& a, const Folded<BG>& fg, const Folded<BH>& fh, hipStream_t s)
{
    const int strips_x = (a.cols + 63) / 64;
    dim3 grid((strips_x + 3) / 4, (a.row_hi - a.row_lo + a.strip_rows - 1) / a.strip_rows, 2), block(256);
    a.grid_x = grid.x;
    a.grid_y = grid.y;
    a.dyn_nz = 2;
    a.warm_bands = a.warm_k > 0 ? (a.grid_y + a.warm_k) / (a.warm_k + 1) : 0;
    if (a.grid_y < 10) a.warm_k = 0;
    if (a.block_order == kOrderDynamic && !a.tile_ctr) a.block_order = 0;
    if (a.block_order != kOrderDynamic && a.block_order != kOrderXcdColumns) a.block_order = 0;
    const bool dyn = a.block_order == kOrderDynamic;
    // dynamic order: the grid is set at the launch (see launch_fast); tiles of both half banks share the queues
    if (a.block_order == kOrderXcdColumns) grid = dim3(8u * (unsigned)(((a.grid_x + 7) / 8) * a.grid_y), 1, 2);
    const bool steer = a.steer_g != nullptr && a.steer_h != nullptr;
    const bool banded = a.row_lo != 0 || a.row_hi != a.rows || a.row_base != 0;  // see launch_fast
    const bool one = !banded && a.state_bytes > 0 && a.state_bytes <= kMaxPlaneBytes;
#define CVS_PAIR_K(...)                                                                                         \
    do {                                                                                                        \
        if (dyn) grid = dim3(dynamic_blocks((size_t)a.grid_x * a.grid_y * 2, &a.dyn_static), 1, 1);             \
        hipLaunchKernelGGL((__VA_ARGS__), grid, block, 0, s, a, fg, fh);                         \
    } while (0)
#define CVS_PAIR(FL, ST, ON)                                                      \
    do {                                                                          \
        if (a.in_u8) CVS_PAIR_K(k_basis_pair<BG, BH, FL, ST, ON, true>);          \
        else CVS_PAIR_K(k_basis_pair<BG, BH, FL, ST, ON, false>);                 \
    } while (0)
    if (steer) {
        if (a.nt_stores) { if (one) CVS_PAIR(F_STEER, true, true); else CVS_PAIR(F_STEER, true, false); }
        else { if (one) CVS_PAIR(F_STEER, false, true); else CVS_PAIR(F_STEER, false, false); }
    } else {
        if (a.nt_stores) { if (one) CVS_PAIR(0, true, true); else CVS_PAIR(0, true, false); }
        else { if (one) CVS_PAIR(0, false, true); else CVS_PAIR(0, false, false); }
    }
#undef CVS_PAIR
#undef CVS_PAIR_K
    return hipGetLastError();
}

template <class BG, class BH>
static hipError_t launch_pair(const BasisArgs& a_in, const Folded<BG>& fg, const Folded<BH>& fh, hipStream_t s)
{
    BasisArgs a = a_in;
    bool captured = false;
    const hipError_t qe = dynamic_queues(a, s, &captured);
    if (qe != hipSuccess) return qe;
    return dynamic_queues_done(a, s, captured, launch_pair_impl<BG, BH>(a, fg, fh, s));
}

static hipError_t launch_generic(int kind, int width, const float (*taps)[kMaxTaps], const BasisArgs& a,
                                 float* scratch, hipStream_t s)
{
    const int nb = host_num_basis(kind);
    dim3 block(256), grid((a.cols + 255) / 256, a.rows);
    const size_t spitch = ((size_t)a.cols + 63) / 64 * 64;   // the scratch plane is dense whatever the layout of the state planes
    for (int p = 0; p < nb; ++p) {
        int ix, iy;
        host_basis_taps(kind, p, &ix, &iy);
        TapVec kx, ky;
        for (int i = 0; i < 2 * width + 1; ++i) { kx.k[i] = taps[ix][i]; ky.k[i] = taps[iy][i]; }
        hipLaunchKernelGGL(k_rowpass_generic, grid, block, 0, s, a.in, a.in_pitch, a.rows, a.cols, scratch, spitch, kx, width);
        int sym = 1, asym = ky.k[width] == 0.0f ? 1 : 0;
        for (int i = 1; i <= width; ++i) {
            if (ky.k[width + i] != ky.k[width - i]) sym = 0;
            if (ky.k[width + i] != -ky.k[width - i]) asym = 0;
        }
        const PlaneRef bp = basis_plane_ref(a, kind, p);
        hipLaunchKernelGGL(k_colpass_generic, grid, block, 0, s, (const float*)scratch, spitch, a.rows, a.cols, bp.p, bp.pitch, ky, width,
                           sym ? 1 : asym ? -1 : 0);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // epilogues as separate pointwise passes
    if (a.orient && kind == 2) {
        PointArgs pa{};
        pa.rows = a.rows; pa.cols = a.cols; pa.atan_mode = a.atan_mode; pa.nt_stores = a.nt_stores;
        for (int p = 0; p < 7; ++p) pa.in[p] = basis_plane_ref(a, kind, p);
        for (int i = 0; i < 5; ++i) pa.out[i] = {a.orient + (size_t)i * a.orient_stride, a.orient_pitch};
        e = launch_point(OP_G2_ORIENT, pa, s);
        if (e != hipSuccess) return e;
    }
    if (a.orient && kind == 2 && a.pipe) {
        PointArgs pa{};
        pa.rows = a.rows; pa.cols = a.cols; pa.atan_mode = a.atan_mode; pa.nt_stores = a.nt_stores;
        pa.find_on_e = a.find_on_e;
        for (int p = 0; p < 7; ++p) pa.in[p] = basis_plane_ref(a, kind, p);
        for (int i = 0; i < 3; ++i) pa.in[7 + i] = {a.orient + (size_t)i * a.orient_stride, a.orient_pitch};
        pa.in[10] = {a.orient + (size_t)3 * a.orient_stride, a.orient_pitch};
        for (int k = 0; k < 8; ++k) pa.out[k] = a.pipe_out[k];
        e = launch_point(OP_G2_PIPELINE, pa, s);
        if (e != hipSuccess) return e;
    }
    if (a.steer_g && a.steer_h) {
        PointArgs pa{};
        pa.rows = a.rows; pa.cols = a.cols; pa.atan_mode = a.atan_mode; pa.nt_stores = a.nt_stores;
        for (int p = 0; p < nb; ++p) { pa.in[p] = basis_plane_ref(a, kind, p); pa.w[p] = a.steer_w[p]; }
        pa.out[0] = {a.steer_g, a.steer_g_pitch};
        pa.out[1] = {a.steer_h, a.steer_h_pitch};
        e = launch_point(kind == 2 ? OP_G2_STEER_SCALAR : OP_G4_STEER_SCALAR, pa, s);
    }
    return e;
}

// widest row pitch (bytes) of any plane the launch touches
static size_t max_pitch_bytes(const BasisArgs& a)
{
    size_t mx = a.in_pitch > a.pitch ? a.in_pitch : a.pitch;
    if (a.basis2) mx = max(mx, a.pitch2);
    if (a.orient) mx = max(mx, a.orient_pitch);
    if (a.steer_g) { mx = max(mx, a.steer_g_pitch); mx = max(mx, a.steer_h_pitch); }
    for (int k = 0; k < 8; ++k)
        if (a.pipe && a.pipe_out[k].p) mx = max(mx, a.pipe_out[k].pitch);
    return mx * sizeof(float);
}

// single-step reflection (reflect1) needs a minimum image size; smaller images take the generic path
static bool fast_geometry_ok(const BasisArgs& a, int width)
{
    return a.rows >= 3 * width + 1 && a.cols >= width + 1;
}

// Output rows one launch may cover so that every 32-bit buffer offset (band + halo rows, relative to the
// shifted plane base) stays below 2 GiB: the whole image for ordinary planes, a band for huge ones.
// 0 = a single strip does not fit (rows of ~100 MiB and more): generic path.
static int band_rows(const BasisArgs& a, int width)
{
    const size_t fit = kMaxPlaneBytes / max_pitch_bytes(a);
    if (fit >= (size_t)a.rows) return a.rows;
    if (fit < (size_t)(2 * width + 2)) return 0;
    const int room = (int)fit - 2 * width;
    return room >= a.strip_rows ? room / a.strip_rows * a.strip_rows : room;  // very wide rows: one short strip per band
}

// true when launch_basis can emit the next pyramid level from inside the filter launch (a.pyr_out): the G2 bank at its
// default width, one un-banded launch over the whole image, no fused steer / pipeline epilogue
bool basis_fuses_pyr(int kind, int width, const float (*taps)[kMaxTaps], const BasisArgs& a)
{
    if (kind != 2 || width != BankG2::W || !basis_fast_path(kind, width, taps)) return false;
    if (!fast_geometry_ok(a, width) || band_rows(a, width) < a.rows) return false;
    if (a.out_row_hi > a.out_row_lo || a.frames || a.batch_regular || a.pipe || (a.steer_g && a.steer_h)) return false;
    return (size_t)((a.rows + 1) / 2) * a.pyr_pitch * sizeof(float) <= kMaxPlaneBytes;
}

// cv::pyrDown of one plane as a strip march (cvs_pyr_down): the F_PYR emission of the basis kernel with the filters
// compiled out -- every wave stages its rows in LDS once, keeps five blurred rows in registers and writes every second one:
// each input line is requested once per strip (the stand-alone k_pyr_down asks for it 10.5 times per output pixel through
// the caches).  Same arithmetic, bit-identical.  false = the geometry is not covered (tiny images, planes of 2 GiB and more).
bool launch_pyr_strip(const float* src, size_t spitch, int rows, int cols, float* dst, size_t dpitch, hipStream_t s, hipError_t* err)
{
    constexpr int W = BankG2::W;
    if (rows < 3 * W + 1 || cols < W + 1) return false;
    if ((size_t)rows * spitch * sizeof(float) > kMaxPlaneBytes || (size_t)((rows + 1) / 2) * dpitch * sizeof(float) > kMaxPlaneBytes) return false;
    BasisArgs a{};
    a.in = src;
    a.in_pitch = spitch;
    a.rows = rows;
    a.cols = cols;
    a.pitch = spitch;          // no state plane is touched; the resources built from these are never used
    a.plane_stride = 0;
    a.pyr_out = dst;
    a.pyr_pitch = dpitch;
    a.strip_rows = 6 * (2 * W + 1) - 2 * W;   // 46 rows: 17 % more rows staged than written, ~11 k waves at 8192^2
    a.row_lo = 0;
    a.row_hi = rows;
    a.row_base = 0;
    const int strips_x = (cols + 63) / 64;
    dim3 grid((strips_x + 3) / 4, (rows + a.strip_rows - 1) / a.strip_rows), block(256);
    a.grid_x = grid.x;
    a.grid_y = grid.y;
    Folded<BankG2> f{};
    hipLaunchKernelGGL((k_basis<BankG2, F_PYR | F_PYRONLY, false, 0, false, 4>), grid, block, 0, s, a, f);
    *err = hipGetLastError();
    return true;
}

// launch `fn(args)` once per row band, with the plane pointers shifted to the band's first halo row
template <class F>
static hipError_t for_each_band(const BasisArgs& a_in, int width, F&& fn)
{
    const int per = band_rows(a_in, width);
    // a caller-given row range (one GPU's band of a large image) is walked exactly like the whole image
    const int lo0 = a_in.out_row_hi > a_in.out_row_lo ? a_in.out_row_lo : 0;
    const int hi0 = a_in.out_row_hi > a_in.out_row_lo ? a_in.out_row_hi : a_in.rows;
    for (int lo = lo0; lo < hi0; lo += per) {
        BasisArgs a = a_in;
        if (a.strip_rows > per) a.strip_rows = per;
        a.row_lo = lo;
        a.row_hi = lo + per < hi0 ? lo + per : hi0;
        // the bottom band also reads reflected rows: rows - 2 - k, which lie above row_hi - 1, never below row_base
        a.row_base = lo > width ? lo - width : 0;
        if (per >= a.rows) a.row_base = 0;
        const size_t rb = (size_t)a.row_base;
        a.in = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.in) + rb * a.in_pitch * (a.in_u8 ? 1 : sizeof(float)));
        a.basis += rb * a.pitch;
        if (a.basis2) a.basis2 += rb * a.pitch2;
        if (a.orient) a.orient += rb * a.orient_pitch;
        if (a.steer_g) a.steer_g += rb * a.steer_g_pitch;
        if (a.steer_h) a.steer_h += rb * a.steer_h_pitch;
        for (int k = 0; k < 8; ++k)
            if (a.pipe_out[k].p) a.pipe_out[k].p += rb * a.pipe_out[k].pitch;
        const hipError_t e = fn(a);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_basis(int kind, int width, const float (*taps)[kMaxTaps], const BasisArgs& a,
                        float* scratch, hipStream_t s)
{
    if (a.frames || a.batch_regular) {  // batched launch: the API layer has already checked geometry and taps
        Folded<BankG2> f;
        if (kind != 2 || width != BankG2::W || !fold_taps<BankG2>(taps, f)) return hipErrorInvalidValue;
        BasisArgs b = a;
        b.row_lo = b.row_base = 0;
        b.row_hi = b.rows;
        return launch_fast<BankG2>(b, f, s);
    }
    if (a.pyr_out && !basis_fuses_pyr(kind, width, taps, a)) {  // not a launch the fused form covers: two launches, same values
        BasisArgs b = a;
        b.pyr_out = nullptr;
        const hipError_t e = launch_basis(kind, width, taps, b, scratch, s);
        return e != hipSuccess ? e : launch_pyr_down(a.in, a.in_pitch, a.rows, a.cols, a.pyr_out, a.pyr_pitch, s);
    }
    // 8-bit images are read by the strip kernels only; the API layer widens the image first wherever this function may take
    // another path (basis_may_need_scratch), so these are defensive
    if (a.in_u8 && (a.pyr_out || !fast_geometry_ok(a, width) || band_rows(a, width) == 0 || !basis_fast_path(kind, width, taps))) return hipErrorInvalidValue;
    if (!fast_geometry_ok(a, width) || band_rows(a, width) == 0) return launch_generic(kind, width, taps, a, scratch, s);
    if (kind == 2 && width == BankG2::W) {
        Folded<BankG2> f;
        if (fold_taps<BankG2>(taps, f)) return for_each_band(a, width, [&](const BasisArgs& b) { return launch_fast<BankG2>(b, f, s); });
    }
    if (kind == 4 && width == BankG4G::W) {
        Folded<BankG4G> fg;
        Folded<BankG4H> fh;
        if (fold_taps<BankG4G>(taps, fg) && fold_taps<BankG4H>(taps, fh))
            return for_each_band(a, width, [&](const BasisArgs& b) { return launch_pair<BankG4G, BankG4H>(b, fg, fh, s); });
    }
    return launch_generic(kind, width, taps, a, scratch, s);
}

}  // namespace cvs

#ifdef CVS_DIAG_CANARY
// canary twin only (never in the product library, not declared in the public header): read -- and reset -- the counters
extern "C" int cvs_diag_canary(unsigned long long out[4], int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(cvs::g_canary), 4 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(cvs::g_canary), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
