// cvs_internal.h -- launch descriptors shared between the C-ABI layer (cvs_api.cpp) and the
// kernel translation units.  Not part of the public boundary (that is include/cvsteer_hip.h).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace cvs {

constexpr int kMaxWidth = 32;           // generic path: up to 65 taps
constexpr int kMaxTaps = 2 * kMaxWidth + 1;
constexpr int kMaxBasis = 11;
constexpr int kOrderXcdColumns = 1000000;  // BasisArgs::block_order: every XCD owns a contiguous range of column blocks
constexpr int kOrderDynamic = 2000000;     // BasisArgs::block_order: the tail of the launch is taken from per-XCD queues (tile_ctr)

struct PlaneRef {
    float* p;       // nullptr = not requested
    size_t pitch;   // elements
};

// per-frame pointers of a batched launch (device-resident table, one entry per frame = blockIdx.z)
struct BatchFrame {
    const float* in;
    size_t in_pitch;
    PlaneRef out[8];   // g2,h2,e,mag,phase,edges,dark,bright (any may be {nullptr,0})
};

// One launch of the fused basis kernel ("K1").  All pitches/strides are in ELEMENTS.
struct BasisArgs {
    const float* in;      // image, device (in_u8: the pointer really is a const uint8_t*)
    size_t in_pitch;      // elements of the image's own type (floats, or bytes with in_u8)
    int in_u8;            // 1 = the image is 8-bit: the strip kernels read bytes and widen them in registers (buffer_load_ubyte +
                          // v_cvt_f32_ubyte0), exactly cv::Mat1f(const Mat&) of test/test.cpp:85 -- no f32 copy of the image exists
    int rows, cols;
    float* basis;         // first group of basis planes (G2: all 7; G4: g4a..g4e), plane p at basis + p*plane_stride
    size_t pitch;         // row pitch of the planes of that group
    size_t plane_stride;
    float* basis2;        // G4: second group of basis planes (h4a..h4f), plane 5+q at basis2 + q*plane_stride2, row pitch pitch2.
    size_t pitch2;        // The half banks of the pair launch write one group each; with row-interleaved groups each of them
    size_t plane_stride2; // streams a dense linear sweep (cvs_handle.cpp ensure_state).  G2: unused.
    float* orient;        // c1,c2,c3,theta,strength at orient + i*orient_stride (row pitch orient_pitch), or nullptr
    size_t orient_pitch;  // the state is laid out as two groups of planes, basis and orientation, each either planar
    size_t orient_stride; // (pitch = row length, stride = plane size) or ROW-INTERLEAVED (stride = row length, pitch = planes x row
                          // length: row r of all planes of the group lies side by side) -- see cvs_handle.cpp layout_state
    size_t state_bytes;   // bytes of one frame's state block from `basis` on (basis group, then orientation group)
    float* steer_g;       // fused scalar-steer outputs, or nullptr
    size_t steer_g_pitch;
    float* steer_h;
    size_t steer_h_pitch;
    float steer_w[kMaxBasis];  // scalar steering weights (host-computed)
    int strip_rows;       // output rows per wave strip
    int wg_per_cu;        // host only: at most this many workgroups per CU (0 = whatever registers and LDS allow); the launcher turns it into
                          // dynamic LDS no kernel touches (cvs_tune.cpp default_config says where and why)
    int atan_mode;
    int nt_stores;        // 1 = nontemporal (streaming) output stores
    int warm_k;           // new images (set by the API layer; 0 = off): the waves of the launch's first row bands also touch the rows of warm_k
                          // bands further down each, so that the rest of the image is requested while the launch is young (dma_warm)
    int warm_bands;       // ... how many bands do that: ceil(bands / (warm_k + 1)); filled by the launcher
    int merge_orient;     // host only (cvs_tune.cpp -> do_setup): lay the G2 orientation planes out in one group with the basis planes
    int row_lo, row_hi, row_base;  // set by launch_basis: output rows of this launch / row the plane pointers start at
    int out_row_lo, out_row_hi;    // caller: compute output rows [out_row_lo, out_row_hi) only (0, 0 = the whole image);
                                   // the band-split of one large image over several GPUs (cvs_setup_rows)
    int lit_taps;         // 1 = the launcher may use the instances with the reference's default taps compiled in (it still holds the handle's taps
                          // against the table, bit for bit); 0: CVS_OPTS lit=0
    int* lit_used;        // host: the launcher notes here whether the launch ran such an instance (cvs_launch_info.literal_taps); may be nullptr
    int block_order;      // 0 = row-major grid (default), kOrderXcdColumns, kOrderDynamic
    int grid_x, grid_y;   // filled by the launcher
    int dyn_static;       // dynamic order: the first dyn_static tiles are dealt statically (tile = workgroup index); filled by the launcher
    int dyn_nz;           // dynamic order: planes of tiles (frames of a batch / half banks of the G4 pair); filled by the launcher
    unsigned* tile_ctr;   // dynamic order: the handle's tile queues in device memory -- two sets, used alternately (cvs_kernels_basis.hip);
                          // the API layer passes the slot, the launcher picks the set
    unsigned* tile_ctr_next;  // the other set, zeroed by this launch (filled by the launcher)
    int* tile_parity;     // host: which set the next dynamic launch of this state block takes (StateBlock::ctr_parity)
    // fused caller pipeline (needs orient): g2,h2,e,mag,phase,edges,dark,bright at theta_dom
    int pipe;             // 1 = run the pipeline epilogue
    int no_state;         // pipeline only: 1 = do not persist basis / orientation planes (outputs only)
    int find_on_e;        // 1 = find*(e, phase), 0 = find*(magnitude, phase)
    PlaneRef pipe_out[8]; // any entry may be {nullptr, 0}
    // batched launch: n frames of identical geometry, frame z takes in/out from frames[z] and its
    // state planes at basis + z*frame_stride (orient likewise)
    const BatchFrame* frames;  // device pointer, or nullptr = single image / regular batch
    int batch;
    int z_ways;           // frames are dispatched dealt from z_ways equal parts of the batch in turn (1 = in order)
    // regular batch (frames == nullptr, batch_regular = 1): frame z reads in + z*in_frame_stride and writes
    // pipe_out[k].p + z*out_frame_stride -- pointers from kernel arguments only, no table, no upload
    int batch_regular;
    size_t in_frame_stride, out_frame_stride;  // elements
    // regular batch whose outputs of one frame share a pitch and lie within 2 GiB of each other (one [K, H, W] block per
    // frame, the usual case): ONE buffer resource per frame for all of them -- out_base + z*out_frame_stride, out_bytes
    // long -- and output k at byte offset out_off[k]; out_mask = which outputs exist.  Keeps 8 pointers + 8 pitches out
    // of the scalar registers (the batched pipeline kernel spilled ~290 SGPR lane moves per row without this).
    int out_one;
    unsigned out_mask;
    unsigned out_off[8];
    float* out_base;
    size_t out_pitch;   // elements
    size_t out_bytes;
    size_t frame_stride;       // elements between the state blocks of consecutive frames
    // fused pyramid level (BASELINE config 3): the same launch also writes cv::pyrDown(image) -- ((rows+1)/2) x ((cols+1)/2),
    // 5-tap [1 4 6 4 1]/16 blur + 2x decimation from the rows the wave has staged anyway -- so that the image is read once
    // for "filter this level and make the next one".  nullptr = off.  G2 bank, un-banded single-image launches only; the
    // API layer runs the stand-alone k_pyr_down otherwise (identical values).
    float* pyr_out;
    size_t pyr_pitch;          // elements
    // diagnostic builds only (-DCVS_DIAG_STAMPS, tools/k1_timeline.py): per-wave {start, first store, end}
    // 100 MHz real-time stamps; never read by the product, nullptr in normal builds
    unsigned long long* diag;
};

// basis plane p of a launch, whatever group it is in
inline PlaneRef basis_plane_ref(const BasisArgs& a, int kind, int p)
{
    if (kind == 4 && p >= 5) return {a.basis2 + (size_t)(p - 5) * a.plane_stride2, a.pitch2};
    return {a.basis + (size_t)p * a.plane_stride, a.pitch};
}

// taps[i] = the handle's i-th tap vector (member order), 2*width+1 floats each
hipError_t launch_basis(int kind, int width, const float (*taps)[kMaxTaps], const BasisArgs& a,
                        float* scratch, hipStream_t s);
// elements of scratch the generic-width path needs for this image (0 on the fast paths)
size_t basis_scratch_elems(int kind, int width, int rows, size_t pitch);
bool basis_fast_path(int kind, int width, const float (*taps)[kMaxTaps]);
bool basis_may_need_scratch(int kind, int width, const float (*taps)[kMaxTaps], int rows, int cols, size_t max_pitch);
// true when a.pyr_out (the next pyramid level) is written by the filter launch itself; otherwise launch_basis adds a
// k_pyr_down launch behind it (same values either way)
bool basis_fuses_pyr(int kind, int width, const float (*taps)[kMaxTaps], const BasisArgs& a);

// ---- pointwise kernels ("K2..K5") ----
enum PointOp {
    OP_G2_ORIENT = 0,      // in: 7 basis                     out: c1,c2,c3,theta,strength
    OP_G2_STEER_SCALAR,    // in: 7 basis [,c1,c2,c3]          out: g,h[,e,mag,phase]
    OP_G2_STEER_MAP,       // in: 7 basis, theta [,c1,c2,c3]   out: g,h[,e,mag,phase]
    OP_G4_STEER_SCALAR,    // in: 11 basis [,c1,c2,c3]         out: g,h[,e,mag,phase: extension]
    OP_G4_STEER_MAP,       // in: 11 basis [,c1..c3], theta@14 out: g,h[,e,mag,phase: extension]
    OP_MAG_PHASE,          // in: g,h                          out: mag,phase
    OP_PHASE_WEIGHTS,      // in: phase                        out: lambda
    OP_FIND,               // in: e,phase                      out: edges,dark,bright
    OP_G2_PIPELINE,        // in: 7 basis, theta, c1,c2,c3     out: g,h,e,mag,phase,edges,dark,bright
    OP_WRAP,               // in: angle                        out: wrapped angle
    OP_G4_ORIENT           // in: 11 basis                     out: c1,c2,c3,theta,strength (extension)
};

constexpr int kMaxIn = 15;
constexpr int kMaxOut = 8;

struct PointArgs {
    int rows, cols;
    PlaneRef in[kMaxIn];
    PlaneRef out[kMaxOut];
    float w[kMaxBasis];   // scalar steering weights
    float c2t, s2t;       // cos/sin(2 theta) for the scalar-theta energy
    float phi;            // phaseWeights
    int signum;
    int atan_mode;
    int find_on_e;        // pipeline: 1 = find*(e, phase), 0 = find*(magnitude, phase)
    int nt_stores;        // 1 = nontemporal (streaming) output stores
    int nt_loads;         // 1 = nontemporal input loads (steer stages on the state planes of a large image)
};

hipError_t launch_point(PointOp op, const PointArgs& a, hipStream_t s);
// single-pixel steer (G2.cpp:115-134): uses a.w, a.c2t, a.s2t; writes {g2,h2,e,mag,phase} to out5 (device)
hipError_t launch_steer_point(const float* basis, size_t plane_stride, size_t offset, const float* orient, size_t orient_stride,
                              size_t orient_offset, const PointArgs& a, float* out5, hipStream_t s);   // orient may be nullptr

// per-image min/max + 8-bit quantise (cv::normalize NORM_MINMAX -> CV_8UC1)
hipError_t launch_minmax(const float* src, size_t pitch, int rows, int cols, float* minmax2, hipStream_t s);
// n equally sized planes at a constant stride, one launch: min / max per plane (minmax = true, minmax2n = 2n floats of
// scratch) or convertTo(alpha, beta), 8-bit results at dst + z * dst_plane_stride
hipError_t launch_to_u8_n(const float* src, size_t plane_stride, size_t pitch, int rows, int cols, int n, bool minmax, float* minmax2n,
                          float alpha, float beta, uint8_t* dst, size_t dst_plane_stride, size_t dst_step, hipStream_t s);
hipError_t launch_quantize_u8(const float* src, size_t pitch, int rows, int cols, const float* minmax2,
                              uint8_t* dst, size_t dst_step, hipStream_t s);

hipError_t launch_convert_u8(const float* src, size_t pitch, int rows, int cols, float alpha, float beta, uint8_t* dst,
                             size_t dst_step, hipStream_t s);
hipError_t launch_u8_to_f32(const uint8_t* src, size_t sstep, int rows, int cols, float* dst, size_t dpitch, hipStream_t s);
hipError_t launch_pyr_down(const float* src, size_t spitch, int rows, int cols, float* dst, size_t dpitch, hipStream_t s);
// the same as a strip march of the basis kernel's machinery (cvs_kernels_basis.hip); false = geometry not covered, use launch_pyr_down
bool launch_pyr_strip(const float* src, size_t spitch, int rows, int cols, float* dst, size_t dpitch, hipStream_t s, hipError_t* err);

// ---- state blocks (cvs_state.cpp): one plain hipMalloc block per handle, parked in a process-wide cache between handles ----
struct StateBlock {
    float* base = nullptr;
    size_t elems = 0;          // usable floats from base
    int device = 0;
    // a PARKED block (cvs_destroy / a handle that changed geometry): recorded on the stream that last used the block; whoever
    // takes the block over makes its own stream wait for it -- the device is never drained for a destroy.  nullptr = idle.
    hipEvent_t ready = nullptr;
    // tile queues of the dynamic launch order (two sets of 8 counters + a flag, 1 KiB each): a slot of a per-device slab, taken
    // with the block and returned with it -- so that whoever uses the block next has waited for `ready` first
    unsigned* tile_ctr = nullptr;
    int ctr_parity = 0;        // bit 0: the set the next dynamic launch uses (that launch zeroes the other one); bit 1: the slot has served
                               // another block -- the next dynamic launch resets its set with a kernel first (cvs_state.cpp tile_ctr_alloc)
};
hipError_t state_block_alloc(int device, size_t elems, StateBlock& b);
void state_block_free(StateBlock& b);

// host-side tap math (cvs_taps.cpp, no HIP)
int host_num_basis(int kind);
int host_make_taps(int kind, int idx, int width, float spacing, float* out);
int host_basis_taps(int kind, int p, int* kx, int* ky);
int host_steer_weights(int kind, float theta, float* out);

}  // namespace cvs
