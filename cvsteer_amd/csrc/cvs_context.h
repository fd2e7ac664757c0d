// cvs_context.h -- the handle behind the C ABI and the host-side helpers its translation units share.
// Internal: cvs_api.cpp (entry points), cvs_handle.cpp (argument checks, staging arena, state blocks), cvs_tune.cpp (launch
// configuration), cvs_host.cpp (overlapped host path).  The public boundary is include/cvsteer_hip.h.
#pragma once
#include <hip/hip_runtime_api.h>

#include <initializer_list>
#include <string>
#include <vector>

#include "cvs_internal.h"
#include "cvsteer_hip.h"

struct cvs_context {
    static constexpr int kMaxBasis = cvs::kMaxBasis, kMaxTaps = cvs::kMaxTaps;
    using StateBlock = cvs::StateBlock;
    using BatchFrame = cvs::BatchFrame;
    int kind = 0, width = 0, nb = 0, device = 0;
    float spacing = 0.f;
    hipStream_t stream = nullptr;
    float taps[kMaxBasis][kMaxTaps];
    // state planes: nb basis, then c1,c2,c3,theta,strength
    int rows = 0, cols = 0;
    // geometry of the state: groups of planes -- G2: 7 basis | c1,c2,c3,theta,strength; G4: g4a..e | h4a..f | the five
    // orientation planes -- each group planar or row-interleaved (ensure_state); off / pitch / stride in elements
    struct PlaneGroup {
        int first = 0, count = 0;    // state plane indices [first, first + count)
        size_t off = 0;              // first plane of the group, from the frame's base
        size_t pitch = 0;            // row pitch of the group's planes
        size_t stride = 0;           // plane to plane
    };
    PlaneGroup grp[3];
    int ngrp = 0;
    size_t dense_pitch = 0;                  // round_up(cols, 64): the length of one row of one plane
    size_t layout_stride = 0;                // planar form: plane to plane (the block's plane stride)
    float* state = nullptr;      // = sb.base
    size_t state_elems = 0;      // = sb.elems
    StateBlock sb;               // owner of the state memory (cvs_state.cpp)
    bool have_basis = false, have_orient = false;
    // batched state: num_frames blocks of (nb+5) planes; cur_frame selects the block all state
    // accessors and steer calls address
    int num_frames = 1, cur_frame = 0;
    size_t frame_stride = 0;
    BatchFrame* frame_tab = nullptr;
    int frame_tab_cap = 0;
    // staging arena for host planes and scratch (bump allocated per call)
    float* arena = nullptr;
    size_t arena_elems = 0, arena_used = 0;
    float* point_out = nullptr;
    unsigned long long* diag = nullptr;  // diagnostic builds only
    const void* last_image = nullptr;    // input pointer of the previous setup (fresh-input heuristic)
    int layout = 1;   // CVS_OPT_STATE_LAYOUT: 0 = planar, 1 = row-interleaved (default), 2 = one group of twelve for full G2 setups
    int atan_mode = 0, strip_rows = 0, find_on = 0, block_order = -1, persist = 1, g4_ext = 0, autotune = 1;
    cvs_launch_info last{};   // what the last basis launch of this handle did (cvs_get_launch_info)
    int pyr_strip = 1;   // cvs_pyr_down as a strip march (CVS_OPTS pyr_strip=0: the stand-alone kernel; A/B only, same values)
    hipEvent_t ev_order = nullptr;            // cvs_set_stream: orders the new stream behind the old one
    // overlapped host path (host_pipeline): copy streams and per-band events, created on first use
    hipStream_t s_up = nullptr, s_down = nullptr;
    std::vector<hipEvent_t> band_ev;
    int host_overlap = 1;                     // CVS_OPT_HOST_OVERLAP
    bool used = false;                        // any work queued on `stream` so far
    std::string err;
};

// rows that are dense on both sides travel as ONE linear copy: over the host link a pitched 2-D copy of the same bytes
// is served row by row and reaches a fraction of the rate (tools/d2h_probe.hip, tools/bytes_probe.py)
static inline hipError_t copy_rows(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows,
                                   hipMemcpyKind kind, hipStream_t stream)
{
    if (dpitch == width && spitch == width) return hipMemcpyAsync(dst, src, width * rows, kind, stream);
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, kind, stream);
}

namespace cvs {

int fail(cvs_handle h, int code, const char* what);
int fail_hip(cvs_handle h, hipError_t e, const char* where);

#define HIP_TRY(h, expr)                                        \
    do {                                                        \
        hipError_t e__ = (expr);                                \
        if (e__ != hipSuccess) return fail_hip(h, e__, #expr);  \
    } while (0)

// Event queries / synchronisations and allocations are "potentially unsafe" calls while ANY stream of the process is being
// captured in the global capture mode (what torch.cuda.graph uses): they would invalidate that capture.  The engine makes
// such calls only on objects that belong to no capture; this guard says so for the calling thread (as PyTorch's caching
// allocator does around its own cudaMalloc).
struct RelaxedCapture {
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    RelaxedCapture() { (void)hipThreadExchangeStreamCaptureMode(&mode); }
    ~RelaxedCapture() { (void)hipThreadExchangeStreamCaptureMode(&mode); }
    RelaxedCapture(const RelaxedCapture&) = delete;
    RelaxedCapture& operator=(const RelaxedCapture&) = delete;
};

inline size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }
inline bool is_u8(const cvs_plane* p) { return (p->mem & CVS_DEPTH_U8) != 0; }
inline int mem_of(const cvs_plane* p) { return p->mem & 0xff; }

// ---- staging arena: device copies of host planes for the duration of one call ----
struct Pending {
    const cvs_plane* host;
    float* dev;
    size_t pitch;
};

struct Call {
    cvs_handle h;
    std::vector<Pending> outs;
    bool touched_host = false;
    size_t need = 0;
    // overlapped host path: in_ref only reserves the device copy of a HOST image, host_pipeline moves the bytes
    bool defer = false;
    const cvs_plane* deferred_image = nullptr;
    uint8_t* deferred_u8 = nullptr;   // device staging of an 8-bit host image
    size_t deferred_u8_pitch = 0;
    // 8-bit image read by the strip kernel itself (buffer_load_ubyte, widened in registers): no f32 copy of the image is made
    bool u8_direct = false;
};

// ---- cvs_handle.cpp ----
int check_plane(cvs_handle h, const cvs_plane* p, const char* name, bool allow_u8 = false);
bool planes_overlap(const cvs_plane* a, const cvs_plane* b);
int check_no_overlap(cvs_handle h, const cvs_plane* input, const cvs_plane* const* outs, int n);
int check_point_overlaps(cvs_handle h, std::initializer_list<const cvs_plane*> ins, std::initializer_list<const cvs_plane*> outs);
int check_same(cvs_handle h, const cvs_plane* p, int rows, int cols);
int arena_reserve(cvs_handle h, size_t elems);
float* arena_take(cvs_handle h, size_t elems);
size_t u8_stage_elems(const cvs_plane* p);
size_t staged_elems(const cvs_plane* p);
int in_ref(Call& c, const cvs_plane* p, PlaneRef& r);
int out_ref(Call& c, const cvs_plane* p, PlaneRef& r);
int finish(Call& c);
int begin(cvs_handle h, Call& c, std::initializer_list<const cvs_plane*> planes, size_t extra = 0);
float* state_plane(cvs_handle h, int idx);
PlaneRef state_ref(cvs_handle h, int idx);
const cvs_context::PlaneGroup& state_group(cvs_handle h, int idx);
void fill_state_args(cvs_handle h, BasisArgs& a, bool orient);   // basis / basis2 / orient pointers, pitches, strides, state_bytes of the current frame
void pool_give(StateBlock& blk);
void pool_release_all();
void release_state(cvs_handle h);
bool state_interleaved(cvs_handle h, int rows, size_t dense_pitch);
int ensure_state(cvs_handle h, int rows, int cols, int nframes = 1);
bool state_merge_fits(cvs_handle h, int rows, size_t dense_pitch);
void layout_state(cvs_handle h, bool merge_orient);   // (re)lays the planes out inside the block; ensure_state leaves the two-group form

// ---- cvs_tune.cpp ----
int default_strip_rows(cvs_handle h, int rows, int cols, bool fresh_input = false);
int use_nt_stores(cvs_handle h, size_t npix);
// process-wide overrides parsed once from the environment variable CVS_OPTS="name=value,..." (include/cvsteer_hip.h)
struct EnvOpts {
    int autotune = -1, layout = -1, pyr_strip = -1, batch_ways = -1, nt_stores = -1, warm = -1, wgcap = -1, lit = -1, verbose = 0;
    long pool_mb = 4096;
};
EnvOpts env_opts();
// the configuration of the launch about to be queued (defaults, or the candidate whose turn it is while the shape is being
// compared on the caller's own launches); tune_end goes right behind the launch
struct TuneToken {
    void* entry = nullptr;
    int cand = 0, calls = 1;
    double npix = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};
int tune_begin(cvs_handle h, BasisArgs& a, int variant, bool fresh_input, TuneToken& tok);
void tune_end(cvs_handle h, const TuneToken& tok);
void note_launch(cvs_handle h, const BasisArgs& a);

// ---- cvs_host.cpp ----
int host_pipeline(cvs_handle h, Call& c, BasisArgs& a, float* scr);

}  // namespace cvs
