/*
 * cvsteer_hip.h -- C ABI of libcvsteer_hip.so: the MI355X (gfx950) engine behind the
 * fa::SteerableFilters / SteerableFiltersG2 / SteerableFiltersG4 classes of
 * headupinclouds/cvsteer.
 *
 * The reference has no FFI layer: its boundary is the C++ class surface
 * (cvsteer/SteerableFilters.h:41-50, SteerableFiltersG2.h:35-67, SteerableFiltersG4.h:35-57)
 * and every arithmetic step is an OpenCV call.  Each entry point below names the reference
 * member function (file:line) whose work it replaces.  The C++ facade in
 * the include/cvsteer/ headers keep the reference's class/method names on top of this ABI;
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *  - every function returns an int status (CVS_OK = 0, negative = error); nothing throws.
 *  - images are `cvs_plane`: row-major f32, `step` bytes between rows (multiple of 4,
 *    >= cols*4), living in host memory (CVS_MEM_HOST) or in the handle's device
 *    (CVS_MEM_DEVICE).  cv::Mat1f maps 1:1: {(float*)m.data, m.rows, m.cols, m.step}.
 *  - a handle owns its state planes (basis, C1..C3, theta, strength) in device memory;
 *    callers own every plane they pass in.  State stays valid until the next cvs_setup.
 *  - all device work is enqueued on the handle's HIP stream (default: the null stream;
 *    cvs_set_stream to share e.g. PyTorch's current stream).  Calls with only DEVICE planes
 *    are asynchronous; calls that touch a HOST plane return after the data has landed.
 *  - a handle is not re-entrant; different handles may be used from different threads.
 *  - there is no CPU fallback: without a usable HIP device cvs_create fails with CVS_E_HIP.
 */
#ifndef CVSTEER_HIP_H
#define CVSTEER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): cvs_launch_info carries its size; the placement-search, XCD-weight, store-policy, G4-split and
 * workgroups-per-CU options are gone (9 options left) */
#define CVS_ABI_VERSION 2

/* status codes */
enum {
    CVS_OK = 0,
    CVS_E_BADARG = -1,      /* null pointer, unknown enum, bad index */
    CVS_E_SIZE = -2,        /* empty image, mismatched plane sizes, bad step */
    CVS_E_HIP = -3,         /* HIP runtime error (see cvs_last_error) */
    CVS_E_NOMEM = -4,       /* device or host allocation failed */
    CVS_E_STATE = -5,       /* state not available (no setup yet / orientation not computed) */
    CVS_E_UNSUPPORTED = -6  /* operation the reference does not define for this kind */
};

/* Planes that share memory.  The reference never checks (an output that overlaps its input gives whatever OpenCV's loops
 * happen to produce).  Here it is CVS_E_BADARG, for host planes and device planes alike:
 *   - filter-bank entries (cvs_setup*, cvs_pipeline*, cvs_pyr_down): no output may share a byte with the input image (the
 *     kernels read rows ahead of the rows they write) or with another output;
 *   - per-pixel entries (cvs_steer_*, cvs_mag_phase, cvs_phase_weights, cvs_find, cvs_wrap): an output may BE an input -- same
 *     first pixel, same step; the reference itself calls wrap(m_theta, m_theta) -- but may not overlap one in any other way,
 *     nor another output.
 * Two views with the same step are compared exactly (column ranges of one buffer side by side share nothing and are fine);
 * views with different steps by their address ranges.  f32 planes must be 4-byte aligned. */
enum { CVS_KIND_G2 = 2, CVS_KIND_G4 = 4 };
enum { CVS_MEM_HOST = 0, CVS_MEM_DEVICE = 1 };
/* OR-ed into cvs_plane.mem of an INPUT IMAGE (cvs_setup / cvs_setup_steer / cvs_pipeline[_batch]): `data` points
 * at 8-bit samples (`step` >= cols bytes).  The reference's callers hand 8-bit images to the constructor and let
 * cv::Mat1f(const Mat&) widen them unscaled (test/test.cpp:73,85; example/steer.cpp:73-86); here the 8-bit data
 * crosses PCIe as bytes and is widened on the device. */
enum { CVS_DEPTH_U8 = 0x100 };

/* cvs_setup flags */
enum {
    CVS_SETUP_BASIS = 1u,   /* the 7 (G2) / 11 (G4) separable basis planes */
    CVS_SETUP_ORIENT = 2u,  /* + C1,C2,C3, dominant angle, strength (G2 only) */
    CVS_SETUP_FULL = 3u
};

/* options for cvs_set_option.  Process-wide overrides for new handles (A/B aids): the environment variable
 * CVS_OPTS="name=value,..." with autotune=0|1, layout=0|1|2|3, pyr_strip=0|1, batch_ways=N, warm=K (0 = off), wgcap=N (workgroups per CU, 0 = no cap),
 * lit=0|1 (0: never the kernel instances with the default taps compiled in),
 * nt_stores=0|1 (output stores plain / nontemporal instead of by size), verbose=1 (the tuner prints its decisions to stderr),
 * pool_mb=N (state-block cache, default 4096, 0 = off).  Read at every call; results never depend on any of them. */
enum {
    CVS_OPT_ATAN_MODE = 1,   /* 0 = OpenCV-compatible fastAtan2 polynomial (default), 1 = exact atan2f */
    CVS_OPT_STRIP_ROWS = 2,  /* rows per wave strip of the basis kernel (0 = default: chosen by the engine / its tuner) */
    CVS_OPT_FIND_ON = 3,     /* cvs_pipeline: 0 = find*(magnitude, phase) as the reference's callers do
                                (test/test.cpp:88-90), 1 = find*(e, phase) */
    CVS_OPT_G4_EXTENSIONS = 6, /* 0 (default) = G4 exactly as the reference leaves it (no orientation, no e/mag/phase);
                                  1 = EXTENSION beyond the reference: cvs_setup(G4, CVS_SETUP_FULL) fills C1..C3 / theta /
                                  strength from the G4/H4 steering polynomials, and cvs_steer_* accept e/mag/phase */
    CVS_OPT_BLOCK_ORDER = 8, /* order in which the basis kernel's workgroups take their tiles: -1 (default) = the engine's choice
                                (row-major, unless its tuner finds one of the others faster for this shape on the caller's own
                                launches); 0 = row-major; 1000000 = every XCD walks its own range of column blocks;
                                2000000 = row-major with a DYNAMIC TAIL: the last tenth of the tiles is handed out on demand
                                from eight per-XCD queues in device memory (grid = static tiles + 1.25 x the tail tiles), so an
                                XCD that is ahead takes over tiles of the others.  Results do not depend on it. */
    CVS_OPT_PERSIST_STATE = 9, /* cvs_pipeline / cvs_pipeline_batch: 1 (default) = keep basis + orientation planes like the
                                  reference object does; 0 = write the requested outputs only (no state afterwards) */
    CVS_OPT_AUTOTUNE = 12,   /* 1 (default): while a (kind, entry point, shape bucket) is undecided, each call runs one of a few launch
                                configurations, every candidate in sustained turns of 20-100 consecutive calls timed as a whole between
                                two events on the caller's stream (no extra launches, no waiting); the engine keeps a challenger only
                                when its turns are 3 % ahead of the default's, consistently; 0 = always the
                                defaults */
    CVS_OPT_HOST_OVERLAP = 13, /* cvs_setup / cvs_setup_steer / cvs_pipeline with HOST planes on images of 1 Mpix and more:
                                  1 (default) = the image goes up, is filtered and comes down in row bands, all three at once
                                  (full-duplex host link, a second host thread for the downloads); 0 = one after the other */
    CVS_OPT_STATE_LAYOUT = 14  /* how the handle's state planes lie in its block.  1 (default) = ROW-INTERLEAVED: row r of all basis
                                  planes side by side ([row][plane][column]; the five orientation planes likewise, in a group of
                                  their own) -- a launch that writes 7..12 planes then streams ONE linear sweep per group instead of
                                  one stream per plane 64 MiB apart; every plane is still an ordinary strided
                                  image (cvs_state_plane: step = planes x row length).  0 = planar, plane after plane (also what
                                  groups of 2 GiB and more use).  With 1 the engine puts all twelve G2 planes into ONE group for
                                  single-image launches that write the orientation planes too (full setup, caller pipeline: one
                                  write sweep instead of two; cvs_launch_info.state_layout = 2 then) and uses the two groups for
                                  every other launch; 2 = the same, spelled out; 3 = always two groups.
                                  Takes effect at the next cvs_setup*; results do not depend on it.
                                  NOTE: because the grouping may change from one setup to the next, a cvs_state_plane view is
                                  valid only until the handle's next cvs_setup* / cvs_pipeline* call. */
};

/* state planes addressable through cvs_state_plane / cvs_read_state */
enum {
    CVS_PLANE_BASIS0 = 0,    /* + p, p < cvs_num_basis(kind): m_g2a..m_h2d / m_g4a..m_h4f */
    CVS_PLANE_C1 = 32, CVS_PLANE_C2 = 33, CVS_PLANE_C3 = 34,
    CVS_PLANE_THETA = 35,    /* getDominantOrientationAngle()    */
    CVS_PLANE_STRENGTH = 36  /* getDominantOrientationStrength() */
};

typedef struct cvs_plane {
    float* data;
    int32_t rows;
    int32_t cols;
    size_t step;   /* bytes */
    int32_t mem;   /* CVS_MEM_HOST | CVS_MEM_DEVICE */
} cvs_plane;

typedef struct cvs_context* cvs_handle;

/* ---------------- library / host-only helpers (no GPU needed) ---------------- */
int cvs_abi_version(void);
const char* cvs_status_string(int status);
/* number of 1-D tap vectors == number of basis planes: 7 (G2), 11 (G4) */
int cvs_num_basis(int kind);
/* SteerableFilters::create (SteerableFilters.cpp:33-42) applied to the idx-th tap function
 * (SteerableFiltersG2.cpp:35-42 order m_g1,m_g2,m_g3,m_h1..m_h4; SteerableFiltersG4.cpp:34-45
 * order m_g1..m_g5,m_h1..m_h6).  out: 2*width+1 floats. */
int cvs_make_taps(int kind, int idx, int width, float spacing, float* out);
/* which taps build basis plane p: sepFilter2D(image, kx=taps[*kx], ky=taps[*ky])
 * (SteerableFiltersG2.cpp:62-68, SteerableFiltersG4.cpp:69-80) */
int cvs_basis_taps(int kind, int p, int* kx, int* ky);
/* scalar steering weights for theta (SteerableFiltersG2.cpp:140-142, G4.cpp:116-119):
 * out[0..2]=ga,gb,gc out[3..6]=ha..hd (G2) ; out[0..4]=ga..ge out[5..10]=ha..hf (G4) */
int cvs_steer_weights(int kind, float theta, float* out);

/* ---------------- handle ---------------- */
/* SteerableFiltersG2::SteerableFiltersG2 / G4 ctor minus setup (G2.cpp:44-56, G4.cpp:47-63):
 * builds the tap vectors, binds HIP device `device`. */
int cvs_create(int kind, int width, float spacing, int device, cvs_handle* out);
/* ~SteerableFiltersG2/G4.  The handle's state block (its largest allocation) is not freed but parked in a
 * process-wide cache -- the reference's callers build one short-lived object per image (example/steer.cpp:86,
 * test/test.cpp:85), and the next handle on the same device takes the block over instead of allocating.
 * The cache is bounded (CVS_OPTS pool_mb, default 4096; 0 = off). */
int cvs_destroy(cvs_handle h);
/* frees every block held by that cache */
int cvs_release_cached_memory(void);
const char* cvs_last_error(cvs_handle h);
/* Bind the handle to a HIP stream.  The handle's buffers are reused from call to call, so when the stream changes
 * the new stream is made to wait (event, no host sync) for the work already queued on the old one. */
int cvs_set_stream(cvs_handle h, void* hip_stream);
int cvs_set_option(cvs_handle h, int option, int value);
int cvs_get_option(cvs_handle h, int option, int* value);
/* How the handle's last basis launch was configured (launch order, strip height, store policy, state layout: the defaults
 * or what the online tuner kept).  For benchmarks and tests; nothing of this changes results.  The caller sets struct_size =
 * sizeof(cvs_launch_info) before the call; the library fills at most that many bytes (a client built against an older, shorter
 * struct keeps working). */
typedef struct cvs_launch_info {
    uint32_t struct_size;     /* in: sizeof(cvs_launch_info) of the caller */
    int32_t block_order;      /* last basis launch: CVS_OPT_BLOCK_ORDER value in effect */
    int32_t strip_rows;       /* ... output rows per wave strip */
    int32_t nt_stores;        /* ... 1 = streaming (nontemporal) stores */
    int32_t state_layout;     /* layout of the current state block: 0 = planar, 1 = row-interleaved groups (CVS_OPT_STATE_LAYOUT),
                                 2 = row-interleaved with the G2 orientation planes in the basis planes' group (what launches that write them use) */
    int32_t warm;             /* last basis launch: K > 0 = the launch took its image for a NEW one (another pointer than the handle's
                                 previous call) and the waves of its first row bands requested the rest of the image ahead of need,
                                 K bands each (f32 images of 24 MiB and more; not for 8-bit images, frame batches and launches that
                                 also emit a pyramid level); 0 = not */
    int32_t tuning_launches;  /* launches the engine has issued on this handle's stream beyond the caller's own calls: always 0
                                 (configurations are compared on the caller's launches) */
    int32_t tuned;            /* 1 = the configuration above is a challenger the online tuner decided for; 0 = the engine's default */
    int32_t tune_state;       /* the online tuner for this launch's key: 0 = off / not a tunable launch, 1 = still comparing on the
                                 caller's launches, 2 = decided */
    int32_t wg_per_cu;        /* last basis launch: workgroups per CU it was held to (single G2 images of 2 Mpix and more: three or four
                                 instead of the six the registers allow -- fewer write fronts); 0 = no cap */
    int32_t literal_taps;     /* last basis launch: 1 = it ran a kernel instance with the reference's default G2 / H2 taps compiled in as literal
                                 operands (the caller-pipeline variants of a handle made with width 4, spacing 0.67f: same values, cheaper
                                 instruction issue); 0 = taps from the kernel arguments (any other handle or launch) */
} cvs_launch_info;
int cvs_get_launch_info(cvs_handle h, cvs_launch_info* out);
/* the handle's idx-th tap vector (m_g1.. members), 2*width+1 floats */
int cvs_taps(cvs_handle h, int idx, float* out);
int cvs_kind(cvs_handle h, int* kind, int* width, float* spacing);
/* size of the image of the last cvs_setup (0,0 before) */
int cvs_shape(cvs_handle h, int* rows, int* cols);
int cvs_sync(cvs_handle h);

/* ---------------- the hot path ---------------- */
/* SteerableFiltersG2::setup (G2.cpp:60-100) / SteerableFiltersG4::setup (G4.cpp:67-81).
 * One fused kernel: the image is read once; the row pass, the column pass and (with
 * CVS_SETUP_ORIENT) the C1..C3 / cartToPolar / wrap / *0.5 steps run in registers. */
int cvs_setup(cvs_handle h, const cvs_plane* image, unsigned flags);

/* setup + steer(float theta, g, hq) in the same kernel launch: G2.cpp:60-100 followed by
 * G2.cpp:137-145 (G4.cpp:67-81 + G4.cpp:114-122).  This is the headline "filter+steer" unit. */
int cvs_setup_steer(cvs_handle h, const cvs_plane* image, unsigned flags, float theta,
                    const cvs_plane* g, const cvs_plane* hq);

/* cvs_setup restricted to the output rows [row_lo, row_hi) of the image: the band one GPU takes when a single large
 * image (a pyramid level, BASELINE config 3) is split over several GPUs.  The whole image must be
 * present (the rows around the band are read, the image borders reflect as usual); state rows outside the band keep
 * whatever they held.  Values inside the band are bit-identical to those of a whole-image cvs_setup.  Images that
 * take the generic path (non-default taps, tiny images) are filtered whole. */
int cvs_setup_rows(cvs_handle h, const cvs_plane* image, unsigned flags, int row_lo, int row_hi);

/* device view of a state plane (zero copy; valid until the handle's next cvs_setup* / cvs_pipeline* call: the engine may
 * re-lay the planes between calls, see CVS_OPT_STATE_LAYOUT) */
int cvs_state_plane(cvs_handle h, int which, cvs_plane* view);
/* copy a state plane out (getDominantOrientationAngle()/Strength() getters, G2.h:40-41,
 * and the protected m_g2a.. members for tests) */
int cvs_read_state(cvs_handle h, int which, const cvs_plane* dst);

/* steer(float theta, g2, h2) G2.cpp:137-145 / G4.cpp:114-122; with e, mag, phase non-NULL:
 * steer(float theta, g2, h2, e, magnitude, phase) G2.cpp:157-165.  G4: e/mag/phase must be NULL. */
int cvs_steer_scalar(cvs_handle h, float theta, const cvs_plane* g, const cvs_plane* hq,
                     const cvs_plane* e, const cvs_plane* mag, const cvs_plane* phase);
/* steer(const Mat1f& theta, ...) G2.cpp:147-155, :167-177 / G4.cpp:92-112.
 * theta == NULL steers at the handle's own dominant-orientation plane (what both reference
 * callers do: test/test.cpp:86, example/steer.cpp:87). */
int cvs_steer_map(cvs_handle h, const cvs_plane* theta, const cvs_plane* g, const cvs_plane* hq,
                  const cvs_plane* e, const cvs_plane* mag, const cvs_plane* phase);
/* steer(const cv::Point& p, theta, g2, h2, e, magnitude, phase) G2.cpp:115-134 (p.x=col, p.y=row).
 * out = {g2, h2, e, magnitude, phase}; e is NaN when orientation state is absent. */
int cvs_steer_point(cvs_handle h, int x, int y, float theta, float out[5]);

/* computeMagnitudeAndPhase G2.cpp:107-112 (cartToPolar, wrap, patchNaNs) */
int cvs_mag_phase(cvs_handle h, const cvs_plane* g, const cvs_plane* hq,
                  const cvs_plane* mag, const cvs_plane* phase);
/* SteerableFilters::wrap, SteerableFilters.cpp:46-51: out = angle > pi ? angle - 2pi : angle */
int cvs_wrap(cvs_handle h, const cvs_plane* angle, const cvs_plane* out);
/* static phaseWeights G2.cpp:179-186 (k accepted and ignored, like the reference) */
int cvs_phase_weights(cvs_handle h, const cvs_plane* phase, const cvs_plane* lambda,
                      float phi, int signum, float k);
/* findEdges / findDarkLines / findBrightLines G2.cpp:194-212 in one pass; any output may be NULL */
int cvs_find(cvs_handle h, const cvs_plane* e, const cvs_plane* phase,
             const cvs_plane* edges, const cvs_plane* dark, const cvs_plane* bright);

/* the whole caller sequence of test/test.cpp:85-90 / example/steer.cpp:86-90 for one image:
 * setup(FULL) -> steer(theta_dom, g2,h2,e,mag,phase) -> find*(mag|e, phase).
 * outs[8] = {g2, h2, e, magnitude, phase, edges, dark, bright}; any entry may be NULL. */
int cvs_pipeline(cvs_handle h, const cvs_plane* image, const cvs_plane* const outs[8]);

/* The batch axis (example/steer.cpp:69-124,169: one independent pipeline per file): cvs_pipeline for
 * n images of identical size in ONE kernel launch (grid.z = frame).  outs is a flat array of n*8
 * planes, frame-major, order {g2,h2,e,magnitude,phase,edges,dark,bright}; an entry with data == NULL
 * is not written (outs == NULL: state only).  The state of every frame is kept; cvs_select_frame
 * picks the frame that cvs_state_plane / cvs_read_state / cvs_steer_* address (default 0).
 * Host planes, images too small for the fused kernel or non-default taps are processed frame by
 * frame with the same results. */
int cvs_pipeline_batch(cvs_handle h, const cvs_plane* images, int n, const cvs_plane* outs);
int cvs_select_frame(cvs_handle h, int frame);
int cvs_num_frames(cvs_handle h, int* n);

/* One Gaussian-pyramid level (BASELINE config 3; absent from the reference): cv::pyrDown
 * semantics -- 5-tap [1 4 6 4 1]/16 separable blur, BORDER_REFLECT_101, every second pixel.
 * dst must be ((rows+1)/2) x ((cols+1)/2). */
int cvs_pyr_down(cvs_handle h, const cvs_plane* src, const cvs_plane* dst);
/* cvs_setup(h, image, flags) and cvs_pyr_down(h, image, next_level) in ONE pass over the image ("filter this
 * pyramid level and make the next one", BASELINE config 3): the basis kernel emits the decimated level from
 * the rows it has staged for SteerableFiltersG2::setup (G2.cpp:62-68) anyway, so the image is read once instead
 * of twice.  Values are identical to the two separate calls; G4 handles, non-default widths, host planes and
 * planes of 2 GiB and more take the two launches internally. */
int cvs_setup_pyr(cvs_handle h, const cvs_plane* image, unsigned flags, const cvs_plane* next_level);

/* BASELINE config 3 in one call: `levels` handles (one per pyramid level, same device and stream), the level-0 image, and
 * levels - 1 caller-owned planes that receive the pyramid levels 1 .. levels-1 (sizes as for cvs_pyr_down).  The chain
 * cvs_setup_pyr(hs[0], image, ..), cvs_setup_pyr(hs[1], level 1, ..), ..., cvs_setup(hs[levels-1], last level): every level
 * image is read once (the filter launch of a level writes the next one).  Afterwards handle l holds the state of level l. */
int cvs_pyramid_setup(cvs_handle* hs, int levels, const cvs_plane* image, unsigned flags, const cvs_plane* level_images);

/* per-image min/max (cv::normalize NORM_MINMAX, test.cpp:92-94 / steer.cpp:96-98) and the
 * 8-bit quantise that follows; dst is rows*cols bytes with dst_step bytes per row. */
int cvs_normalize_u8(cvs_handle h, const cvs_plane* src, uint8_t* dst, size_t dst_step, int dst_mem);
/* Mat::convertTo(dst, CV_8UC1, alpha, beta) -- the `--gain` branch of example/steer.cpp:92-97 */
int cvs_convert_u8(cvs_handle h, const cvs_plane* src, float alpha, float beta, uint8_t* dst, size_t dst_step, int dst_mem);
/* the same for n planes in one go (a driver turning a whole block of feature maps into 8-bit files, steer.cpp:92-122 per
 * file): equally sized device planes at a constant stride take one min/max launch, one quantise launch and one
 * synchronisation for all of them; anything else goes plane by plane.  dst[i] receives plane i. */
int cvs_normalize_u8_batch(cvs_handle h, const cvs_plane* src, int n, uint8_t* const* dst, size_t dst_step, int dst_mem);
int cvs_convert_u8_batch(cvs_handle h, const cvs_plane* src, int n, float alpha, float beta, uint8_t* const* dst, size_t dst_step, int dst_mem);

/* ---------------- the batch axis over the GPUs of one node (cvs_batch.cpp) ----------------
 * example/steer.cpp:169 runs cv::parallel_for_(Range(0, N), body): one independent SteerableFiltersG2 pipeline per
 * file (steer.cpp:69-124).  Here frame f of F belongs to rank floor(f * G / F) (contiguous blocks), every rank runs
 * cvs_pipeline_batch on its block -- no collective on the data path -- and RCCL moves data only at the edges.
 * A world is formed either by ONE process driving several devices (cvs_batch_create_local: ncclCommInitAll; a
 * device listed twice = rehearsal on a smaller box, transport = device copies instead of RCCL) or by one process per
 * GPU (cvs_batch_unique_id on one rank, the 128 bytes distributed by the caller, cvs_batch_create_rank everywhere).
 * Every rank of the world calls cvs_batch_run / cvs_batch_pyramid_setup with the same arguments; ranks that do not
 * hold the root pass NULL planes.  Calls return when the root holds the results.
 * In a world of several processes the ranks first AGREE (one 4-int ncclAllReduce, before any data is queued) that
 * every one of them can run the call -- the root's planes are what the call needs, every rank's staging fits, all ranks
 * were given the same geometry -- and otherwise all of them return an error with nothing queued; no rank is left
 * waiting in a receive.  (A failure after that point -- a kernel launch error on one rank -- is not recoverable across
 * processes: destroy the batch.)  An RCCL group that was started is always ended, also on errors. */
typedef struct cvs_batch_context* cvs_batch;
enum { CVS_BATCH_ID_BYTES = 128 };
enum { CVS_BATCH_TRANSPORT_NONE = 0, CVS_BATCH_TRANSPORT_RCCL = 1, CVS_BATCH_TRANSPORT_COPY = 2 };

typedef struct cvs_batch_cfg {
    int32_t rows, cols;          /* size of every frame */
    int32_t n_frames;            /* frames in the batch (all of them, on the root) */
    uint32_t outputs;            /* bit k = pipeline output k is wanted: g2,h2,e,magnitude,phase,edges,dark,bright */
    int32_t root;                /* rank that holds the inputs and receives the outputs */
    int32_t gather;              /* 1 = outputs are gathered on the root; 0 = they stay on the ranks (cvs_batch_local_result) */
    int32_t self_via_transport;  /* tests: the root's own block also travels through send / recv */
} cvs_batch_cfg;

typedef struct cvs_batch_timing {  /* HIP-event times on the ranks' streams, maximum over this process's ranks */
    double scatter_ms, compute_ms, gather_ms;
} cvs_batch_timing;

int cvs_batch_create_local(int kind, int width, float spacing, int ndev, const int* devices, cvs_batch* out);
int cvs_batch_unique_id(void* id128);
int cvs_batch_create_rank(int kind, int width, float spacing, const void* id128, int world, int rank, int device, cvs_batch* out);
int cvs_batch_destroy(cvs_batch b);
const char* cvs_batch_last_error(cvs_batch b);
int cvs_batch_info(cvs_batch b, int* world, int* nlocal, int* transport);
/* cvs_set_option on every engine of the batch (e.g. CVS_OPT_PERSIST_STATE = 0: outputs only) */
int cvs_batch_set_option(cvs_batch b, int option, int value);
/* BASELINE config 4 -- the loop of example/steer.cpp:169 over the node: inputs = n_frames dense f32 device planes on the
 * root, outputs = n_frames * 8 planes on the root, frame-major, order of cvs_pipeline (entries not selected by
 * cfg->outputs are ignored).  scatter (grouped ncclSend/ncclRecv) -> one cvs_pipeline_batch launch per rank ->
 * gather (grouped ncclSend/ncclRecv).  The root's own block is processed in place.
 * HOST planes (what the example holds: cv::Mat, steer.cpp:73-104; rows may be padded): inputs and requested outputs all
 * CVS_MEM_HOST; the inputs may be 8-bit (all of them CVS_MEM_HOST | CVS_DEPTH_U8, step in bytes: a quarter of the upload).  Nothing passes through the root's GPU then -- every rank uploads ITS frames from the caller's planes over
 * its own host link and downloads its outputs the same way, all ranks at once, upload / launch / download overlapped
 * chunk by chunk inside a rank.  Needs every rank in the calling process (cvs_batch_create_local, or a world of 1);
 * CVS_E_UNSUPPORTED otherwise.  timing: scatter = upload, gather = download, compute = the slowest rank's whole span.
 * The requested HOST outputs may be 8-bit planes as well (all of them CVS_MEM_HOST | CVS_DEPTH_U8, step in bytes) -- what the
 * example writes (steer.cpp:92-122): every map is then turned into bytes on the device, normalize(0, 255, NORM_MINMAX,
 * CV_8UC1) per map (steer.cpp:98-104) or convertTo(CV_8UC1, gain) (steer.cpp:92-97) as set by cvs_batch_set_u8_gain, chunk by
 * chunk behind the pipeline launch, and only bytes come back while the next chunk is uploaded and filtered. */
int cvs_batch_run(cvs_batch b, const cvs_batch_cfg* cfg, const cvs_plane* inputs, const cvs_plane* outputs, cvs_batch_timing* timing);
/* 8-bit host outputs of cvs_batch_run: gain = 0 (default) normalises every map to its own min / max (the example without
 * --gain, steer.cpp:98-104), gain > 0 is Mat::convertTo(CV_8UC1, gain) (steer.cpp:92-97) */
int cvs_batch_set_u8_gain(cvs_batch b, float gain);
/* after a run with gather = 0 (or on a non-root rank): this rank's block, [n_frames][n_planes][rows][cols] dense */
int cvs_batch_local_result(cvs_batch b, int rank, float** data, int* n_frames, int* n_planes, int* rows, int* cols);
/* BASELINE config 3 -- one large image and its Gaussian pyramid split over the ranks by rows: ncclBroadcast of the
 * image from the root, every rank builds the (cheap) pyramid and runs cvs_setup_rows on its band of every level, the
 * bands are gathered into the ROOT's state planes.  Afterwards cvs_batch_level hands out, on the root, one ordinary
 * handle per level whose state (cvs_state_plane / cvs_read_state / cvs_steer_*) is complete -- bit-identical to a
 * single-GPU cvs_setup of that level. */
int cvs_batch_pyramid_setup(cvs_batch b, const cvs_plane* image, int rows, int cols, int levels, unsigned flags, int root,
                            cvs_batch_timing* timing);
int cvs_batch_level(cvs_batch b, int level, cvs_handle* h, cvs_plane* level_image);

#ifdef __cplusplus
}
#endif
#endif /* CVSTEER_HIP_H */
