// cvs_batch.cpp -- the batch axis over the GPUs of one node, natively (BASELINE configs 3 and 4).
//
// The reference's only parallel axis is independent images: example/steer.cpp:169 runs
// cv::parallel_for_(Range(0, N), body) with one fa::SteerableFiltersG2 per file (steer.cpp:69-124).  Here that
// axis is spread over GPUs: frame f of F belongs to rank floor(f * G / F) (contiguous blocks), every rank runs the
// fused pipeline on its block with NO collective on the data path, and RCCL moves data only at the edges:
//   * cvs_batch_run:            grouped ncclSend / ncclRecv from the root (scatter of input frames) and to the root
//                               (gather of the requested output planes) -- one point-to-point xGMI link per peer;
//   * cvs_batch_pyramid_setup:  one large image (+ its Gaussian pyramid): ncclBroadcast of the image, every rank
//                               filters its band of rows of every level (cvs_setup_rows; halo rows are read from the
//                               broadcast copy, so there is no exchange step), bands gathered into the root's state.
// Two ways to form the world: one process driving several devices (ncclCommInitAll), or one process per GPU
// (ncclCommInitRank with an id the caller distributes -- torch.distributed, MPI, a file).
//
// Built on the public single-GPU ABI only (cvs_create, cvs_pipeline_batch, cvs_setup_rows, ...).  RCCL is loaded at
// run time (dlopen of librccl.so.1, whichever copy the process already has) so that single-GPU users of the library
// do not need it.  Rehearsal: when a device is listed more than once (a one-GPU box standing in for several ranks)
// RCCL cannot be used -- it refuses two ranks on one device -- and the transport falls back to stream-ordered
// device copies; `cvs_batch_info` reports which transport is in use.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "cvsteer_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------------- RCCL, lazily
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl* rccl(std::string* why = nullptr)
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = std::getenv("CVS_RCCL_LIB");
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
        }
        if (!r.so) {
            r.why = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "not found");
            return;
        }
        bool ok = true;
        auto sym = [&](const char* name) {
            void* p = dlsym(r.so, name);
            if (!p) { ok = false; r.why = std::string("librccl lacks ") + name; }
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) {
            dlclose(r.so);
            r.so = nullptr;
        }
    });
    if (!r.so && why) *why = r.why;
    return r.so ? &r : nullptr;
}

// contiguous block [lo, hi) of `rank`: item f belongs to rank floor(f * world / n)
void shard_range(int n, int world, int rank, int* lo, int* hi)
{
    *lo = (int)(((long long)rank * n + world - 1) / world);
    *hi = (int)(((long long)(rank + 1) * n + world - 1) / world);
}

enum { TRANSPORT_NONE = 0, TRANSPORT_RCCL = 1, TRANSPORT_COPY = 2 };

struct DevBuf {
    float* p = nullptr;
    size_t elems = 0;
};

// one rank this process drives
struct Slot {
    int rank = 0, device = 0;
    hipStream_t stream = nullptr;
    cvs_handle h = nullptr;                 // the frame batch engine
    std::vector<cvs_handle> level;          // pyramid: one handle per level
    std::vector<DevBuf> level_img;          // pyramid: levels 1.. (level 0 = the broadcast image)
    ncclComm_t comm = nullptr;
    DevBuf in, out, image;                  // staging: input frames / output planes of this rank's shard; broadcast image
    DevBuf out8;                            // host 8-bit outputs: the shard's maps as bytes, ready to go down
    int last_n = 0, last_k = 0, last_rows = 0, last_cols = 0;  // layout of `out` after the last cvs_batch_run
    bool last_staged = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t up = nullptr, down = nullptr;  // host planes: upload / download streams beside `stream`
    std::vector<hipEvent_t> host_ev;           // host planes: the chunk events of host_rank, created once (4 timing + 2 per chunk)
    int* agree = nullptr;                   // 4 ints of device memory for the status agreement of a multi-process world
};

}  // namespace

struct cvs_batch_context {
    int kind = 0, width = 0, world = 0, transport = TRANSPORT_NONE;
    float spacing = 0.f;
    float u8_gain = 0.f;      // 8-bit host outputs: 0 = normalize(0, 255, MINMAX) per map, > 0 = convertTo(CV_8U, gain)
    std::vector<Slot> slots;  // local ranks, ascending
    // pyramid state of the last cvs_batch_pyramid_setup
    int pyr_levels = 0, pyr_root = 0;
    std::string err;
};

namespace {

int fail(cvs_batch b, int code, const std::string& what)
{
    if (b) b->err = what;
    return code;
}

#define B_HIP(b, expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e__ = (expr);                                                                         \
        if (e__ != hipSuccess) return fail(b, e__ == hipErrorOutOfMemory ? CVS_E_NOMEM : CVS_E_HIP,      \
                                           std::string(#expr) + ": " + hipGetErrorString(e__));         \
    } while (0)

#define B_NCCL(b, expr)                                                                                  \
    do {                                                                                                 \
        ncclResult_t r__ = (expr);                                                                       \
        if (r__ != ncclSuccess) return fail(b, CVS_E_HIP, std::string(#expr) + ": " + rccl()->GetErrorString(r__)); \
    } while (0)

#define B_CVS(b, h, expr)                                                                                \
    do {                                                                                                 \
        int rc__ = (expr);                                                                               \
        if (rc__ != CVS_OK) return fail(b, rc__, std::string(#expr) + ": " + cvs_last_error(h));         \
    } while (0)

int reserve(cvs_batch b, Slot& s, DevBuf& buf, size_t elems)
{
    if (elems <= buf.elems) return CVS_OK;
    B_HIP(b, hipSetDevice(s.device));
    if (buf.p) {
        B_HIP(b, hipStreamSynchronize(s.stream));
        B_HIP(b, hipFree(buf.p));
        buf = DevBuf();
    }
    B_HIP(b, hipMalloc(&buf.p, elems * sizeof(float)));
    buf.elems = elems;
    return CVS_OK;
}

Slot* local(cvs_batch b, int rank)
{
    for (Slot& s : b->slots)
        if (s.rank == rank) return &s;
    return nullptr;
}

int init_slot(cvs_batch b, Slot& s)
{
    B_HIP(b, hipSetDevice(s.device));
    B_HIP(b, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    for (hipEvent_t& e : s.ev) B_HIP(b, hipEventCreate(&e));
    // the 16 bytes the status agreement of a multi-process world travels in: allocated HERE, where a failure is a clean
    // create error -- inside agree() a rank that could not allocate them would drop out of the all-reduce and leave its
    // peers waiting, which is exactly what the agreement exists to prevent
    B_HIP(b, hipMalloc(&s.agree, 4 * sizeof(int)));
    int rc = cvs_create(b->kind, b->width, b->spacing, s.device, &s.h);
    if (rc != CVS_OK) return fail(b, rc, "cvs_create");
    B_CVS(b, s.h, cvs_set_stream(s.h, s.stream));
    return CVS_OK;
}

// ---- transport: point-to-point moves between ranks, enqueued on the ranks' streams ----
// A "move" = count floats from (src_rank, src) to (dst_rank, dst).  Both ends are queued inside one group; with RCCL an
// end that is not local to this process is simply not queued here (its own process queues it).
struct Move {
    int src_rank, dst_rank;
    const float* src;
    float* dst;
    size_t count;
};

// An RCCL group that was started MUST be ended, whatever happens inside it: a return between ncclGroupStart and
// ncclGroupEnd leaves the group open, and every later RCCL call of the process is silently queued into it.  So errors
// inside a group are collected, the group is closed, and only then does the call fail.
struct GroupGuard {
    Rccl* R;
    bool open = false;
    int rc = CVS_OK;
    std::string err;
    explicit GroupGuard(Rccl* r) : R(r) {}
    void start()
    {
        const ncclResult_t r = R->GroupStart();
        if (r == ncclSuccess) open = true;
        else note(CVS_E_HIP, std::string("ncclGroupStart: ") + R->GetErrorString(r));
    }
    void note(int code, const std::string& what)
    {
        if (rc == CVS_OK) { rc = code; err = what; }
    }
    void hip(hipError_t e, const char* what)
    {
        if (e != hipSuccess) note(e == hipErrorOutOfMemory ? CVS_E_NOMEM : CVS_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    }
    void nccl(ncclResult_t r, const char* what)
    {
        if (r != ncclSuccess) note(CVS_E_HIP, std::string(what) + ": " + R->GetErrorString(r));
    }
    bool ok() const { return rc == CVS_OK; }
    int end(cvs_batch b)
    {
        if (open) {
            open = false;
            nccl(R->GroupEnd(), "ncclGroupEnd");
        }
        return rc == CVS_OK ? CVS_OK : fail(b, rc, err);
    }
    ~GroupGuard()
    {
        if (open) (void)R->GroupEnd();
    }
};

int run_moves(cvs_batch b, const std::vector<Move>& moves)
{
    if (moves.empty()) return CVS_OK;
    if (b->transport == TRANSPORT_RCCL) {
        GroupGuard g(rccl());
        g.start();
        for (const Move& m : moves) {
            if (!g.ok()) break;
            if (Slot* s = local(b, m.src_rank)) {
                g.hip(hipSetDevice(s->device), "hipSetDevice");
                if (g.ok()) g.nccl(g.R->Send(m.src, m.count, ncclFloat, m.dst_rank, s->comm, s->stream), "ncclSend");
            }
            if (!g.ok()) break;
            if (Slot* d = local(b, m.dst_rank)) {
                g.hip(hipSetDevice(d->device), "hipSetDevice");
                if (g.ok()) g.nccl(g.R->Recv(m.dst, m.count, ncclFloat, m.src_rank, d->comm, d->stream), "ncclRecv");
            }
        }
        return g.end(b);
    }
    // rehearsal transport (ranks sharing a device, one process): the copy runs on the destination's stream after an
    // event on the source's stream -- the same ordering a send / recv pair gives
    for (const Move& m : moves) {
        Slot* s = local(b, m.src_rank);
        Slot* d = local(b, m.dst_rank);
        if (!s || !d) return fail(b, CVS_E_UNSUPPORTED, "copy transport needs both ranks in this process");
        B_HIP(b, hipSetDevice(s->device));
        hipEvent_t e;
        B_HIP(b, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        B_HIP(b, hipEventRecord(e, s->stream));
        B_HIP(b, hipSetDevice(d->device));
        B_HIP(b, hipStreamWaitEvent(d->stream, e, 0));
        B_HIP(b, hipMemcpyAsync(m.dst, m.src, m.count * sizeof(float), hipMemcpyDeviceToDevice, d->stream));
        // the source must not be overwritten before the copy has run: order the source stream behind it
        hipEvent_t done;
        B_HIP(b, hipEventCreateWithFlags(&done, hipEventDisableTiming));
        B_HIP(b, hipEventRecord(done, d->stream));
        B_HIP(b, hipSetDevice(s->device));
        B_HIP(b, hipStreamWaitEvent(s->stream, done, 0));
        B_HIP(b, hipEventDestroy(e));
        B_HIP(b, hipEventDestroy(done));
    }
    return CVS_OK;
}

// Ranks of a world spread over SEVERAL processes agree on "may data be queued?" before any of them queues a message:
// a failure that only one process can see (the root's planes are not what cvs_batch_run needs, a staging allocation
// fails on one rank, the ranks were handed different geometries) would otherwise leave the others inside ncclRecv for
// ever.  One 4-int all-reduce (min) per call: {status, geometry hash, -geometry hash, 0}; the hashes agree iff
// min(h) == -min(-h).  A world that lives in one process has nothing to agree on (the same code path sees every rank's
// status before anything is queued) and skips this.  Returns local_rc if that is already an error.
int agree(cvs_batch b, int local_rc, uint32_t geometry_hash)
{
    // CVS_BATCH_FORCE_AGREE=1 (tests): a world that lives in one process goes through the agreement too, so that the code
    // path can be exercised on a box with one GPU (a one-rank RCCL world: the all-reduce is degenerate but real)
    static const bool force = std::getenv("CVS_BATCH_FORCE_AGREE") != nullptr;
    if (b->transport != TRANSPORT_RCCL || ((int)b->slots.size() == b->world && !force)) return local_rc;
    Rccl* R = rccl();
    const int h31 = (int)(geometry_hash & 0x3fffffffu);
    int mine[4] = {local_rc, h31, -h31, 0};
    int rc = CVS_OK;
    std::string err;
    for (Slot& s : b->slots) {
        if (hipSetDevice(s.device) != hipSuccess || hipMemcpyAsync(s.agree, mine, sizeof(mine), hipMemcpyHostToDevice, s.stream) != hipSuccess) {
            // the status could not be staged: this rank still takes part in the all-reduce below (its peers must not wait),
            // and reports the failure afterwards
            (void)hipGetLastError();
            rc = CVS_E_HIP;
            err = "status agreement: the status could not be staged on the device";
        }
    }
    {
        GroupGuard g(R);
        g.start();
        for (Slot& s : b->slots) {
            if (!g.ok()) break;
            g.hip(hipSetDevice(s.device), "hipSetDevice");
            if (g.ok()) g.nccl(R->AllReduce(s.agree, s.agree, 4, ncclInt32, ncclMin, s.comm, s.stream), "ncclAllReduce");
        }
        const int grc = g.end(b);
        if (grc != CVS_OK) return grc;
    }
    if (rc != CVS_OK) return fail(b, rc, err);
    int agreed[4] = {0, 0, 0, 0};
    for (Slot& s : b->slots) {
        B_HIP(b, hipSetDevice(s.device));
        B_HIP(b, hipMemcpyAsync(agreed, s.agree, sizeof(agreed), hipMemcpyDeviceToHost, s.stream));
        B_HIP(b, hipStreamSynchronize(s.stream));
        if (agreed[0] != CVS_OK && local_rc == CVS_OK)
            return fail(b, agreed[0], "a peer rank cannot run this call (status " + std::to_string(agreed[0]) + "); nothing was queued");
        if (agreed[1] != -agreed[2] && local_rc == CVS_OK)
            return fail(b, CVS_E_BADARG, "the ranks were called with different geometries / options; nothing was queued");
    }
    return local_rc;
}

uint32_t hash_ints(std::initializer_list<long long> v)
{
    uint64_t h = 1469598103934665603ull;
    for (long long x : v) {
        h ^= (uint64_t)x;
        h *= 1099511628211ull;
    }
    return (uint32_t)(h ^ (h >> 32));
}

int record_all(cvs_batch b, int which)
{
    for (Slot& s : b->slots) {
        B_HIP(b, hipSetDevice(s.device));
        B_HIP(b, hipEventRecord(s.ev[which], s.stream));
    }
    return CVS_OK;
}

int sync_all(cvs_batch b)
{
    for (Slot& s : b->slots) {
        B_HIP(b, hipSetDevice(s.device));
        B_HIP(b, hipStreamSynchronize(s.stream));
    }
    return CVS_OK;
}

void fill_timing(cvs_batch b, cvs_batch_timing* t)
{
    if (!t) return;
    t->scatter_ms = t->compute_ms = t->gather_ms = 0.0;
    for (Slot& s : b->slots) {
        float ms[3] = {0.f, 0.f, 0.f};
        (void)hipSetDevice(s.device);
        for (int i = 0; i < 3; ++i) (void)hipEventElapsedTime(&ms[i], s.ev[i], s.ev[i + 1]);
        t->scatter_ms = std::max(t->scatter_ms, (double)ms[0]);
        t->compute_ms = std::max(t->compute_ms, (double)ms[1]);
        t->gather_ms = std::max(t->gather_ms, (double)ms[2]);
    }
}

// is `p` a dense rows x cols f32 device plane (no row padding)?
bool dense_device(const cvs_plane* p, int rows, int cols)
{
    return p && p->data && p->mem == CVS_MEM_DEVICE && p->rows == rows && p->cols == cols && p->step == (size_t)cols * sizeof(float);
}

// is `p` a rows x cols f32 HOST plane (rows may be padded)?
bool host_plane(const cvs_plane* p, int rows, int cols, bool u8 = false)
{
    return p && p->data && p->mem == (u8 ? (CVS_MEM_HOST | CVS_DEPTH_U8) : CVS_MEM_HOST) && p->rows == rows && p->cols == cols &&
           p->step >= (size_t)cols * (u8 ? 1 : sizeof(float));
}

// Host planes (the reference's callers hold cv::Mat: example/steer.cpp:73-104).  Nothing goes through the root's GPU:
// every rank pulls ITS frames from the caller's host planes over its own PCIe link and pushes its outputs back the
// same way, all ranks at once.  Inside a rank the shard is cut into chunks of frames and three things overlap, as
// in the single-image host path (cvs_host.cpp): the upload of chunk c+1 (stream `up`, this rank's worker thread),
// the launch for chunk c (the rank's stream) and the download of chunk c-1 (stream `down`, a second thread --
// copies from / to pageable memory hold the calling thread).  Needs every rank in this process.
struct HostRun {
    cvs_batch b;
    const cvs_batch_cfg* cfg;
    const cvs_plane *inputs, *outputs;
    int sel[8], K;
    bool u8;  // 8-bit input frames: bytes cross the link, the engine widens them on the device
    bool out8;  // 8-bit host output planes: the maps are normalised / converted on the device, bytes come back
};

int host_rank(const HostRun& R, Slot& s, std::string& err, double ms[3])
{
    cvs_batch b = R.b;
    const int rows = R.cfg->rows, cols = R.cfg->cols, K = R.K;
    const size_t plane = (size_t)rows * cols, rowb = (size_t)cols * sizeof(float);
    int lo, hi;
    shard_range(R.cfg->n_frames, b->world, s.rank, &lo, &hi);
    const int n = hi - lo;
    ms[0] = ms[1] = ms[2] = 0.0;
    if (!n) {
        s.last_n = 0;
        s.last_staged = true;
        return CVS_OK;
    }
#define H_TRY(expr)                                                                              \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e__); return e__ == hipErrorOutOfMemory ? CVS_E_NOMEM : CVS_E_HIP; } \
    } while (0)
    H_TRY(hipSetDevice(s.device));
    if (!s.up) {
        H_TRY(hipStreamCreateWithFlags(&s.up, hipStreamNonBlocking));
        H_TRY(hipStreamCreateWithFlags(&s.down, hipStreamNonBlocking));
    }
    int persist = 1;
    (void)cvs_get_option(s.h, CVS_OPT_PERSIST_STATE, &persist);
    // with state kept, the handle's frames after the call must be the whole shard: one chunk
    // Chunks GROW (round 6): the downloads set the pace (three maps down for one frame up) and run back to back once the first chunk's
    // maps exist, so what the chunking costs is the time until then -- upload + launch of the FIRST chunk.  A thirty-second of the shard
    // first, every later chunk twice its predecessor (its upload and launch hide behind the predecessor's download): 1 | 2 | 4 | 8 | 17 frames
    // for a shard of 32 instead of four chunks of 8: 0.79 -> 0.84 of the link's roof for 8-bit frames in / three 8-bit maps out
    // (profiles/r06_host_chunks.txt).
    std::vector<int> c0(1, 0);
    if (persist) c0.push_back(n);
    else
        for (int sz = std::max(1, n / 32); c0.back() < n; sz *= 2) c0.push_back((n - c0.back() <= sz + sz / 2 || c0.size() >= 5) ? n : c0.back() + sz);
    const int nchunks = (int)c0.size() - 1;
    // the events are the slot's own (created once: a dozen hipEventCreate / Destroy per call were a tenth of a small batch's time)
    while ((int)s.host_ev.size() < 4 + 2 * nchunks) {
        hipEvent_t ev = nullptr;
        H_TRY(s.host_ev.size() < 4 ? hipEventCreate(&ev) : hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        s.host_ev.push_back(ev);
    }
    hipEvent_t* t = s.host_ev.data();
    hipEvent_t* up_ev = s.host_ev.data() + 4;
    hipEvent_t* done_ev = s.host_ev.data() + 4 + nchunks;
    // the download thread: chunk c may start once its launch has been queued (counter) and has finished (event)
    std::mutex mu;
    std::condition_variable cv;
    int queued = 0;
    bool stop = false;
    int drc = CVS_OK;
    std::string derr;
    std::thread down;
    if (R.cfg->gather) {
        down = std::thread([&]() {
            if (hipSetDevice(s.device) != hipSuccess) { drc = CVS_E_HIP; derr = "hipSetDevice"; return; }
            for (int c = 0; c < nchunks; ++c) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return queued > c || stop; });
                    if (stop) return;
                }
                hipError_t e = hipStreamWaitEvent(s.down, done_ev[c], 0);
                if (c == 0 && e == hipSuccess) e = hipEventRecord(t[2], s.down);
                if (R.out8) {
                    // bytes: dense planes that lie back to back on the host (the usual [n][K][rows][cols] block) leave as
                    // ONE linear copy per run -- a pitched 2-D copy of the same bytes runs at a third of the link rate
                    const uint8_t* dev8 = reinterpret_cast<const uint8_t*>(s.out8.p);
                    size_t first = (size_t)c0[c] * K, count = 0;
                    uint8_t* run_dst = nullptr;
                    auto flush = [&]() {
                        if (count && e == hipSuccess) e = hipMemcpyAsync(run_dst, dev8 + first * plane, count * plane, hipMemcpyDeviceToHost, s.down);
                        count = 0;
                    };
                    for (int i = c0[c]; e == hipSuccess && i < c0[c + 1]; ++i)
                        for (int j = 0; e == hipSuccess && j < K; ++j) {
                            const cvs_plane& o = R.outputs[(size_t)(lo + i) * 8 + R.sel[j]];
                            uint8_t* dst = reinterpret_cast<uint8_t*>(o.data);
                            const size_t idx = (size_t)i * K + j;
                            if (o.step != (size_t)cols) {  // padded rows: a 2-D copy of this plane alone
                                flush();
                                if (e == hipSuccess) e = hipMemcpy2DAsync(dst, o.step, dev8 + idx * plane, (size_t)cols, (size_t)cols, rows, hipMemcpyDeviceToHost, s.down);
                                continue;
                            }
                            if (count && dst == run_dst + count * plane) { ++count; continue; }
                            flush();
                            first = idx; run_dst = dst; count = 1;
                        }
                    flush();
                } else {
                    // f32 planes: the same -- dense planes that lie back to back leave as one linear copy per run
                    size_t first = (size_t)c0[c] * K, count = 0;
                    float* run_dst = nullptr;
                    auto flush = [&]() {
                        if (count && e == hipSuccess) e = hipMemcpyAsync(run_dst, s.out.p + first * plane, count * plane * sizeof(float), hipMemcpyDeviceToHost, s.down);
                        count = 0;
                    };
                    for (int i = c0[c]; e == hipSuccess && i < c0[c + 1]; ++i)
                        for (int j = 0; e == hipSuccess && j < K; ++j) {
                            const cvs_plane& o = R.outputs[(size_t)(lo + i) * 8 + R.sel[j]];
                            const size_t idx = (size_t)i * K + j;
                            if (o.step != rowb) {
                                flush();
                                if (e == hipSuccess) e = hipMemcpy2DAsync(o.data, o.step, s.out.p + idx * plane, rowb, rowb, rows, hipMemcpyDeviceToHost, s.down);
                                continue;
                            }
                            if (count && o.data == run_dst + count * plane) { ++count; continue; }
                            flush();
                            first = idx; run_dst = o.data; count = 1;
                        }
                    flush();
                }
                if (e != hipSuccess) { drc = CVS_E_HIP; derr = std::string("download: ") + hipGetErrorString(e); return; }
            }
            hipError_t e = hipEventRecord(t[3], s.down);
            if (e == hipSuccess) e = hipStreamSynchronize(s.down);
            if (e != hipSuccess) { drc = CVS_E_HIP; derr = std::string("download: ") + hipGetErrorString(e); }
        });
    }
    auto stop_down = [&](bool failed) {
        if (down.joinable()) {
            {
                std::lock_guard<std::mutex> lk(mu);
                if (failed) stop = true;
            }
            cv.notify_all();
            down.join();
        }
    };
    int rc = CVS_OK;
    hipError_t e = hipEventRecord(t[0], s.up);
    for (int c = 0; c < nchunks && e == hipSuccess && rc == CVS_OK; ++c) {
        {
            // dense frames that lie back to back on the host (one [n][rows][cols] block) go up as one linear copy per run
            const size_t esz = R.u8 ? 1 : sizeof(float), frame_b = plane * esz;
            uint8_t* dev = reinterpret_cast<uint8_t*>(s.in.p);
            int first = c0[c], count = 0;
            const uint8_t* run_src = nullptr;
            auto flush = [&]() {
                if (count && e == hipSuccess) e = hipMemcpyAsync(dev + (size_t)first * frame_b, run_src, (size_t)count * frame_b, hipMemcpyHostToDevice, s.up);
                count = 0;
            };
            for (int i = c0[c]; e == hipSuccess && i < c0[c + 1]; ++i) {
                const cvs_plane& im = R.inputs[lo + i];
                const uint8_t* src = reinterpret_cast<const uint8_t*>(im.data);
                if (im.step != (size_t)cols * esz) {
                    flush();
                    if (e == hipSuccess) e = hipMemcpy2DAsync(dev + (size_t)i * frame_b, (size_t)cols * esz, im.data, im.step, (size_t)cols * esz, rows, hipMemcpyHostToDevice, s.up);
                    continue;
                }
                if (count && src == run_src + (size_t)count * frame_b) { ++count; continue; }
                flush();
                first = i; run_src = src; count = 1;
            }
            flush();
        }
        if (e == hipSuccess) e = hipEventRecord(up_ev[c], s.up);
        if (e == hipSuccess && c == nchunks - 1) e = hipEventRecord(t[1], s.up);
        if (e == hipSuccess) e = hipStreamWaitEvent(s.stream, up_ev[c], 0);
        if (e != hipSuccess) break;
        const int cn = c0[c + 1] - c0[c];
        std::vector<cvs_plane> im(cn), ou((size_t)cn * 8);
        std::memset(ou.data(), 0, ou.size() * sizeof(cvs_plane));
        for (int i = 0; i < cn; ++i) {
            if (R.u8) im[i] = cvs_plane{reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(s.in.p) + (size_t)(c0[c] + i) * plane), rows, cols, (size_t)cols, CVS_MEM_DEVICE | CVS_DEPTH_U8};
            else im[i] = cvs_plane{s.in.p + (size_t)(c0[c] + i) * plane, rows, cols, rowb, CVS_MEM_DEVICE};
            for (int j = 0; j < K; ++j)
                ou[(size_t)i * 8 + R.sel[j]] = cvs_plane{s.out.p + ((size_t)(c0[c] + i) * K + j) * plane, rows, cols, rowb, CVS_MEM_DEVICE};
        }
        rc = cvs_pipeline_batch(s.h, im.data(), cn, ou.data());
        if (rc != CVS_OK) { err = std::string("cvs_pipeline_batch: ") + cvs_last_error(s.h); break; }
        if (R.out8) {  // the chunk's maps -> bytes on the device (one min/max + one quantise launch, queued behind the pipeline launch)
            const size_t m = (size_t)cn * K;
            std::vector<cvs_plane> src(m);
            std::vector<uint8_t*> dst(m);
            for (size_t q = 0; q < m; ++q) {
                src[q] = cvs_plane{s.out.p + ((size_t)c0[c] * K + q) * plane, rows, cols, rowb, CVS_MEM_DEVICE};
                dst[q] = reinterpret_cast<uint8_t*>(s.out8.p) + ((size_t)c0[c] * K + q) * plane;
            }
            rc = b->u8_gain > 0.f ? cvs_convert_u8_batch(s.h, src.data(), (int)m, b->u8_gain, 0.f, dst.data(), (size_t)cols, CVS_MEM_DEVICE)
                                  : cvs_normalize_u8_batch(s.h, src.data(), (int)m, dst.data(), (size_t)cols, CVS_MEM_DEVICE);
            if (rc != CVS_OK) { err = std::string("cvs_*_u8_batch: ") + cvs_last_error(s.h); break; }
        }
        e = hipEventRecord(done_ev[c], s.stream);
        if (e == hipSuccess) {
            {
                std::lock_guard<std::mutex> lk(mu);
                queued = c + 1;
            }
            cv.notify_all();
        }
    }
    const bool failed = e != hipSuccess || rc != CVS_OK;
    if (e != hipSuccess) { err = std::string("upload: ") + hipGetErrorString(e); rc = CVS_E_HIP; }
    stop_down(failed);
    if (!failed && drc != CVS_OK) { rc = drc; err = derr; }
    if (rc == CVS_OK) {
        e = hipStreamSynchronize(s.stream);
        if (e != hipSuccess) { err = hipGetErrorString(e); rc = CVS_E_HIP; }
    }
    if (rc == CVS_OK) {
        float a = 0.f, g = 0.f, w = 0.f;
        (void)hipEventElapsedTime(&a, t[0], t[1]);
        ms[0] = a;
        if (R.cfg->gather) {
            (void)hipEventElapsedTime(&g, t[2], t[3]);
            (void)hipEventElapsedTime(&w, t[0], t[3]);
            ms[2] = g;
            ms[1] = w;  // host planes: the middle figure is the rank's whole span, upload start to download end
        }
    } else {
        (void)hipDeviceSynchronize();
    }
    s.last_n = n;
    s.last_k = K;
    s.last_rows = rows;
    s.last_cols = cols;
    s.last_staged = true;
#undef H_TRY
    return rc;
}

int run_host(cvs_batch b, const cvs_batch_cfg* cfg, const cvs_plane* inputs, const cvs_plane* outputs, const int* sel, int K,
             cvs_batch_timing* timing)
{
    const int rows = cfg->rows, cols = cfg->cols, F = cfg->n_frames;
    if ((int)b->slots.size() != b->world)
        return fail(b, CVS_E_UNSUPPORTED, "host planes need every rank in the calling process (each GPU pulls its frames over its own link)");
    const bool u8 = (inputs[0].mem & CVS_DEPTH_U8) != 0;
    // host outputs may be 8-bit planes too (what the example writes, steer.cpp:92-122): the maps are then normalised /
    // converted on the device chunk by chunk and only bytes come back, overlapped with the uploads and launches of the next chunk
    const bool out8 = cfg->gather && outputs && (outputs[sel[0]].mem & CVS_DEPTH_U8) != 0;
    for (int f = 0; f < F; ++f) {
        if (!host_plane(&inputs[f], rows, cols, u8)) return fail(b, CVS_E_SIZE, "input frames must be all host (all f32 or all 8-bit) or all dense device f32 planes of rows x cols");
        for (int j = 0; cfg->gather && j < K; ++j)
            if (!host_plane(&outputs[(size_t)f * 8 + sel[j]], rows, cols, out8))
                return fail(b, CVS_E_SIZE, "with host input frames the requested output planes must be host planes of rows x cols, all f32 or all 8-bit");
    }
    const size_t plane = (size_t)rows * cols;
    int rc;
    for (Slot& s : b->slots) {
        int lo, hi;
        shard_range(F, b->world, s.rank, &lo, &hi);
        const size_t n = (size_t)(hi - lo);
        if (n && (rc = reserve(b, s, s.in, n * plane))) return rc;
        if (n && (rc = reserve(b, s, s.out, n * K * plane))) return rc;
        if (n && out8 && (rc = reserve(b, s, s.out8, (n * K * plane + 3) / 4 + 64))) return rc;
    }
    HostRun R{b, cfg, inputs, outputs, {0}, K, u8, out8};
    for (int j = 0; j < K; ++j) R.sel[j] = sel[j];
    const size_t nl = b->slots.size();
    std::vector<int> rcs(nl, CVS_OK);
    std::vector<std::string> errs(nl);
    std::vector<double> ms(nl * 3, 0.0);
    std::vector<std::thread> th;
    for (size_t i = 1; i < nl; ++i) th.emplace_back([&, i]() { rcs[i] = host_rank(R, b->slots[i], errs[i], &ms[i * 3]); });
    rcs[0] = host_rank(R, b->slots[0], errs[0], &ms[0]);
    for (std::thread& t : th) t.join();
    for (size_t i = 0; i < nl; ++i)
        if (rcs[i] != CVS_OK) return fail(b, rcs[i], "rank " + std::to_string(b->slots[i].rank) + ": " + errs[i]);
    if (timing) {
        timing->scatter_ms = timing->compute_ms = timing->gather_ms = 0.0;
        for (size_t i = 0; i < nl; ++i) {
            timing->scatter_ms = std::max(timing->scatter_ms, ms[i * 3]);
            timing->compute_ms = std::max(timing->compute_ms, ms[i * 3 + 1]);
            timing->gather_ms = std::max(timing->gather_ms, ms[i * 3 + 2]);
        }
    }
    return CVS_OK;
}

int create_common(int kind, int width, float spacing, cvs_batch* out, cvs_batch* made)
{
    if (!out) return CVS_E_BADARG;
    *out = nullptr;
    if (cvs_num_basis(kind) == 0 || width < 1) return CVS_E_BADARG;
    cvs_batch b = new (std::nothrow) cvs_batch_context();
    if (!b) return CVS_E_NOMEM;
    b->kind = kind;
    b->width = width;
    b->spacing = spacing;
    *made = b;
    return CVS_OK;
}

}  // namespace

extern "C" {

int cvs_batch_create_local(int kind, int width, float spacing, int ndev, const int* devices, cvs_batch* out)
{
    cvs_batch b = nullptr;
    int rc = create_common(kind, width, spacing, out, &b);
    if (rc) return rc;
    if (ndev < 1 || !devices) { delete b; return CVS_E_BADARG; }
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) { delete b; return CVS_E_HIP; }  // no CPU fallback
    bool dup = false;
    for (int i = 0; i < ndev; ++i) {
        if (devices[i] < 0 || devices[i] >= have) { delete b; return CVS_E_BADARG; }
        for (int j = 0; j < i; ++j) dup = dup || devices[j] == devices[i];
    }
    b->world = ndev;
    b->slots.resize(ndev);
    for (int i = 0; i < ndev; ++i) {
        b->slots[i].rank = i;
        b->slots[i].device = devices[i];
        if ((rc = init_slot(b, b->slots[i]))) { *out = b; return rc; }  // caller reads the error, then destroys
    }
    if (dup) {
        b->transport = TRANSPORT_COPY;   // rehearsal: several ranks on one device
    } else {
        std::string why;
        Rccl* R = rccl(&why);
        if (!R) {
            if (ndev == 1) b->transport = TRANSPORT_NONE;  // one rank needs no transport at all
            else { *out = b; return fail(b, CVS_E_HIP, "RCCL is not available: " + why); }
        } else {
            std::vector<ncclComm_t> comms(ndev);
            ncclResult_t r = R->CommInitAll(comms.data(), ndev, devices);
            if (r != ncclSuccess) { *out = b; return fail(b, CVS_E_HIP, std::string("ncclCommInitAll: ") + R->GetErrorString(r)); }
            for (int i = 0; i < ndev; ++i) b->slots[i].comm = comms[i];
            b->transport = TRANSPORT_RCCL;
        }
    }
    *out = b;
    return CVS_OK;
}

int cvs_batch_unique_id(void* id128)
{
    if (!id128) return CVS_E_BADARG;
    Rccl* R = rccl();
    if (!R) return CVS_E_HIP;
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) return CVS_E_HIP;
    static_assert(sizeof(id) == CVS_BATCH_ID_BYTES, "ncclUniqueId size");
    std::memcpy(id128, &id, sizeof(id));
    return CVS_OK;
}

int cvs_batch_create_rank(int kind, int width, float spacing, const void* id128, int world, int rank, int device, cvs_batch* out)
{
    cvs_batch b = nullptr;
    int rc = create_common(kind, width, spacing, out, &b);
    if (rc) return rc;
    if (!id128 || world < 1 || rank < 0 || rank >= world) { delete b; return CVS_E_BADARG; }
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) { delete b; return CVS_E_HIP; }
    if (device < 0 || device >= have) { delete b; return CVS_E_BADARG; }
    b->world = world;
    b->slots.resize(1);
    b->slots[0].rank = rank;
    b->slots[0].device = device;
    *out = b;
    if ((rc = init_slot(b, b->slots[0]))) return rc;
    std::string why;
    Rccl* R = rccl(&why);
    if (!R) return fail(b, CVS_E_HIP, "RCCL is not available: " + why);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    B_HIP(b, hipSetDevice(device));
    ncclResult_t r = R->CommInitRank(&b->slots[0].comm, world, id, rank);
    if (r != ncclSuccess) return fail(b, CVS_E_HIP, std::string("ncclCommInitRank: ") + R->GetErrorString(r));
    b->transport = TRANSPORT_RCCL;
    return CVS_OK;
}

int cvs_batch_destroy(cvs_batch b)
{
    if (!b) return CVS_E_BADARG;
    for (Slot& s : b->slots) {
        (void)hipSetDevice(s.device);
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.comm && rccl()) (void)rccl()->CommDestroy(s.comm);
        if (s.h) (void)cvs_destroy(s.h);
        for (cvs_handle h : s.level)
            if (h) (void)cvs_destroy(h);
        for (DevBuf& d : s.level_img)
            if (d.p) (void)hipFree(d.p);
        for (DevBuf* d : {&s.in, &s.out, &s.image, &s.out8})
            if (d->p) (void)hipFree(d->p);
        for (hipEvent_t e : s.ev)
            if (e) (void)hipEventDestroy(e);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        for (hipEvent_t ev : s.host_ev) (void)hipEventDestroy(ev);
        if (s.up) (void)hipStreamDestroy(s.up);
        if (s.down) (void)hipStreamDestroy(s.down);
        if (s.agree) (void)hipFree(s.agree);
    }
    delete b;
    return CVS_OK;
}

const char* cvs_batch_last_error(cvs_batch b) { return b ? b->err.c_str() : "null batch"; }

int cvs_batch_info(cvs_batch b, int* world, int* nlocal, int* transport)
{
    if (!b) return CVS_E_BADARG;
    if (world) *world = b->world;
    if (nlocal) *nlocal = (int)b->slots.size();
    if (transport) *transport = b->transport;
    return CVS_OK;
}

int cvs_batch_set_u8_gain(cvs_batch b, float gain)
{
    if (!b) return CVS_E_BADARG;
    if (!(gain >= 0.f)) return fail(b, CVS_E_BADARG, "gain");
    b->u8_gain = gain;
    return CVS_OK;
}

int cvs_batch_set_option(cvs_batch b, int option, int value)
{
    if (!b) return CVS_E_BADARG;
    for (Slot& s : b->slots) B_CVS(b, s.h, cvs_set_option(s.h, option, value));
    return CVS_OK;
}

int cvs_batch_run(cvs_batch b, const cvs_batch_cfg* cfg, const cvs_plane* inputs, const cvs_plane* outputs, cvs_batch_timing* timing)
{
    if (!b || !cfg) return CVS_E_BADARG;
    if (b->kind != CVS_KIND_G2) return fail(b, CVS_E_UNSUPPORTED, "the caller pipeline exists for G2 only");
    const int rows = cfg->rows, cols = cfg->cols, F = cfg->n_frames, root = cfg->root;
    if (rows <= 0 || cols <= 0 || F < 1) return fail(b, CVS_E_SIZE, "empty batch");
    if (root < 0 || root >= b->world) return fail(b, CVS_E_BADARG, "root");
    if (!(cfg->outputs & 0xffu)) return fail(b, CVS_E_BADARG, "no output requested");
    int sel[8], K = 0;
    for (int k = 0; k < 8; ++k)
        if (cfg->outputs & (1u << k)) sel[K++] = k;
    const size_t plane = (size_t)rows * cols;
    Slot* rs = local(b, root);
    const bool via = cfg->self_via_transport != 0 && b->transport != TRANSPORT_NONE;
    if (b->world > 1 && b->transport == TRANSPORT_NONE) return fail(b, CVS_E_HIP, "no transport");
    // What only THIS process can know -- are the root's planes usable, do the staging buffers fit -- is collected into
    // local_rc without returning, and the ranks agree on it before anybody queues a message (see agree()): a rank that
    // returned early here would leave its peers inside ncclRecv for ever.
    int local_rc = CVS_OK;
    auto lfail = [&](int code, const char* what) {
        if (local_rc == CVS_OK) local_rc = fail(b, code, what);
    };
    bool host = false;
    if (rs) {
        if (!inputs) lfail(CVS_E_BADARG, "the root rank needs the input frames");
        else if (cfg->gather && !outputs) lfail(CVS_E_BADARG, "the root rank needs output planes to gather into");
        else if ((inputs[0].mem & 0xff) == CVS_MEM_HOST) {
            host = true;
            if ((int)b->slots.size() != b->world)
                lfail(CVS_E_UNSUPPORTED, "host planes need every rank in the calling process (each GPU pulls its frames over its own link)");
        } else {
            for (int f = 0; f < F && local_rc == CVS_OK; ++f) {
                if (!dense_device(&inputs[f], rows, cols)) lfail(CVS_E_SIZE, "input frames must be dense f32 device planes of rows x cols");
                for (int j = 0; cfg->gather && j < K && local_rc == CVS_OK; ++j)
                    if (!dense_device(&outputs[(size_t)f * 8 + sel[j]], rows, cols))
                        lfail(CVS_E_SIZE, "requested output planes must be dense f32 device planes of rows x cols");
            }
        }
    }
    int rc;
    // staging on every local rank that does not work in place
    for (Slot& s : b->slots) {
        if (local_rc != CVS_OK || host) break;
        int lo, hi;
        shard_range(F, b->world, s.rank, &lo, &hi);
        const size_t n = (size_t)(hi - lo);
        const bool in_place = s.rank == root && !via;
        if (!in_place && n && (rc = reserve(b, s, s.in, n * plane))) { local_rc = rc; break; }
        // the root writes its own shard straight into the caller's planes when it gathers; everybody else stages
        const bool out_in_place = s.rank == root && cfg->gather && !via;
        if (!out_in_place && n && (rc = reserve(b, s, s.out, n * K * plane))) { local_rc = rc; break; }
    }
    if ((rc = agree(b, local_rc, hash_ints({rows, cols, F, (long long)cfg->outputs, root, cfg->gather, cfg->self_via_transport, 4}))))
        return rc;
    if (host) return run_host(b, cfg, inputs, outputs, sel, K, timing);   // every rank is in this process (checked above)
    if ((rc = record_all(b, 0))) return rc;
    // ---- scatter: root -> ranks, frame by frame (a frame is one contiguous message) ----
    {
        std::vector<Move> mv;
        for (int r = 0; r < b->world; ++r) {
            if (r == root && !via) continue;
            int lo, hi;
            shard_range(F, b->world, r, &lo, &hi);
            Slot* d = local(b, r);
            for (int f = lo; f < hi; ++f)
                mv.push_back({root, r, rs ? inputs[f].data : nullptr, d ? d->in.p + (size_t)(f - lo) * plane : nullptr, plane});
        }
        if ((rc = run_moves(b, mv))) return rc;
    }
    if ((rc = record_all(b, 1))) return rc;
    // ---- compute: the fused pipeline on every local rank's block, one launch per rank ----
    for (Slot& s : b->slots) {
        int lo, hi;
        shard_range(F, b->world, s.rank, &lo, &hi);
        const int n = hi - lo;
        if (!n) {
            s.last_n = 0;  // cvs_batch_local_result: an empty block, not the previous run's
            s.last_staged = true;
            continue;
        }
        const bool in_place = s.rank == root && !via;
        const bool out_in_place = s.rank == root && cfg->gather && !via;
        std::vector<cvs_plane> im(n), ou((size_t)n * 8);
        std::memset(ou.data(), 0, ou.size() * sizeof(cvs_plane));
        for (int i = 0; i < n; ++i) {
            im[i] = in_place ? inputs[lo + i] : cvs_plane{s.in.p + (size_t)i * plane, rows, cols, (size_t)cols * sizeof(float), CVS_MEM_DEVICE};
            for (int j = 0; j < K; ++j)
                ou[(size_t)i * 8 + sel[j]] = out_in_place ? outputs[(size_t)(lo + i) * 8 + sel[j]]
                                                          : cvs_plane{s.out.p + ((size_t)i * K + j) * plane, rows, cols, (size_t)cols * sizeof(float), CVS_MEM_DEVICE};
        }
        B_CVS(b, s.h, cvs_pipeline_batch(s.h, im.data(), n, ou.data()));
        s.last_n = n;
        s.last_k = K;
        s.last_rows = rows;
        s.last_cols = cols;
        s.last_staged = !out_in_place;
    }
    if ((rc = record_all(b, 2))) return rc;
    // ---- gather: ranks -> root, plane by plane ----
    if (cfg->gather) {
        std::vector<Move> mv;
        for (int r = 0; r < b->world; ++r) {
            if (r == root && !via) continue;
            int lo, hi;
            shard_range(F, b->world, r, &lo, &hi);
            Slot* s = local(b, r);
            for (int f = lo; f < hi; ++f)
                for (int j = 0; j < K; ++j)
                    mv.push_back({r, root, s ? s->out.p + ((size_t)(f - lo) * K + j) * plane : nullptr,
                                  rs ? outputs[(size_t)f * 8 + sel[j]].data : nullptr, plane});
        }
        if ((rc = run_moves(b, mv))) return rc;
    }
    if ((rc = record_all(b, 3))) return rc;
    if ((rc = sync_all(b))) return rc;   // outputs are on the root when the call returns
    fill_timing(b, timing);
    return CVS_OK;
}

int cvs_batch_local_result(cvs_batch b, int rank, float** data, int* n_frames, int* n_planes, int* rows, int* cols)
{
    if (!b || !data) return CVS_E_BADARG;
    Slot* s = local(b, rank);
    if (!s) return fail(b, CVS_E_BADARG, "rank is not local to this process");
    if (!s->last_staged) return fail(b, CVS_E_STATE, "this rank wrote its block straight into the caller's planes");
    *data = s->out.p;
    if (n_frames) *n_frames = s->last_n;
    if (n_planes) *n_planes = s->last_k;
    if (rows) *rows = s->last_rows;
    if (cols) *cols = s->last_cols;
    return CVS_OK;
}

int cvs_batch_pyramid_setup(cvs_batch b, const cvs_plane* image, int rows, int cols, int levels, unsigned flags, int root,
                            cvs_batch_timing* timing)
{
    if (!b) return CVS_E_BADARG;
    if (rows <= 0 || cols <= 0 || levels < 1 || levels > 16) return fail(b, CVS_E_BADARG, "pyramid geometry");
    if (root < 0 || root >= b->world) return fail(b, CVS_E_BADARG, "root");
    if (b->world > 1 && b->transport == TRANSPORT_NONE) return fail(b, CVS_E_HIP, "no transport");
    if (!(flags & CVS_SETUP_BASIS)) flags |= CVS_SETUP_BASIS;
    Slot* rs = local(b, root);
    // root-only and per-rank failures are agreed on before anything is queued (see agree() / cvs_batch_run)
    int local_rc = CVS_OK;
    if (rs && !dense_device(image, rows, cols))
        local_rc = fail(b, CVS_E_SIZE, "the image must be a dense f32 device plane of rows x cols on the root");
    const int nb = cvs_num_basis(b->kind);
    const int nplanes = nb + ((flags & CVS_SETUP_ORIENT) ? 5 : 0);
    std::vector<int> lr(levels), lc(levels);
    lr[0] = rows;
    lc[0] = cols;
    for (int l = 1; l < levels; ++l) {
        lr[l] = (lr[l - 1] + 1) / 2;
        lc[l] = (lc[l - 1] + 1) / 2;
    }
    int rc;
    const size_t plane0 = (size_t)rows * cols;
    auto prepare = [&]() -> int {
        for (Slot& s : b->slots) {
            if (s.rank != root && (rc = reserve(b, s, s.image, plane0))) return rc;
            while ((int)s.level.size() < levels) {
                cvs_handle h = nullptr;
                rc = cvs_create(b->kind, b->width, b->spacing, s.device, &h);
                if (rc != CVS_OK) return fail(b, rc, "cvs_create (level handle)");
                B_CVS(b, h, cvs_set_stream(h, s.stream));
                // planar state: the bands of a plane are gathered as ONE contiguous message per plane and rank (below); in the
                // row-interleaved layout a band of one plane is not contiguous
                B_CVS(b, h, cvs_set_option(h, CVS_OPT_STATE_LAYOUT, 0));
                s.level.push_back(h);
                s.level_img.push_back(DevBuf());
            }
            for (int l = 1; l < levels; ++l)
                if ((rc = reserve(b, s, s.level_img[l], (size_t)lr[l] * lc[l]))) return rc;
        }
        return CVS_OK;
    };
    if (local_rc == CVS_OK) local_rc = prepare();
    if ((rc = agree(b, local_rc, hash_ints({rows, cols, levels, (long long)flags, root, 3})))) return rc;
    b->pyr_levels = levels;
    b->pyr_root = root;
    if ((rc = record_all(b, 0))) return rc;
    // ---- broadcast of the image: every rank reads its bands AND their halo rows from its own copy ----
    // CVS_BATCH_SELF_TRANSPORT=1 (tests): a one-rank world still issues the (degenerate) ncclBroadcast
    const bool self_bcast = b->world == 1 && b->transport == TRANSPORT_RCCL && std::getenv("CVS_BATCH_SELF_TRANSPORT");
    if (b->world > 1 || self_bcast) {
        if (b->transport == TRANSPORT_RCCL) {
            GroupGuard g(rccl());
            g.start();
            for (Slot& s : b->slots) {
                if (!g.ok()) break;
                g.hip(hipSetDevice(s.device), "hipSetDevice");
                const float* src = s.rank == root ? image->data : s.image.p;
                float* dst = s.rank == root ? image->data : s.image.p;
                if (g.ok()) g.nccl(g.R->Broadcast(src, dst, plane0, ncclFloat, root, s.comm, s.stream), "ncclBroadcast");
            }
            if ((rc = g.end(b))) return rc;
        } else {
            std::vector<Move> mv;
            for (Slot& s : b->slots)
                if (s.rank != root) mv.push_back({root, s.rank, image->data, s.image.p, plane0});
            if ((rc = run_moves(b, mv))) return rc;
        }
    }
    if ((rc = record_all(b, 1))) return rc;
    // ---- every rank: the whole pyramid (cheap, redundant), then its band of rows of every level ----
    for (Slot& s : b->slots) {
        const float* lvl0 = s.rank == root ? image->data : s.image.p;
        for (int l = 0; l < levels; ++l) {
            cvs_plane cur{l == 0 ? const_cast<float*>(lvl0) : s.level_img[l].p, lr[l], lc[l], (size_t)lc[l] * sizeof(float), CVS_MEM_DEVICE};
            if (l + 1 < levels) {
                cvs_plane nxt{s.level_img[l + 1].p, lr[l + 1], lc[l + 1], (size_t)lc[l + 1] * sizeof(float), CVS_MEM_DEVICE};
                B_CVS(b, s.level[l], cvs_pyr_down(s.level[l], &cur, &nxt));
            }
            int lo, hi;
            shard_range(lr[l], b->world, s.rank, &lo, &hi);
            if (hi > lo) {
                B_CVS(b, s.level[l], cvs_setup_rows(s.level[l], &cur, flags, lo, hi));
            } else if (s.rank == root) {
                // an empty band on the root (a level with fewer rows than ranks): its state planes must still exist at
                // full size, because the other ranks' bands are received into them -- row 0 is rewritten by the gather
                B_CVS(b, s.level[l], cvs_setup_rows(s.level[l], &cur, flags, 0, 1));
            }
        }
    }
    if ((rc = record_all(b, 2))) return rc;
    // ---- gather of the bands into the root's state planes (rows of a plane are contiguous: one message per plane) ----
    if (b->world > 1) {
        std::vector<Move> mv;
        for (int l = 0; l < levels; ++l) {
            for (int r = 0; r < b->world; ++r) {
                if (r == root) continue;
                int lo, hi;
                shard_range(lr[l], b->world, r, &lo, &hi);
                if (hi <= lo) continue;
                Slot* s = local(b, r);
                for (int p = 0; p < nplanes; ++p) {
                    const int which = p < nb ? CVS_PLANE_BASIS0 + p : CVS_PLANE_C1 + (p - nb);
                    cvs_plane sv{}, dv{};
                    if (s) B_CVS(b, s->level[l], cvs_state_plane(s->level[l], which, &sv));
                    if (rs) B_CVS(b, rs->level[l], cvs_state_plane(rs->level[l], which, &dv));
                    const size_t pitch = (s ? sv.step : dv.step) / sizeof(float);
                    if (pitch != ((size_t)lc[l] + 63) / 64 * 64) return fail(b, CVS_E_STATE, "band gather needs planar state planes (CVS_OPT_STATE_LAYOUT = 0)");
                    mv.push_back({r, root, s ? sv.data + (size_t)lo * pitch : nullptr, rs ? dv.data + (size_t)lo * pitch : nullptr,
                                  (size_t)(hi - lo) * pitch});
                }
            }
        }
        if ((rc = run_moves(b, mv))) return rc;
    }
    if ((rc = record_all(b, 3))) return rc;
    if ((rc = sync_all(b))) return rc;
    fill_timing(b, timing);
    return CVS_OK;
}

int cvs_batch_level(cvs_batch b, int level, cvs_handle* h, cvs_plane* level_image)
{
    if (!b || !h) return CVS_E_BADARG;
    Slot* rs = local(b, b->pyr_root);
    if (!rs) return fail(b, CVS_E_STATE, "the gathered state lives on the root rank, which is not local to this process");
    if (level < 0 || level >= b->pyr_levels) return fail(b, CVS_E_BADARG, "level");
    *h = rs->level[level];
    if (level_image) {
        int r = 0, c = 0;
        (void)cvs_shape(*h, &r, &c);
        *level_image = cvs_plane{level == 0 ? nullptr : rs->level_img[level].p, r, c, (size_t)c * sizeof(float), CVS_MEM_DEVICE};
    }
    return CVS_OK;
}

}  // extern "C"
