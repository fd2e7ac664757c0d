// cvs_handle.cpp -- what every entry point of the C ABI needs around a kernel launch: argument checks (sizes, steps,
// overlaps), the per-call staging arena for host planes, and the handle's state block -- its layout (ensure_state) and the
// process-wide cache of released blocks (the reference builds one object per image, example/steer.cpp:86).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <set>
#include <tuple>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "cvs_context.h"

namespace cvs {

// CVS_OPTS="autotune=0,layout=1,pyr_strip=1,batch_ways=2,warm=4,wgcap=3,nt_stores=1,lit=0,verbose=1,pool_mb=4096": the one documented
// environment hook (A/B aids for new handles; unknown names are reported once on stderr and ignored)
EnvOpts env_opts()
{
    static std::mutex mu;
    static std::string cached_text;
    static EnvOpts cached;
    static bool have = false;
    const char* e = std::getenv("CVS_OPTS");
    const std::string text = e ? e : "";
    std::lock_guard<std::mutex> lock(mu);
    if (have && text == cached_text) return cached;   // parsed again only when the variable has changed (tests do that)
    EnvOpts v;
    size_t pos = 0;
    while (pos < text.size()) {
        size_t end = text.find(',', pos);
        if (end == std::string::npos) end = text.size();
        const std::string item = text.substr(pos, end - pos);
        pos = end + 1;
        const size_t eq = item.find('=');
        if (eq == std::string::npos) continue;
        const std::string name = item.substr(0, eq);
        const long val = std::atol(item.c_str() + eq + 1);
        if (name == "autotune") v.autotune = val != 0;
        else if (name == "layout") v.layout = (int)std::max(0L, std::min(3L, val));
        else if (name == "pyr_strip") v.pyr_strip = val != 0;
        else if (name == "batch_ways") v.batch_ways = (int)std::max(1L, val);
        else if (name == "nt_stores") v.nt_stores = val != 0;
        else if (name == "warm") v.warm = (int)std::max(0L, std::min(16L, val));
        else if (name == "wgcap") v.wgcap = (int)std::max(0L, std::min(8L, val));
        else if (name == "lit") v.lit = val != 0;
        else if (name == "verbose") v.verbose = val != 0;
        else if (name == "pool_mb") v.pool_mb = val;
        else std::fprintf(stderr, "[cvsteer] CVS_OPTS: unknown name '%s' ignored\n", name.c_str());
    }
    cached = v;
    cached_text = text;
    have = true;
    return v;
}

int fail(cvs_handle h, int code, const char* what)
{
    if (h) h->err = what;
    return code;
}

int fail_hip(cvs_handle h, hipError_t e, const char* where)
{
    if (h) h->err = std::string(where) + ": " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? CVS_E_NOMEM : CVS_E_HIP;
}



int check_plane(cvs_handle h, const cvs_plane* p, const char* name, bool allow_u8)
{
    if (!p) return fail(h, CVS_E_BADARG, name);
    if (p->rows <= 0 || p->cols <= 0) return fail(h, CVS_E_SIZE, "empty plane");
    if (!p->data) return fail(h, CVS_E_BADARG, name);
    if ((p->mem & ~(0xff | CVS_DEPTH_U8)) || (is_u8(p) && !allow_u8)) return fail(h, CVS_E_BADARG, "bad mem / depth flags");
    if (mem_of(p) != CVS_MEM_HOST && mem_of(p) != CVS_MEM_DEVICE) return fail(h, CVS_E_BADARG, "bad mem kind");
    if (is_u8(p)) {
        if (p->step < (size_t)p->cols) return fail(h, CVS_E_SIZE, "bad step");
    } else if (p->step < (size_t)p->cols * sizeof(float) || p->step % sizeof(float)) {
        return fail(h, CVS_E_SIZE, "bad step");
    } else if (reinterpret_cast<uintptr_t>(p->data) % alignof(float)) {
        return fail(h, CVS_E_BADARG, "f32 plane not aligned to 4 bytes");   // a float* the C language itself does not allow
    }
    return CVS_OK;
}

// Do two planes share a byte?  Same kind of memory only (a host plane and a device plane never do).  Planes with the same row
// step are compared exactly -- two column ranges of one buffer side by side (ROI views) interleave in address space without
// sharing anything --, planes with different steps by their address ranges (conservative).
bool planes_overlap(const cvs_plane* a, const cvs_plane* b)
{
    if (!a || !b || !a->data || !b->data || mem_of(a) != mem_of(b)) return false;
    auto width = [](const cvs_plane* p) { return (size_t)p->cols * (is_u8(p) ? 1 : sizeof(float)); };
    auto extent = [&](const cvs_plane* p) { return (size_t)(p->rows - 1) * p->step + width(p); };
    const uintptr_t pa = reinterpret_cast<uintptr_t>(a->data), pb = reinterpret_cast<uintptr_t>(b->data);
    if (pa + extent(a) <= pb || pb + extent(b) <= pa) return false;
    if (a->step != b->step || a->step == 0) return true;
    const cvs_plane* lo = pa <= pb ? a : b;
    const cvs_plane* hi = pa <= pb ? b : a;
    const size_t d = (size_t)(reinterpret_cast<uintptr_t>(hi->data) - reinterpret_cast<uintptr_t>(lo->data));
    const size_t r = d / lo->step, c = d % lo->step;   // hi's first pixel sits at (row r, byte column c) of lo's frame
    return (r < (size_t)lo->rows && c < width(lo)) || (r + 1 < (size_t)lo->rows && c + width(hi) > lo->step);
}

// outputs of one call: none may share memory with the input it is computed from (the kernels read rows ahead of the rows
// they write, and the overlapped host path downloads results while later rows are still being uploaded), nor with another output
int check_no_overlap(cvs_handle h, const cvs_plane* input, const cvs_plane* const* outs, int n)
{
    for (int i = 0; i < n; ++i) {
        if (!outs[i] || !outs[i]->data) continue;
        if (input && planes_overlap(outs[i], input)) return fail(h, CVS_E_BADARG, "an output plane overlaps the input image");
        for (int j = i + 1; j < n; ++j)
            if (outs[j] && planes_overlap(outs[i], outs[j])) return fail(h, CVS_E_BADARG, "two output planes overlap each other");
    }
    return CVS_OK;
}

// per-pixel stages read a pixel and write the same pixel: an output may BE an input (same first pixel, same step -- the
// reference itself calls wrap(m_theta, m_theta)), but it may not overlap one in any other way, nor another output
int check_point_overlaps(cvs_handle h, std::initializer_list<const cvs_plane*> ins, std::initializer_list<const cvs_plane*> outs)
{
    for (auto o = outs.begin(); o != outs.end(); ++o) {
        if (!*o || !(*o)->data) continue;
        for (const cvs_plane* i : ins)
            if (i && planes_overlap(*o, i) && !((*o)->data == i->data && (*o)->step == i->step))
                return fail(h, CVS_E_BADARG, "an output plane overlaps an input plane without being it");
        for (auto q = o + 1; q != outs.end(); ++q)
            if (*q && planes_overlap(*o, *q)) return fail(h, CVS_E_BADARG, "two output planes overlap each other");
    }
    return CVS_OK;
}

int check_same(cvs_handle h, const cvs_plane* p, int rows, int cols)
{
    if (p->rows != rows || p->cols != cols) return fail(h, CVS_E_SIZE, "plane size mismatch");
    return CVS_OK;
}


int arena_reserve(cvs_handle h, size_t elems)
{
    if (elems <= h->arena_elems) return CVS_OK;
    if (h->arena) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, hipFree(h->arena));
        h->arena = nullptr;
        h->arena_elems = 0;
    }
    HIP_TRY(h, hipMalloc(&h->arena, elems * sizeof(float)));
    h->arena_elems = elems;
    return CVS_OK;
}

float* arena_take(cvs_handle h, size_t elems)
{
    float* p = h->arena + h->arena_used;
    h->arena_used += round_up(elems, 64);
    return p;
}

// bytes-as-floats of the device staging of an 8-bit HOST image (rows padded to 256 bytes)
size_t u8_stage_elems(const cvs_plane* p) { return round_up(round_up((size_t)p->cols, 256) * p->rows / 4 + 64, 64); }

size_t staged_elems(const cvs_plane* p)
{
    if (!p) return 0;
    const size_t plane = round_up(round_up((size_t)p->cols, 64) * p->rows, 64);
    if (is_u8(p))  // widened copy: f32 plane on the device, plus the byte image itself when it comes from the host
        return plane + (mem_of(p) == CVS_MEM_HOST ? u8_stage_elems(p) : 0);
    return mem_of(p) == CVS_MEM_HOST ? plane : 0;
}

// resolve an input plane to a device pointer (uploading host data)
int in_ref(Call& c, const cvs_plane* p, PlaneRef& r)
{
    cvs_handle h = c.h;
    if (is_u8(p) && c.u8_direct) {  // the strip kernel reads the bytes itself; r.pitch is then in BYTES (BasisArgs::in_u8)
        if (mem_of(p) == CVS_MEM_DEVICE) {
            r = {p->data, p->step};
            return CVS_OK;
        }
        const size_t bpitch = round_up((size_t)p->cols, 256);
        uint8_t* b = reinterpret_cast<uint8_t*>(arena_take(h, u8_stage_elems(p)));
        HIP_TRY(h, copy_rows(b, bpitch, p->data, p->step, (size_t)p->cols, p->rows, hipMemcpyHostToDevice, h->stream));
        c.touched_host = true;
        r = {reinterpret_cast<float*>(b), bpitch};
        return CVS_OK;
    }
    if (is_u8(p)) {  // 8-bit image on a path the strip kernels do not cover (generic widths, tiny images, pyramid emission):
                     // bytes cross PCIe, a widening pass makes the f32 plane on the device
        const size_t pitch = round_up((size_t)p->cols, 64);
        float* d = arena_take(h, pitch * p->rows);
        const uint8_t* src = reinterpret_cast<const uint8_t*>(p->data);
        size_t sstep = p->step;
        if (mem_of(p) == CVS_MEM_HOST) {
            const size_t bpitch = round_up((size_t)p->cols, 256);
            uint8_t* b = reinterpret_cast<uint8_t*>(arena_take(h, round_up(bpitch * p->rows / 4 + 64, 64)));
            if (c.defer) {
                c.deferred_image = p;
                c.deferred_u8 = b;
                c.deferred_u8_pitch = bpitch;
                r = {d, pitch};
                return CVS_OK;
            }
            HIP_TRY(h, copy_rows(b, bpitch, p->data, p->step, (size_t)p->cols, p->rows, hipMemcpyHostToDevice, h->stream));
            c.touched_host = true;
            src = b;
            sstep = bpitch;
        }
        HIP_TRY(h, launch_u8_to_f32(src, sstep, p->rows, p->cols, d, pitch, h->stream));
        r = {d, pitch};
        return CVS_OK;
    }
    if (p->mem == CVS_MEM_DEVICE) {
        r = {p->data, p->step / sizeof(float)};
        return CVS_OK;
    }
    const size_t pitch = round_up((size_t)p->cols, 64);
    float* d = arena_take(h, pitch * p->rows);
    if (c.defer) {
        c.deferred_image = p;
        r = {d, pitch};
        return CVS_OK;
    }
    HIP_TRY(h, copy_rows(d, pitch * sizeof(float), p->data, p->step, (size_t)p->cols * sizeof(float), p->rows,
                                hipMemcpyHostToDevice, h->stream));
    c.touched_host = true;
    r = {d, pitch};
    return CVS_OK;
}

int out_ref(Call& c, const cvs_plane* p, PlaneRef& r)
{
    cvs_handle h = c.h;
    if (!p) {
        r = {nullptr, 0};
        return CVS_OK;
    }
    if (p->mem == CVS_MEM_DEVICE) {
        r = {p->data, p->step / sizeof(float)};
        return CVS_OK;
    }
    const size_t pitch = round_up((size_t)p->cols, 64);
    float* d = arena_take(h, pitch * p->rows);
    c.outs.push_back({p, d, pitch});
    r = {d, pitch};
    return CVS_OK;
}

// download pending host outputs; host-touching calls return with the data landed
int finish(Call& c)
{
    cvs_handle h = c.h;
    for (const Pending& o : c.outs) {
        HIP_TRY(h, copy_rows(o.host->data, o.host->step, o.dev, o.pitch * sizeof(float),
                                    (size_t)o.host->cols * sizeof(float), o.host->rows, hipMemcpyDeviceToHost, h->stream));
        c.touched_host = true;
    }
    if (c.touched_host) HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CVS_OK;
}

int begin(cvs_handle h, Call& c, std::initializer_list<const cvs_plane*> planes, size_t extra)
{
    c.h = h;
    HIP_TRY(h, hipSetDevice(h->device));
    h->used = true;
    size_t need = extra;
    for (const cvs_plane* p : planes) need += staged_elems(p);
    if (need) {
        int rc = arena_reserve(h, need);
        if (rc) return rc;
    }
    h->arena_used = 0;
    return CVS_OK;
}

const cvs_context::PlaneGroup& state_group(cvs_handle h, int idx)
{
    for (int g = 0; g + 1 < h->ngrp; ++g)
        if (idx < h->grp[g].first + h->grp[g].count) return h->grp[g];
    return h->grp[h->ngrp > 0 ? h->ngrp - 1 : 0];
}

float* state_plane(cvs_handle h, int idx)
{
    const cvs_context::PlaneGroup& g = state_group(h, idx);
    return h->state + (size_t)h->cur_frame * h->frame_stride + g.off + (size_t)(idx - g.first) * g.stride;
}

// a state plane with the row pitch of its group
PlaneRef state_ref(cvs_handle h, int idx) { return {state_plane(h, idx), state_group(h, idx).pitch}; }

void fill_state_args(cvs_handle h, BasisArgs& a, bool orient)
{
    const cvs_context::PlaneGroup& g0 = h->grp[0];
    a.basis = state_plane(h, 0);
    a.pitch = g0.pitch;
    a.plane_stride = g0.stride;
    if (h->kind == CVS_KIND_G4) {
        const cvs_context::PlaneGroup& g1 = h->grp[1];
        a.basis2 = state_plane(h, g1.first);
        a.pitch2 = g1.pitch;
        a.plane_stride2 = g1.stride;
    }
    const cvs_context::PlaneGroup& go = state_group(h, h->nb);   // the orientation planes' group (group 0 itself when merged)
    a.orient = orient ? state_plane(h, h->nb) : nullptr;
    a.orient_pitch = go.pitch;
    a.orient_stride = go.stride;
    a.state_bytes = h->frame_stride * sizeof(float);
    a.tile_ctr = h->sb.tile_ctr;   // queues of the dynamic launch order (nullptr: none, static orders only)
    a.tile_parity = &h->sb.ctr_parity;
    a.lit_taps = env_opts().lit != 0;   // (CVS_OPTS lit=0: A/B and tests of the instances that take their taps from the kernel arguments)
    a.lit_used = &h->last.literal_taps;
}

// Process-wide cache of released state blocks.  The reference's usage model is one short-lived object per image
// (example/steer.cpp:86 inside the parallel_for_ body; test/test.cpp:85): a hipMalloc + hipFree of the 0.8 GiB state
// block per image costs more than the filtering itself, so cvs_destroy parks the block here (after its stream has
// drained) and the next handle on the same device that needs a block of about that size takes it over.  Bounded:
// CVS_OPTS pool_mb megabytes in all (default 4096, 0 = off), blocks at most twice the size asked for;
// cvs_release_cached_memory() empties it.
static std::mutex g_pool_mutex;
static std::vector<StateBlock> g_pool;

size_t pool_limit_bytes()
{
    const long mb = env_opts().pool_mb;
    return mb > 0 ? (size_t)mb << 20 : (size_t)0;
}

// a block of about the size asked for
bool pool_take(int device, size_t elems, StateBlock& out)
{
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    int best = -1;
    for (int i = 0; i < (int)g_pool.size(); ++i) {
        const StateBlock& b = g_pool[i];
        if (b.device != device) continue;
        if (b.elems >= elems && b.elems <= 2 * elems && (best < 0 || b.elems < g_pool[best].elems)) best = i;
    }
    if (best < 0) return false;
    out = g_pool[best];
    g_pool.erase(g_pool.begin() + best);
    return true;
}

// blk.ready (if any) marks the end of the work that last used the block
void pool_give(StateBlock& blk)
{
    const size_t lim = pool_limit_bytes(), bytes = blk.elems * sizeof(float);
    std::vector<StateBlock> drop;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        if (bytes > lim) drop.push_back(blk);
        else {
            size_t held = bytes;
            for (const StateBlock& b : g_pool) held += b.elems * sizeof(float);
            while (held > lim && !g_pool.empty()) {  // oldest first
                held -= g_pool.front().elems * sizeof(float);
                drop.push_back(g_pool.front());
                g_pool.erase(g_pool.begin());
            }
            g_pool.push_back(blk);
        }
    }
    blk = StateBlock();
    for (StateBlock& d : drop) state_block_free(d);
}

// every parked block goes back to the allocator (cvs_release_cached_memory)
void pool_release_all()
{
    std::vector<StateBlock> blocks;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        blocks.swap(g_pool);
    }
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (StateBlock& b : blocks) {
        (void)hipSetDevice(b.device);
        state_block_free(b);
    }
    if (have_cur) (void)hipSetDevice(cur);
    (void)hipGetLastError();
}

// The handle lets go of its state block WITHOUT draining the device: an event recorded on its stream travels with the
// parked block, and the next taker's stream waits for it (ensure_state).  The reference's callers build one object per
// image (example/steer.cpp:86): object k+1's launch is queued while object k's is still running.
void release_state(cvs_handle h)
{
    if (!h->state) return;
    h->sb.ready = nullptr;
    if (h->used) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(h->stream, &cap);
        hipEvent_t ev = nullptr;
        if (cap == hipStreamCaptureStatusNone && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess &&
            hipEventRecord(ev, h->stream) == hipSuccess) {
            h->sb.ready = ev;
        } else {
            if (ev) (void)hipEventDestroy(ev);
            (void)hipGetLastError();
            if (cap == hipStreamCaptureStatusNone) (void)hipStreamSynchronize(h->stream);
        }
    }
    pool_give(h->sb);
    h->state = nullptr;
    h->state_elems = 0;
}

// Where the planes of the current geometry (h->rows, h->dense_pitch, h->layout_stride) lie inside a frame's block.
// merge_orient (G2, interleaved layout): the five orientation planes join the basis planes in ONE group of twelve -- a full
// setup or a pipeline launch then writes one sweep instead of two.  Two groups run the 12-plane launch at 0.755 or at 0.854 of
// the HBM roofline depending on where the allocator put the block, one group at 0.81-0.82 either way
// (profiles/r04_groups_probe.txt); rounds 4-5 let the tuner compare the two, since round 6 one group is what single-image
// launches that write the orientation planes use (cvs_tune.cpp default_config; profiles/r06_m4_layout_ab.txt); a basis-only
// launch, a frame batch and CVS_OPT_STATE_LAYOUT = 3 use the two-group form.
// twelve planes in one group must still lie within the 32-bit buffer offsets of one launch
bool state_merge_fits(cvs_handle h, int rows, size_t dense_pitch)
{
    return h->kind == CVS_KIND_G2 && (size_t)rows * dense_pitch * sizeof(float) * (size_t)(h->nb + 5) <= (size_t)0x7ffffff0;
}

void layout_state(cvs_handle h, bool merge_orient)
{
    const size_t pitch = h->dense_pitch, stride = h->layout_stride;
    const int rows = h->rows;
    const bool inter = state_interleaved(h, rows, pitch);
    // groups: G2 = 7 basis planes | 5 orientation planes; G4 = the 5 G planes | the 6 H planes | 5 orientation planes (the
    // half banks of the G4 pair launch write one group each, so each of them streams a dense sweep as well)
    int counts[3] = {h->kind == CVS_KIND_G4 ? 5 : h->nb, h->kind == CVS_KIND_G4 ? 6 : 5, 5};
    h->ngrp = h->kind == CVS_KIND_G4 ? 3 : 2;
    const bool merged = merge_orient && inter && h->kind == CVS_KIND_G2 && state_merge_fits(h, rows, pitch);
    if (merged) {
        counts[0] = h->nb + 5;
        h->ngrp = 1;
    }
    size_t off = 0;
    int first = 0;
    for (int g = 0; g < h->ngrp; ++g) {
        cvs_context::PlaneGroup& G = h->grp[g];
        G.first = first;
        G.count = counts[g];
        G.off = off;
        if (inter) {
            G.pitch = pitch * G.count;
            G.stride = pitch;
            off += round_up(pitch * rows * G.count, 64);   // <= stride * count: the block holds it
        } else {
            G.pitch = pitch;
            G.stride = stride;
            off += stride * G.count;
        }
        first += G.count;
    }
    h->frame_stride = off;
    h->last.state_layout = !inter ? 0 : merged ? 2 : 1;
}

// row-interleaved state planes (CVS_OPT_STATE_LAYOUT = 1, the default) while a whole group of planes stays below 2 GiB, i.e.
// within the 32-bit buffer offsets of one launch (larger states -- 8192^2 G4, 16384^2 G2 -- stay planar and are banded)
bool state_interleaved(cvs_handle h, int rows, size_t dense_pitch)
{
    return h->layout >= 1 && (size_t)rows * dense_pitch * sizeof(float) * (size_t)(h->kind == CVS_KIND_G4 ? 6 : 7) <= (size_t)0x7ffffff0;
}

int ensure_state(cvs_handle h, int rows, int cols, int nframes)
{
    const size_t pitch = round_up((size_t)cols, 64);
    const size_t stride = round_up(pitch * rows, 64);
    const int nplanes = h->nb + 5;
    const size_t elems = stride * nplanes * (size_t)nframes;
    if (!(h->state != nullptr && elems <= h->state_elems)) {
        release_state(h);   // parked, not freed: a handle that alternates between two geometries gets its blocks back
        const bool from_pool = pool_take(h->device, elems, h->sb);
        if (from_pool && h->sb.ready) {   // the previous owner's work on this block comes first
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(h->stream, &cap);
            RelaxedCapture relaxed;
            const hipError_t we = cap == hipStreamCaptureStatusNone ? hipStreamWaitEvent(h->stream, h->sb.ready, 0) : hipEventSynchronize(h->sb.ready);
            (void)hipEventDestroy(h->sb.ready);
            h->sb.ready = nullptr;
            if (we != hipSuccess) return fail_hip(h, we, "waiting for a parked state block");
        }
        if (!from_pool) HIP_TRY(h, state_block_alloc(h->device, elems, h->sb));
        h->state = h->sb.base;
        h->state_elems = h->sb.elems;
    }
    h->rows = rows;
    h->cols = cols;
    h->dense_pitch = pitch;
    // Layout of the planes inside the block.  Round 4 (tools/layout_probe.py, profiles/r04_layout_probe.txt): where the rows
    // of the planes lie relative to each other decides how fast a launch that writes 7..20 planes at once streams.  PLANAR
    // (plane after plane, rounds 1-3): a wave's stores of one output row go to addresses 64 MiB apart, one stream per plane.
    // ROW-INTERLEAVED (default): row r of all planes of a group lies side by side -- [row][plane][column] -- so the launch's
    // write frontier is ONE linear sweep through the block (per group), and every plane is still an ordinary strided view
    // (step = planes x row length), which is all the per-pixel kernels, cvs_state_plane and the facade ever ask for.  Two
    // groups, basis and orientation, so that a basis-only setup writes a dense stream too.  On plain blocks, same handles
    // side by side: basis 0.76 -> 0.80, fused steer 0.70 -> 0.80, full setup 0.65 -> 0.82, pipeline 0.66 -> 0.73 of the HBM
    // roofline, fresh images +4-5 points.
    h->layout_stride = stride;
    layout_state(h, false);
    h->num_frames = nframes;
    if (h->cur_frame >= nframes) h->cur_frame = 0;
    return CVS_OK;
}

}  // namespace cvs
