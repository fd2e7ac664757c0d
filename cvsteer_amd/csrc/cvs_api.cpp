// cvs_api.cpp -- the C ABI of libcvsteer_hip.so (declared in include/cvsteer_hip.h).
//
// Thin by design: argument checks, device-state ownership, host<->device staging when a
// caller hands over host planes, and kernel dispatch.  All arithmetic on the hot path is in
// the HIP kernels (cvs_kernels_basis.hip, cvs_kernels_point.hip); the only host arithmetic is
// what the reference also does on the host before it touches an image: tap generation and the
// scalar steering weights of a given theta (cvs_taps.cpp).  No CPU fallback exists.
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

using namespace cvs;

namespace {

int do_setup(cvs_handle h, const cvs_plane* image, unsigned flags, bool steer, float theta, const cvs_plane* g,
             const cvs_plane* hq, const cvs_plane* const* pipe_outs = nullptr, int nframes = 1, int frame = 0,
             int out_row_lo = 0, int out_row_hi = 0, const cvs_plane* pyr = nullptr)
{
    if (!h) return CVS_E_BADARG;
    int rc = check_plane(h, image, "image", true);
    if (rc) return rc;
    if (pyr) {  // the next pyramid level, written by the same pass (cvs_setup_pyr)
        if ((rc = check_plane(h, pyr, "next_level")) || (rc = check_same(h, pyr, (image->rows + 1) / 2, (image->cols + 1) / 2))) return rc;
    }
    if (flags & ~(unsigned)CVS_SETUP_FULL) return fail(h, CVS_E_BADARG, "unknown setup flags");
    if (!(flags & CVS_SETUP_BASIS)) flags |= CVS_SETUP_BASIS;
    if ((flags & CVS_SETUP_ORIENT) && h->kind != CVS_KIND_G2 && !h->g4_ext)
        return fail(h, CVS_E_UNSUPPORTED, "the reference computes no orientation for G4 (G4.cpp:67-81); see CVS_OPT_G4_EXTENSIONS");
    if (steer) {
        if ((rc = check_plane(h, g, "g")) || (rc = check_plane(h, hq, "hq"))) return rc;
        if ((rc = check_same(h, g, image->rows, image->cols)) || (rc = check_same(h, hq, image->rows, image->cols))) return rc;
    }
    // the kernel reads rows ahead of the rows it writes: an output that shares memory with the input would be
    // clobbered mid-flight (the reference's sepFilter2D copies in that case; here it is an error)
    {
        const cvs_plane* outs_chk[11] = {steer ? g : nullptr, steer ? hq : nullptr};
        for (int k = 0; k < 8; ++k) outs_chk[2 + k] = pipe_outs ? pipe_outs[k] : nullptr;
        outs_chk[10] = pyr;
        if ((rc = check_no_overlap(h, image, outs_chk, 11))) return rc;
    }
    const size_t pitch = round_up((size_t)image->cols, 64);
    size_t max_pitch = std::max(pitch, is_u8(image) ? pitch : image->step / sizeof(float));
    if (steer) max_pitch = std::max(max_pitch, std::max(g->step, hq->step) / sizeof(float));
    if (pipe_outs)
        for (int k = 0; k < 8; ++k)
            if (pipe_outs[k]) max_pitch = std::max(max_pitch, pipe_outs[k]->step / sizeof(float));
    if (state_interleaved(h, image->rows, pitch)) max_pitch = std::max(max_pitch, pitch * (size_t)(h->kind == CVS_KIND_G4 ? 6 : 7));
    const bool may_generic = basis_may_need_scratch(h->kind, h->width, h->taps, image->rows, image->cols, max_pitch);
    const size_t scratch = may_generic ? round_up(basis_scratch_elems(h->kind, h->width, image->rows, pitch), 64) : 0;
    Call c;
    const cvs_plane* po[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (pipe_outs)
        for (int k = 0; k < 8; ++k) po[k] = pipe_outs[k];
    // 8-bit images (what the reference's callers hold: test/test.cpp:73,85, example/steer.cpp:73-86) are read by the strip kernel
    // as bytes: 1 B/pix of input traffic instead of 1 B read + 4 B written by a widening pass + 4 B read
    c.u8_direct = is_u8(image) && !may_generic && !pyr;
    const size_t u8_stage = (c.u8_direct && mem_of(image) == CVS_MEM_HOST) ? u8_stage_elems(image) : 0;
    rc = begin(h, c, {c.u8_direct ? nullptr : image, steer ? g : nullptr, steer ? hq : nullptr, po[0], po[1], po[2], po[3], po[4], po[5], po[6], po[7], pyr},
               scratch + u8_stage);
    if (rc) return rc;
    // host planes on the fast path of a large enough image: upload, filtering and download overlap band by band
    // (it pays when a sizeable upload can hide behind the downloads: an f32 host image with host outputs -- measured
    // 3.06 vs 3.70 ms per 4096^2 image; with an 8-bit or device image the downloads alone set the pace and the bands
    // only add per-copy overhead, 2.98 vs 2.85 ms: round 2, profiles/HISTORY_rounds_1_3.md)
    bool any_host = false;
    for (const cvs_plane* o : {steer ? g : nullptr, steer ? hq : nullptr, po[0], po[1], po[2], po[3], po[4], po[5], po[6], po[7]})
        any_host = any_host || (o && o->mem == CVS_MEM_HOST);
    any_host = any_host && mem_of(image) == CVS_MEM_HOST && !is_u8(image);
    hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(h->stream, &cap_st);
    const bool overlap = h->host_overlap && any_host && !may_generic && nframes == 1 && out_row_hi <= out_row_lo && !pyr &&
                         cap_st == hipStreamCaptureStatusNone && (size_t)image->rows * image->cols >= ((size_t)1 << 20) &&
                         image->rows >= 16 * (2 * h->width + 1) && !(h->kind == CVS_KIND_G4 && (flags & CVS_SETUP_ORIENT));
    c.defer = overlap;
    h->have_basis = h->have_orient = false;
    if ((rc = ensure_state(h, image->rows, image->cols, nframes))) return rc;
    h->cur_frame = frame;

    BasisArgs a{};
    PlaneRef in;
    if ((rc = in_ref(c, image, in))) return rc;
    a.in = in.p;
    a.in_pitch = in.pitch;
    a.in_u8 = c.u8_direct ? 1 : 0;
    a.rows = image->rows;
    a.cols = image->cols;
    fill_state_args(h, a, (flags & CVS_SETUP_ORIENT) && h->kind == CVS_KIND_G2);
    a.atan_mode = h->atan_mode;
    // a different input pointer than last time = a stream of fresh images (not resident in the Infinity Cache);
    // the pipeline variants keep the taller strips (round 1 shape sweep)
    // (a handle's FIRST call counts as a fresh image too: the reference's callers build one object per image,
    // example/steer.cpp:86 -- only a handle that is handed the same pointer again is re-filtering a resident image)
    const bool fresh = h->last_image != (const void*)image->data && !pipe_outs;
    h->last_image = image->data;
    a.strip_rows = default_strip_rows(h, a.rows, a.cols, fresh);
    // New images of 24 MiB and more: the waves of the launch's first fifth of row bands also touch the rest of the image (four
    // bands each), so that it is requested from HBM while the launch is young and most of the launch streams its writes without
    // reads mixed in.  Same process, alternating settings (profiles/r05_fresh_warm.txt): basis pass on alternating 8192^2 images
    // 0.626 -> 0.744 of the HBM roofline, fused steer on rotating 4096^2 images 0.624 -> 0.716, one object per image 0.614 ->
    // 0.700, full setup 0.624 -> 0.697; a separate read pass in front of the launch (round 4's tuner candidate, removed) reached
    // 0.708 where this reaches 0.729 and cannot serve an 8192^2 image at all.  Not for: the launch that also emits the next pyramid
    // level (-8 %: its cached half-line stores want the L2 for themselves; with streaming stores for the level the launch gains
    // 7-13 % and the next level's launch, which then reads its image from HBM, loses more: config 3 0.658 -> 0.640, measured again in round 6,
    // profiles/r06_c3_chain_localisation.txt), 8-bit images of 4096^2 (-5 ... -6 %) and small images (level to -3 %).  G4 images are requested
    // ahead as well since round 6 (+2.2 ... +2.4 % on rotating 4096^2 images, profiles/r06_fresh_exact_wait.txt).  CVS_OPTS warm=K overrides K (0 = off).
    {
        const size_t in_bytes = (size_t)a.rows * a.cols * (a.in_u8 ? 1 : sizeof(float));
        const int wk = env_opts().warm;
        a.warm_k = (fresh && !pyr && in_bytes >= ((size_t)24 << 20)) ? (wk >= 0 ? wk : 4) : 0;
    }
    a.nt_stores = use_nt_stores(h, (size_t)a.rows * a.cols);
    a.diag = h->diag;
    a.out_row_lo = out_row_lo;
    a.out_row_hi = out_row_hi;
    if (steer) {
        PlaneRef rg, rh;
        if ((rc = out_ref(c, g, rg)) || (rc = out_ref(c, hq, rh))) return rc;
        a.steer_g = rg.p;
        a.steer_g_pitch = rg.pitch;
        a.steer_h = rh.p;
        a.steer_h_pitch = rh.pitch;
        host_steer_weights(h->kind, theta, a.steer_w);
    }
    if (pipe_outs) {
        a.pipe = 1;
        a.no_state = h->persist ? 0 : 1;
        a.find_on_e = h->find_on;
        for (int k = 0; k < 8; ++k)
            if ((rc = out_ref(c, po[k], a.pipe_out[k]))) return rc;
    }
    if (pyr) {
        PlaneRef rp;
        if ((rc = out_ref(c, pyr, rp))) return rc;
        a.pyr_out = rp.p;
        a.pyr_pitch = rp.pitch;
    }
    float* scr = scratch ? arena_take(h, scratch) : nullptr;
    if (overlap) {
        if ((rc = host_pipeline(h, c, a, scr))) return rc;
        h->have_basis = !(pipe_outs && !h->persist);
        h->have_orient = h->have_basis && (flags & CVS_SETUP_ORIENT) != 0;
        return CVS_OK;
    }
    {
        const bool orient_k = a.orient != nullptr;
        const int variant = (orient_k ? 1 : 0) | (steer ? 2 : 0) | (a.pipe ? 4 : 0) | (a.no_state ? 8 : 0);
        TuneToken tok;
        if ((rc = tune_begin(h, a, variant, fresh, tok))) return rc;
        if ((a.merge_orient != 0) != (h->ngrp == 1 && h->kind == CVS_KIND_G2)) {   // the configuration wants the other grouping of the planes
            layout_state(h, a.merge_orient != 0);
            fill_state_args(h, a, orient_k);
        }
        note_launch(h, a);
        const hipError_t le = launch_basis(h->kind, h->width, h->taps, a, scr, h->stream);
        tune_end(h, tok);
        HIP_TRY(h, le);
    }
    if ((flags & CVS_SETUP_ORIENT) && h->kind == CVS_KIND_G4) {  // extension: one per-pixel pass over the 11 planes
        PointArgs pa{};
        pa.rows = a.rows;
        pa.cols = a.cols;
        pa.atan_mode = h->atan_mode;
        pa.nt_stores = a.nt_stores;
        for (int p = 0; p < 11; ++p) pa.in[p] = state_ref(h, p);
        for (int i = 0; i < 5; ++i) pa.out[i] = state_ref(h, h->nb + i);
        HIP_TRY(h, launch_point(OP_G4_ORIENT, pa, h->stream));
    }
    // a pipeline run with CVS_OPT_PERSIST_STATE = 0 wrote its outputs only: no state to address afterwards
    h->have_basis = !(pipe_outs && !h->persist);
    h->have_orient = h->have_basis && (flags & CVS_SETUP_ORIENT) != 0;
    return finish(c);
}

void basis_inputs(cvs_handle h, PointArgs& a)
{
    for (int p = 0; p < h->nb; ++p) a.in[p] = state_ref(h, p);
}

int need_state(cvs_handle h, bool orient)
{
    if (!h) return CVS_E_BADARG;
    if (!h->have_basis) return fail(h, CVS_E_STATE, "no setup yet");
    if (orient && !h->have_orient) return fail(h, CVS_E_STATE, "orientation state not computed (setup without CVS_SETUP_ORIENT)");
    return CVS_OK;
}

int steer_common(cvs_handle h, bool map, float theta, const cvs_plane* theta_map, const cvs_plane* g, const cvs_plane* hq,
                 const cvs_plane* e, const cvs_plane* mag, const cvs_plane* phase)
{
    int rc = need_state(h, false);
    if (rc) return rc;
    if ((rc = check_plane(h, g, "g")) || (rc = check_plane(h, hq, "hq"))) return rc;
    const cvs_plane* all[6] = {g, hq, e, mag, phase, theta_map};
    for (const cvs_plane* p : all) {
        if (!p) continue;
        if ((rc = check_plane(h, p, "plane")) || (rc = check_same(h, p, h->rows, h->cols))) return rc;
    }
    if (h->kind == CVS_KIND_G4 && (e || mag || phase) && !h->g4_ext)
        return fail(h, CVS_E_UNSUPPORTED, "G4 has no energy / magnitude / phase in the reference (G4.cpp:88-90); see CVS_OPT_G4_EXTENSIONS");
    if (e && (rc = need_state(h, true))) return rc;
    if (map && !theta_map && (rc = need_state(h, true))) return rc;
    if ((rc = check_point_overlaps(h, {theta_map}, {g, hq, e, mag, phase}))) return rc;

    Call c;
    if ((rc = begin(h, c, {g, hq, e, mag, phase, theta_map}))) return rc;
    PointArgs a{};
    a.rows = h->rows;
    a.cols = h->cols;
    a.atan_mode = h->atan_mode;
    basis_inputs(h, a);
    const int nb = h->nb;
    if (e) {  // C1..C3 follow the basis planes: in[7..9] (G2) / in[11..13] (G4 extension)
        for (int i = 0; i < 3; ++i) a.in[nb + i] = state_ref(h, nb + i);
    }
    if (map) {
        PlaneRef th;
        if (theta_map) {
            if ((rc = in_ref(c, theta_map, th))) return rc;
        } else {
            th = state_ref(h, nb + 3);
        }
        a.in[h->kind == CVS_KIND_G2 ? 10 : 14] = th;
    } else {
        host_steer_weights(h->kind, theta, a.w);
        // G2.cpp:162: float c2t(std::cos(theta * 2.0)) -- double argument, narrowed
        a.c2t = (float)std::cos((double)theta * 2.0);
        a.s2t = (float)std::sin((double)theta * 2.0);
    }
    const cvs_plane* outs[5] = {g, hq, e, mag, phase};
    for (int o = 0; o < 5; ++o)
        if ((rc = out_ref(c, outs[o], a.out[o]))) return rc;
    PointOp op = h->kind == CVS_KIND_G2 ? (map ? OP_G2_STEER_MAP : OP_G2_STEER_SCALAR)
                                        : (map ? OP_G4_STEER_MAP : OP_G4_STEER_SCALAR);
    a.nt_stores = use_nt_stores(h, (size_t)a.rows * a.cols);
    a.nt_loads = a.nt_stores;  // the state planes of an image that large are not cache-resident and are read once here
    HIP_TRY(h, launch_point(op, a, h->stream));
    return finish(c);
}

}  // namespace

extern "C" {

int cvs_abi_version(void) { return CVS_ABI_VERSION; }

const char* cvs_status_string(int s)
{
    switch (s) {
        case CVS_OK: return "ok";
        case CVS_E_BADARG: return "bad argument";
        case CVS_E_SIZE: return "bad size";
        case CVS_E_HIP: return "HIP error";
        case CVS_E_NOMEM: return "out of memory";
        case CVS_E_STATE: return "state not available";
        case CVS_E_UNSUPPORTED: return "unsupported for this kind";
    }
    return "unknown status";
}

int cvs_num_basis(int kind) { return host_num_basis(kind); }

int cvs_make_taps(int kind, int idx, int width, float spacing, float* out)
{
    return host_make_taps(kind, idx, width, spacing, out) ? CVS_E_BADARG : CVS_OK;
}

int cvs_basis_taps(int kind, int p, int* kx, int* ky) { return host_basis_taps(kind, p, kx, ky) ? CVS_E_BADARG : CVS_OK; }

int cvs_steer_weights(int kind, float theta, float* out) { return host_steer_weights(kind, theta, out) ? CVS_E_BADARG : CVS_OK; }

int cvs_create(int kind, int width, float spacing, int device, cvs_handle* out)
{
    if (!out) return CVS_E_BADARG;
    *out = nullptr;
    const int nb = host_num_basis(kind);
    if (nb == 0 || width < 1 || width > kMaxWidth) return CVS_E_BADARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CVS_E_HIP;  // no CPU fallback
    if (device < 0 || device >= ndev) return CVS_E_BADARG;
    if (hipSetDevice(device) != hipSuccess) return CVS_E_HIP;
    cvs_context* h = new (std::nothrow) cvs_context();
    if (!h) return CVS_E_NOMEM;
    h->kind = kind;
    h->width = width;
    h->spacing = spacing;
    h->nb = nb;
    h->device = device;
    std::memset(h->taps, 0, sizeof(h->taps));
    for (int i = 0; i < nb; ++i) host_make_taps(kind, i, width, spacing, h->taps[i]);
    // (nothing is allocated on the device here: the reference's callers build one short-lived object per image,
    // example/steer.cpp:86, and a hipMalloc + hipFree pair per object costs ~20 us of the ~150 us such an object lives;
    // the 8 bytes of min / max scratch are allocated by the first 8-bit conversion that needs them)
    const EnvOpts eo = env_opts();   // CVS_OPTS: process-wide A/B overrides for new handles
    if (eo.autotune >= 0) h->autotune = eo.autotune;
    if (eo.layout >= 0) h->layout = eo.layout;
    if (eo.pyr_strip >= 0) h->pyr_strip = eo.pyr_strip;
    *out = h;
    return CVS_OK;
}

int cvs_destroy(cvs_handle h)
{
    if (!h) return CVS_E_BADARG;
    (void)hipSetDevice(h->device);
    release_state(h);   // no drain: the block is parked with an event
    // staging memory exists only on handles that were given host planes, 8-bit conversions or irregular batches: those wait
    if (h->arena || h->frame_tab || h->point_out) (void)hipStreamSynchronize(h->stream);
    if (h->arena) (void)hipFree(h->arena);
    if (h->frame_tab) (void)hipFree(h->frame_tab);
    if (h->point_out) (void)hipFree(h->point_out);
    if (h->ev_order) (void)hipEventDestroy(h->ev_order);
    for (hipEvent_t e : h->band_ev) (void)hipEventDestroy(e);
    if (h->s_up) (void)hipStreamDestroy(h->s_up);
    if (h->s_down) (void)hipStreamDestroy(h->s_down);
    delete h;
    return CVS_OK;
}

int cvs_release_cached_memory(void)
{
    pool_release_all();
    return CVS_OK;
}

const char* cvs_last_error(cvs_handle h) { return h ? h->err.c_str() : "null handle"; }

int cvs_set_stream(cvs_handle h, void* s)
{
    if (!h) return CVS_E_BADARG;
    hipStream_t ns = static_cast<hipStream_t>(s);
    if (ns == h->stream) return CVS_OK;
    // The handle's state block, staging arena and frame table are reused from call to call: work already queued
    // on the old stream must finish before the new stream touches them.  One event, recorded on the old stream and
    // waited for by the new one (no host synchronisation); skipped while either stream is being captured.
    hipStreamCaptureStatus c0 = hipStreamCaptureStatusNone, c1 = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(h->stream, &c0);
    (void)hipStreamIsCapturing(ns, &c1);
    (void)hipGetLastError();
    if (h->used && c0 == hipStreamCaptureStatusNone && c1 == hipStreamCaptureStatusNone) {
        HIP_TRY(h, hipSetDevice(h->device));
        if (!h->ev_order) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_order, hipEventDisableTiming));
        HIP_TRY(h, hipEventRecord(h->ev_order, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(ns, h->ev_order, 0));
    }
    h->stream = ns;
    return CVS_OK;
}

int cvs_set_option(cvs_handle h, int option, int value)
{
    if (!h) return CVS_E_BADARG;
    switch (option) {
        case CVS_OPT_ATAN_MODE:
            if (value != 0 && value != 1) return fail(h, CVS_E_BADARG, "atan mode");
            h->atan_mode = value;
            return CVS_OK;
        case CVS_OPT_STRIP_ROWS:
            if (value < 0 || value > 1 << 20) return fail(h, CVS_E_BADARG, "strip rows");
            h->strip_rows = value;
            return CVS_OK;
        case CVS_OPT_FIND_ON:
            if (value != 0 && value != 1) return fail(h, CVS_E_BADARG, "find_on");
            h->find_on = value;
            return CVS_OK;
        case CVS_OPT_G4_EXTENSIONS:
            if (value != 0 && value != 1) return fail(h, CVS_E_BADARG, "g4 extensions");
            h->g4_ext = value;
            return CVS_OK;
        case CVS_OPT_PERSIST_STATE:
            if (value != 0 && value != 1) return fail(h, CVS_E_BADARG, "persist");
            h->persist = value;
            return CVS_OK;
        case CVS_OPT_AUTOTUNE:
            if (value != 0 && value != 1) return fail(h, CVS_E_BADARG, "autotune");
            h->autotune = value;
            return CVS_OK;
        case CVS_OPT_BLOCK_ORDER:
            if (value != -1 && value != 0 && value != kOrderXcdColumns && value != kOrderDynamic) return fail(h, CVS_E_BADARG, "block order");
            h->block_order = value;
            return CVS_OK;
        case CVS_OPT_HOST_OVERLAP:
            if (value != 0 && value != 1) return fail(h, CVS_E_BADARG, "host overlap");
            h->host_overlap = value;
            return CVS_OK;
        case CVS_OPT_STATE_LAYOUT:
            if (value < 0 || value > 3) return fail(h, CVS_E_BADARG, "state layout");
            h->layout = value;
            return CVS_OK;
    }
    return fail(h, CVS_E_BADARG, "unknown option");
}

int cvs_get_option(cvs_handle h, int option, int* value)
{
    if (!h || !value) return CVS_E_BADARG;
    switch (option) {
        case CVS_OPT_ATAN_MODE: *value = h->atan_mode; return CVS_OK;
        case CVS_OPT_STRIP_ROWS: *value = h->strip_rows; return CVS_OK;
        case CVS_OPT_FIND_ON: *value = h->find_on; return CVS_OK;
        case CVS_OPT_BLOCK_ORDER: *value = h->block_order; return CVS_OK;
        case CVS_OPT_HOST_OVERLAP: *value = h->host_overlap; return CVS_OK;
        case CVS_OPT_AUTOTUNE: *value = h->autotune; return CVS_OK;
        case CVS_OPT_PERSIST_STATE: *value = h->persist; return CVS_OK;
        case CVS_OPT_G4_EXTENSIONS: *value = h->g4_ext; return CVS_OK;
        case CVS_OPT_STATE_LAYOUT: *value = h->layout; return CVS_OK;
    }
    return fail(h, CVS_E_BADARG, "unknown option");
}

int cvs_get_launch_info(cvs_handle h, cvs_launch_info* out)
{
    if (!h || !out) return CVS_E_BADARG;
    const uint32_t n = out->struct_size;   // the caller's sizeof(cvs_launch_info): never write beyond it
    if (n < sizeof(uint32_t)) return fail(h, CVS_E_BADARG, "cvs_launch_info.struct_size not set");
    cvs_launch_info li = h->last;
    li.struct_size = (uint32_t)std::min<size_t>(n, sizeof(li));
    std::memcpy(out, &li, li.struct_size);
    return CVS_OK;
}

int cvs_taps(cvs_handle h, int idx, float* out)
{
    if (!h || !out || idx < 0 || idx >= h->nb) return CVS_E_BADARG;
    std::memcpy(out, h->taps[idx], (2 * h->width + 1) * sizeof(float));
    return CVS_OK;
}

int cvs_kind(cvs_handle h, int* kind, int* width, float* spacing)
{
    if (!h) return CVS_E_BADARG;
    if (kind) *kind = h->kind;
    if (width) *width = h->width;
    if (spacing) *spacing = h->spacing;
    return CVS_OK;
}

int cvs_shape(cvs_handle h, int* rows, int* cols)
{
    if (!h) return CVS_E_BADARG;
    if (rows) *rows = h->have_basis ? h->rows : 0;
    if (cols) *cols = h->have_basis ? h->cols : 0;
    return CVS_OK;
}

int cvs_sync(cvs_handle h)
{
    if (!h) return CVS_E_BADARG;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CVS_OK;
}

int cvs_setup(cvs_handle h, const cvs_plane* image, unsigned flags)
{
    return do_setup(h, image, flags, false, 0.f, nullptr, nullptr);
}

int cvs_setup_steer(cvs_handle h, const cvs_plane* image, unsigned flags, float theta, const cvs_plane* g, const cvs_plane* hq)
{
    return do_setup(h, image, flags, true, theta, g, hq);
}

int cvs_setup_pyr(cvs_handle h, const cvs_plane* image, unsigned flags, const cvs_plane* next_level)
{
    if (!h) return CVS_E_BADARG;
    if (!next_level) return fail(h, CVS_E_BADARG, "next_level");
    return do_setup(h, image, flags, false, 0.f, nullptr, nullptr, nullptr, 1, 0, 0, 0, next_level);
}

// BASELINE config 3 in one call: filter every level of a Gaussian pyramid and build the pyramid on the way -- the filter launch
// of level k writes level k + 1 (cvs_setup_pyr: every level image is read once), all on the handles' stream.
// (Measured and NOT done, round 4: filtering the three small levels of a 5-level pyramid of 8192^2 concurrently on side streams
// behind the 4096^2 level -- 57 us of launch-latency-bound launches that could shrink to the longest of them.  The events
// that fork and join the streams cost more than the overlap saves: 0.603 ms against 0.525 ms for the plain chain, same box
// and process.  Nor: ONE launch for the three small levels (grid.z = level, per-level geometry) behind two stand-alone pyrDown
// launches that make their images first: 0.560 ms against 0.545 ms -- what the merged launch saves, the two extra launches
// and the loss of the fused level emission cost again.)
int cvs_pyramid_setup(cvs_handle* hs, int levels, const cvs_plane* image, unsigned flags, const cvs_plane* level_images)
{
    if (!hs || levels < 1 || !hs[0]) return CVS_E_BADARG;
    cvs_handle h0 = hs[0];
    if (!image || (levels > 1 && !level_images)) return fail(h0, CVS_E_BADARG, "image / level_images");
    for (int l = 0; l < levels; ++l) {
        if (!hs[l]) return fail(h0, CVS_E_BADARG, "null level handle");
        if (hs[l]->stream != h0->stream || hs[l]->device != h0->device) return fail(h0, CVS_E_BADARG, "the level handles must share one device and one stream");
        for (int m = 0; m < l; ++m)
            if (hs[m] == hs[l]) return fail(h0, CVS_E_BADARG, "one handle per level");
    }
    auto level_src = [&](int l) { return l == 0 ? image : &level_images[l - 1]; };
    int rc;
    for (int l = 0; l + 1 < levels; ++l) {
        const cvs_plane* s = level_src(l);
        if ((rc = check_plane(h0, &level_images[l], "level image")) || (rc = check_same(h0, &level_images[l], (s->rows + 1) / 2, (s->cols + 1) / 2))) return rc;
    }
    for (int l = 0; l < levels; ++l) {
        rc = do_setup(hs[l], level_src(l), flags, false, 0.f, nullptr, nullptr, nullptr, 1, 0, 0, 0, l + 1 < levels ? &level_images[l] : nullptr);
        if (rc) {
            if (hs[l] != h0) h0->err = hs[l]->err;
            return rc;
        }
    }
    return CVS_OK;
}

int cvs_setup_rows(cvs_handle h, const cvs_plane* image, unsigned flags, int row_lo, int row_hi)
{
    if (!h) return CVS_E_BADARG;
    if (!image || row_lo < 0 || row_hi > image->rows || row_lo >= row_hi) return fail(h, CVS_E_BADARG, "row range");
    if ((flags & CVS_SETUP_ORIENT) && h->kind == CVS_KIND_G4) return fail(h, CVS_E_UNSUPPORTED, "row ranges cover the basis planes only for G4");
    return do_setup(h, image, flags, false, 0.f, nullptr, nullptr, nullptr, 1, 0, row_lo, row_hi);
}

static int state_index(cvs_handle h, int which)
{
    if (which >= CVS_PLANE_BASIS0 && which < CVS_PLANE_BASIS0 + h->nb) return which - CVS_PLANE_BASIS0;
    if (which >= CVS_PLANE_C1 && which <= CVS_PLANE_STRENGTH) return h->nb + (which - CVS_PLANE_C1);
    return -1;
}

int cvs_state_plane(cvs_handle h, int which, cvs_plane* view)
{
    if (!h || !view) return CVS_E_BADARG;
    const int idx = state_index(h, which);
    if (idx < 0) return fail(h, CVS_E_BADARG, "unknown state plane");
    int rc = need_state(h, idx >= h->nb);
    if (rc) return rc;
    view->data = state_plane(h, idx);
    view->rows = h->rows;
    view->cols = h->cols;
    view->step = state_group(h, idx).pitch * sizeof(float);
    view->mem = CVS_MEM_DEVICE;
    return CVS_OK;
}

int cvs_read_state(cvs_handle h, int which, const cvs_plane* dst)
{
    cvs_plane src;
    int rc = cvs_state_plane(h, which, &src);
    if (rc) return rc;
    if ((rc = check_plane(h, dst, "dst")) || (rc = check_same(h, dst, h->rows, h->cols))) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t width = (size_t)h->cols * sizeof(float);
    if (dst->mem == CVS_MEM_HOST && src.step != width && dst->step == width) {
        // a plane of a row-interleaved group on its way to a dense host plane: over the host link a pitched 2-D copy is served
        // row by row at a fraction of the rate of a linear one (cvs_context.h copy_rows), so the rows are gathered on the
        // device first (a device-to-device 2-D copy runs at memory speed) and cross the link as one linear copy
        const size_t elems = (size_t)h->rows * h->cols;
        if ((rc = arena_reserve(h, round_up(elems, 64)))) return rc;
        HIP_TRY(h, hipMemcpy2DAsync(h->arena, width, src.data, src.step, width, h->rows, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(dst->data, h->arena, width * h->rows, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return CVS_OK;
    }
    HIP_TRY(h, copy_rows(dst->data, dst->step, src.data, src.step, width, h->rows,
                                dst->mem == CVS_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, h->stream));
    if (dst->mem == CVS_MEM_HOST) HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CVS_OK;
}

int cvs_steer_scalar(cvs_handle h, float theta, const cvs_plane* g, const cvs_plane* hq, const cvs_plane* e,
                     const cvs_plane* mag, const cvs_plane* phase)
{
    if (!h) return CVS_E_BADARG;
    return steer_common(h, false, theta, nullptr, g, hq, e, mag, phase);
}

int cvs_steer_map(cvs_handle h, const cvs_plane* theta, const cvs_plane* g, const cvs_plane* hq, const cvs_plane* e,
                  const cvs_plane* mag, const cvs_plane* phase)
{
    if (!h) return CVS_E_BADARG;
    return steer_common(h, true, 0.f, theta, g, hq, e, mag, phase);
}

int cvs_steer_point(cvs_handle h, int x, int y, float theta, float out[5])
{
    if (!h || !out) return CVS_E_BADARG;
    if (h->kind != CVS_KIND_G2) return fail(h, CVS_E_UNSUPPORTED, "point steer exists for G2 only (G2.cpp:115-134)");
    int rc = need_state(h, false);
    if (rc) return rc;
    if (x < 0 || y < 0 || x >= h->cols || y >= h->rows) return fail(h, CVS_E_BADARG, "point outside the image");
    HIP_TRY(h, hipSetDevice(h->device));
    // the scalar weights are host math in the reference too (G2.cpp:118-120); the pixel arithmetic
    // runs on the device, next to the state it reads
    PointArgs a{};
    host_steer_weights(CVS_KIND_G2, theta, a.w);
    a.c2t = (float)std::cos((double)theta * 2.0);  // G2.cpp:132: std::cos(theta * 2.0), double argument
    a.s2t = (float)std::sin((double)theta * 2.0);
    if (!h->point_out) HIP_TRY(h, hipMalloc(&h->point_out, 8 * sizeof(float)));
    const cvs_context::PlaneGroup &gb = state_group(h, 0), &go = state_group(h, h->nb);
    HIP_TRY(h, launch_steer_point(state_plane(h, 0), gb.stride, (size_t)y * gb.pitch + x, h->have_orient ? state_plane(h, h->nb) : nullptr,
                                  go.stride, (size_t)y * go.pitch + x, a, h->point_out, h->stream));
    HIP_TRY(h, hipMemcpyAsync(out, h->point_out, 5 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CVS_OK;
}

int cvs_mag_phase(cvs_handle h, const cvs_plane* g, const cvs_plane* hq, const cvs_plane* mag, const cvs_plane* phase)
{
    if (!h) return CVS_E_BADARG;
    int rc;
    if ((rc = check_plane(h, g, "g")) || (rc = check_plane(h, hq, "hq"))) return rc;
    if (!mag && !phase) return fail(h, CVS_E_BADARG, "no output requested");
    for (const cvs_plane* p : {hq, mag, phase}) {
        if (!p) continue;
        if ((rc = check_plane(h, p, "plane")) || (rc = check_same(h, p, g->rows, g->cols))) return rc;
    }
    if ((rc = check_point_overlaps(h, {g, hq}, {mag, phase}))) return rc;
    Call c;
    if ((rc = begin(h, c, {g, hq, mag, phase}))) return rc;
    PointArgs a{};
    a.rows = g->rows;
    a.cols = g->cols;
    a.atan_mode = h->atan_mode;
    if ((rc = in_ref(c, g, a.in[0])) || (rc = in_ref(c, hq, a.in[1]))) return rc;
    if ((rc = out_ref(c, mag, a.out[0])) || (rc = out_ref(c, phase, a.out[1]))) return rc;
    a.nt_stores = use_nt_stores(h, (size_t)a.rows * a.cols);
    HIP_TRY(h, launch_point(OP_MAG_PHASE, a, h->stream));
    return finish(c);
}

int cvs_phase_weights(cvs_handle h, const cvs_plane* phase, const cvs_plane* lambda, float phi, int signum, float k)
{
    (void)k;  // accepted and ignored, like the reference (G2.cpp:179-186)
    if (!h) return CVS_E_BADARG;
    int rc;
    if ((rc = check_plane(h, phase, "phase")) || (rc = check_plane(h, lambda, "lambda"))) return rc;
    if ((rc = check_same(h, lambda, phase->rows, phase->cols))) return rc;
    if ((rc = check_point_overlaps(h, {phase}, {lambda}))) return rc;
    Call c;
    if ((rc = begin(h, c, {phase, lambda}))) return rc;
    PointArgs a{};
    a.rows = phase->rows;
    a.cols = phase->cols;
    a.phi = phi;
    a.signum = signum ? 1 : 0;
    if ((rc = in_ref(c, phase, a.in[0])) || (rc = out_ref(c, lambda, a.out[0]))) return rc;
    a.nt_stores = use_nt_stores(h, (size_t)a.rows * a.cols);
    HIP_TRY(h, launch_point(OP_PHASE_WEIGHTS, a, h->stream));
    return finish(c);
}

int cvs_wrap(cvs_handle h, const cvs_plane* angle, const cvs_plane* out)
{
    if (!h) return CVS_E_BADARG;
    int rc;
    if ((rc = check_plane(h, angle, "angle")) || (rc = check_plane(h, out, "out"))) return rc;
    if ((rc = check_same(h, out, angle->rows, angle->cols))) return rc;
    if ((rc = check_point_overlaps(h, {angle}, {out}))) return rc;
    Call c;
    if ((rc = begin(h, c, {angle, out}))) return rc;
    PointArgs a{};
    a.rows = angle->rows;
    a.cols = angle->cols;
    if ((rc = in_ref(c, angle, a.in[0])) || (rc = out_ref(c, out, a.out[0]))) return rc;
    a.nt_stores = use_nt_stores(h, (size_t)a.rows * a.cols);
    HIP_TRY(h, launch_point(OP_WRAP, a, h->stream));
    return finish(c);
}

int cvs_find(cvs_handle h, const cvs_plane* e, const cvs_plane* phase, const cvs_plane* edges, const cvs_plane* dark,
             const cvs_plane* bright)
{
    if (!h) return CVS_E_BADARG;
    int rc;
    if ((rc = check_plane(h, e, "e")) || (rc = check_plane(h, phase, "phase"))) return rc;
    if (!edges && !dark && !bright) return fail(h, CVS_E_BADARG, "no output requested");
    for (const cvs_plane* p : {phase, edges, dark, bright}) {
        if (!p) continue;
        if ((rc = check_plane(h, p, "plane")) || (rc = check_same(h, p, e->rows, e->cols))) return rc;
    }
    if ((rc = check_point_overlaps(h, {e, phase}, {edges, dark, bright}))) return rc;
    Call c;
    if ((rc = begin(h, c, {e, phase, edges, dark, bright}))) return rc;
    PointArgs a{};
    a.rows = e->rows;
    a.cols = e->cols;
    if ((rc = in_ref(c, e, a.in[0])) || (rc = in_ref(c, phase, a.in[1]))) return rc;
    if ((rc = out_ref(c, edges, a.out[0])) || (rc = out_ref(c, dark, a.out[1])) || (rc = out_ref(c, bright, a.out[2]))) return rc;
    a.nt_stores = use_nt_stores(h, (size_t)a.rows * a.cols);
    HIP_TRY(h, launch_point(OP_FIND, a, h->stream));
    return finish(c);
}

int cvs_pipeline(cvs_handle h, const cvs_plane* image, const cvs_plane* const outs[8])
{
    if (!h || !outs) return CVS_E_BADARG;
    if (h->kind != CVS_KIND_G2) return fail(h, CVS_E_UNSUPPORTED, "the caller pipeline exists for G2 only");
    int rc = check_plane(h, image, "image", true);
    if (rc) return rc;
    for (int o = 0; o < 8; ++o) {
        if (!outs[o]) continue;
        if ((rc = check_plane(h, outs[o], "out")) || (rc = check_same(h, outs[o], image->rows, image->cols))) return rc;
    }
    // one launch: filter bank, orientation and the whole caller sequence in the kernel's epilogue
    return do_setup(h, image, CVS_SETUP_FULL, false, 0.f, nullptr, nullptr, outs);
}

int cvs_pipeline_batch(cvs_handle h, const cvs_plane* images, int n, const cvs_plane* outs)
{
    if (!h || !images || n < 1) return CVS_E_BADARG;
    if (h->kind != CVS_KIND_G2) return fail(h, CVS_E_UNSUPPORTED, "the caller pipeline exists for G2 only");
    int rc;
    const int rows = images[0].rows, cols = images[0].cols;
    bool all_dev = true;
    size_t max_bytes = 0;
    for (int i = 0; i < n; ++i) {
        if ((rc = check_plane(h, &images[i], "image", true)) || (rc = check_same(h, &images[i], rows, cols))) return rc;
        all_dev = all_dev && images[i].mem == CVS_MEM_DEVICE;  // f32 on the device; 8-bit / host frames go frame by frame
        if (!is_u8(&images[i])) max_bytes = std::max(max_bytes, (size_t)rows * images[i].step);
        for (int k = 0; outs && k < 8; ++k) {
            const cvs_plane* o = &outs[(size_t)i * 8 + k];
            if (!o->data) continue;
            if ((rc = check_plane(h, o, "out")) || (rc = check_same(h, o, rows, cols))) return rc;
            all_dev = all_dev && o->mem == CVS_MEM_DEVICE;
            max_bytes = std::max(max_bytes, (size_t)rows * o->step);
        }
        if (outs) {
            const cvs_plane* po[8];
            for (int k = 0; k < 8; ++k) po[k] = outs[(size_t)i * 8 + k].data ? &outs[(size_t)i * 8 + k] : nullptr;
            if ((rc = check_no_overlap(h, &images[i], po, 8))) return rc;
        }
    }
    // 8-bit frames that lie back to back on the device (a driver's upload of a block of byte images): the one-launch path
    // below reads the bytes itself (BasisArgs::in_u8), like any regular f32 batch -- no widened copy
    bool u8_batch = n >= 1 && images[0].mem == (CVS_MEM_DEVICE | CVS_DEPTH_U8);
    {
        const uint8_t* b0 = reinterpret_cast<const uint8_t*>(images[0].data);
        for (int i = 0; i < n && u8_batch; ++i)
            u8_batch = images[i].mem == images[0].mem && images[i].step == images[0].step &&
                       reinterpret_cast<const uint8_t*>(images[i].data) == b0 + (size_t)i * rows * images[0].step;
        u8_batch = u8_batch && (size_t)rows * images[0].step <= (size_t)0x7ffffff0;
        if (u8_batch) {
            all_dev = true;
            for (int i = 0; i < n && all_dev; ++i)
                for (int k = 0; outs && k < 8 && all_dev; ++k)
                    if (outs[(size_t)i * 8 + k].data) all_dev = outs[(size_t)i * 8 + k].mem == CVS_MEM_DEVICE;
        }
    }
    const size_t pitch = round_up((size_t)cols, 64);
    // one launch over grid.z needs every plane below 2 GiB (huge frames are filtered in row bands, frame by frame)
    const bool small_planes = std::max(max_bytes, (size_t)rows * pitch * sizeof(float)) <= (size_t)0x7ffffff0;
    const bool fast = all_dev && small_planes &&
                      !basis_may_need_scratch(h->kind, h->width, h->taps, rows, cols, std::max(pitch, max_bytes / sizeof(float) / rows));
    if (!fast) {
        // host planes, tiny or huge images, non-default taps: frame by frame through the single-image path
        for (int i = 0; i < n; ++i) {
            const cvs_plane* po[8];
            for (int k = 0; k < 8; ++k) po[k] = (outs && outs[(size_t)i * 8 + k].data) ? &outs[(size_t)i * 8 + k] : nullptr;
            if ((rc = do_setup(h, &images[i], CVS_SETUP_FULL, false, 0.f, nullptr, nullptr, po, n, i))) return rc;
        }
        h->cur_frame = 0;
        return CVS_OK;
    }
    HIP_TRY(h, hipSetDevice(h->device));
    h->used = true;
    h->have_basis = h->have_orient = false;
    if ((rc = ensure_state(h, rows, cols, n))) return rc;
    h->cur_frame = 0;
    std::vector<BatchFrame> tab(n);
    for (int i = 0; i < n; ++i) {
        tab[i].in = images[i].data;
        tab[i].in_pitch = u8_batch ? images[i].step : images[i].step / sizeof(float);   // elements of the image's own type
        for (int k = 0; k < 8; ++k) {
            const cvs_plane* o = outs ? &outs[(size_t)i * 8 + k] : nullptr;
            tab[i].out[k] = (o && o->data) ? PlaneRef{o->data, o->step / sizeof(float)} : PlaneRef{nullptr, 0};
        }
    }
    // Regularly strided frames -- one [n, H, W] block in, one [n, K, H, W] block out, the usual case -- need no
    // table: frame z is frame 0 plus z strides, computed in the kernel from its arguments.  Anything else (a list
    // of unrelated planes) goes through a device table, uploaded on the handle's stream.
    bool regular = true;
    ptrdiff_t d_in = 0, d_out = 0;
    bool have_out_stride = false;
    for (int i = 1; i < n && regular; ++i) {
        // (8-bit frames: byte addresses, and u8_batch has already established that they lie back to back)
        const ptrdiff_t di = u8_batch ? (ptrdiff_t)((size_t)i * rows * images[0].step) : tab[i].in - tab[0].in;
        if (i == 1) d_in = di;
        regular = di == d_in * i && d_in >= 0 && tab[i].in_pitch == tab[0].in_pitch;
        for (int k = 0; k < 8 && regular; ++k) {
            if ((tab[i].out[k].p == nullptr) != (tab[0].out[k].p == nullptr)) regular = false;
            else if (tab[i].out[k].p) {
                const ptrdiff_t dk = tab[i].out[k].p - tab[0].out[k].p;
                if (!have_out_stride) { d_out = dk / i; have_out_stride = true; }
                regular = dk == d_out * i && d_out >= 0 && tab[i].out[k].pitch == tab[0].out[k].pitch;
            }
        }
    }
    if (!regular) {
        if (n > h->frame_tab_cap) {
            if (h->frame_tab) {
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                HIP_TRY(h, hipFree(h->frame_tab));
                h->frame_tab = nullptr;
                h->frame_tab_cap = 0;
            }
            HIP_TRY(h, hipMalloc(&h->frame_tab, (size_t)n * sizeof(BatchFrame)));
            h->frame_tab_cap = n;
        }
        // pageable source: the runtime stages it before returning, so `tab` may go out of scope
        HIP_TRY(h, hipMemcpyAsync(h->frame_tab, tab.data(), (size_t)n * sizeof(BatchFrame), hipMemcpyHostToDevice, h->stream));
    }
    BasisArgs a{};
    a.rows = rows;
    a.cols = cols;
    a.in_pitch = pitch;
    a.in_u8 = u8_batch ? 1 : 0;
    if (regular) {
        a.batch_regular = 1;
        a.in = tab[0].in;
        a.in_pitch = tab[0].in_pitch;
        a.in_frame_stride = (size_t)d_in;
        a.out_frame_stride = (size_t)d_out;
        for (int k = 0; k < 8; ++k) a.pipe_out[k] = tab[0].out[k];
        // one buffer resource per frame for all outputs, if frame 0's outputs share a pitch and lie within 2 GiB
        float* lo = nullptr;
        size_t opitch = 0;
        bool one = true;
        for (int k = 0; k < 8; ++k) {
            if (!tab[0].out[k].p) continue;
            if (!lo || tab[0].out[k].p < lo) lo = tab[0].out[k].p;
            if (!opitch) opitch = tab[0].out[k].pitch;
            one = one && tab[0].out[k].pitch == opitch;
        }
        size_t span = 0;
        for (int k = 0; k < 8 && one; ++k) {
            if (!tab[0].out[k].p) continue;
            const size_t off = (size_t)(tab[0].out[k].p - lo) * sizeof(float);
            span = std::max(span, off + (size_t)rows * opitch * sizeof(float));
            one = span <= (size_t)0x7ffffff0;
            a.out_off[k] = (unsigned)off;
            a.out_mask |= 1u << k;
        }
        if (one) {
            a.out_one = 1;
            a.out_base = lo;
            a.out_pitch = opitch;
            a.out_bytes = span;
        } else {
            a.out_mask = 0;
        }
    }
    fill_state_args(h, a, true);   // frame 0 (cur_frame was reset above); frame z adds z * frame_stride in the kernel
    a.atan_mode = h->atan_mode;
    a.strip_rows = default_strip_rows(h, rows, cols);
    a.nt_stores = use_nt_stores(h, (size_t)rows * cols * n);
    a.pipe = 1;
    a.no_state = h->persist ? 0 : 1;
    a.find_on_e = h->find_on;
    a.frames = regular ? nullptr : h->frame_tab;
    a.batch = n;
    // state kept: frames from the two halves of the batch in flight together (see k_basis); the stateless launch is bound by
    // the SIMDs and does not care.  CVS_OPTS batch_ways=<n> is a tuning aid (1 = frames in order).
    a.z_ways = (!a.no_state && n >= 4) ? 2 : 1;
    // ... and on 10-row strips: round 3 sweep (profiles/r03_c4_strip_probe.txt), 32 x 1080p, five state blocks of the allocation lottery, one handle
    // each: against 19 rows in the plain order 0.634 / 0.70 / 0.70 / 0.796 / 0.795 for 0.644 / 0.70 / 0.70 / 0.762 / 0.764 --
    // level on the slow and middle blocks, +4.5 % on the fast ones
    if (!a.no_state && n >= 4 && h->strip_rows <= 0) a.strip_rows = 2 * (2 * h->width + 1) - 2 * h->width;
    if (const int ways = env_opts().batch_ways; ways > 0) a.z_ways = std::max(1, std::min(n, ways));
    a.frame_stride = h->frame_stride;
    // state kept: every frame is a new image -- the waves of a frame's first row bands also request the rest of the FRAME (two bands each:
    // BasisArgs::warm_k, per frame).  32 x 1080p, same handle, alternating, sustained: +1.2 ... +2.3 % in 7 of 7 processes on three boxes
    // (four bands +1.3 %, eight +0.3 %; profiles/r06_c4_warm.txt).  Not for the outputs-only batches (-1 %: they are bound by the SIMDs).
    if (!a.no_state && regular && (size_t)rows * cols >= ((size_t)1 << 20)) a.warm_k = env_opts().warm >= 0 ? env_opts().warm : 2;
    TuneToken tok;
    if ((rc = tune_begin(h, a, 16 | 1 | 4 | (a.no_state ? 8 : 0), false, tok))) return rc;
    note_launch(h, a);
    const hipError_t le = launch_basis(h->kind, h->width, h->taps, a, nullptr, h->stream);
    tune_end(h, tok);
    HIP_TRY(h, le);
    h->have_basis = h->have_orient = h->persist != 0;
    return CVS_OK;
}

#ifdef CVS_DIAG_STAMPS
// diagnostic builds only: per-wave time stamps of the next basis launches go to `buf` (device memory,
// 4 x 8 bytes per wave); not declared in the public header
int cvs_diag_set_buffer(cvs_handle h, void* buf)
{
    if (!h) return CVS_E_BADARG;
    h->diag = static_cast<unsigned long long*>(buf);
    return CVS_OK;
}
#endif

int cvs_select_frame(cvs_handle h, int frame)
{
    if (!h) return CVS_E_BADARG;
    if (frame < 0 || frame >= h->num_frames) return fail(h, CVS_E_BADARG, "frame index");
    h->cur_frame = frame;
    return CVS_OK;
}

int cvs_num_frames(cvs_handle h, int* n)
{
    if (!h || !n) return CVS_E_BADARG;
    *n = h->have_basis ? h->num_frames : 0;
    return CVS_OK;
}

int cvs_pyr_down(cvs_handle h, const cvs_plane* src, const cvs_plane* dst)
{
    if (!h) return CVS_E_BADARG;
    int rc;
    if ((rc = check_plane(h, src, "src")) || (rc = check_plane(h, dst, "dst"))) return rc;
    if ((rc = check_same(h, dst, (src->rows + 1) / 2, (src->cols + 1) / 2))) return rc;
    if (planes_overlap(src, dst)) return fail(h, CVS_E_BADARG, "the level overlaps the image it is made from");
    Call c;
    if ((rc = begin(h, c, {src, dst}))) return rc;
    PlaneRef in, out;
    if ((rc = in_ref(c, src, in)) || (rc = out_ref(c, dst, out))) return rc;
    hipError_t pe = hipSuccess;
    if (h->pyr_strip && launch_pyr_strip(in.p, in.pitch, src->rows, src->cols, out.p, out.pitch, h->stream, &pe)) HIP_TRY(h, pe);
    else HIP_TRY(h, launch_pyr_down(in.p, in.pitch, src->rows, src->cols, out.p, out.pitch, h->stream));
    return finish(c);
}

static int to_u8(cvs_handle h, const cvs_plane* src, uint8_t* dst, size_t dst_step, int dst_mem, bool minmax, float alpha, float beta)
{
    if (!h || !dst) return CVS_E_BADARG;
    int rc = check_plane(h, src, "src");
    if (rc) return rc;
    if (dst_step < (size_t)src->cols) return fail(h, CVS_E_SIZE, "dst_step");
    if (dst_mem != CVS_MEM_HOST && dst_mem != CVS_MEM_DEVICE) return fail(h, CVS_E_BADARG, "dst_mem");
    const size_t dpitch = round_up((size_t)src->cols, 256);
    const size_t u8_elems = dst_mem == CVS_MEM_HOST ? round_up(dpitch * src->rows / 4 + 64, 64) : 0;
    Call c;
    if ((rc = begin(h, c, {src}, u8_elems + (minmax ? 64 : 0)))) return rc;
    PlaneRef in;
    if ((rc = in_ref(c, src, in))) return rc;
    float* mm = minmax ? arena_take(h, 64) : nullptr;   // min / max scratch from the arena (no allocation of its own, cf. to_u8_batch)
    uint8_t* d = dst;
    size_t dstep = dst_step;
    if (dst_mem == CVS_MEM_HOST) {
        d = reinterpret_cast<uint8_t*>(arena_take(h, u8_elems));
        dstep = dpitch;
    }
    if (minmax) {
        HIP_TRY(h, launch_minmax(in.p, in.pitch, src->rows, src->cols, mm, h->stream));
        HIP_TRY(h, launch_quantize_u8(in.p, in.pitch, src->rows, src->cols, mm, d, dstep, h->stream));
    } else {
        HIP_TRY(h, launch_convert_u8(in.p, in.pitch, src->rows, src->cols, alpha, beta, d, dstep, h->stream));
    }
    if (dst_mem == CVS_MEM_DEVICE) return finish(c);
    HIP_TRY(h, copy_rows(dst, dst_step, d, dstep, (size_t)src->cols, src->rows, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CVS_OK;
}

// n planes at once: one min/max launch, one quantise launch, the copies to the host queued behind them and ONE
// synchronisation -- what a driver wants that turns a rank's whole block of feature maps into files (per plane, the
// launch + copy + sync of the single-plane call costs more than the work).  Planes that are not equally sized device
// planes at a constant stride go one by one.
static int to_u8_batch(cvs_handle h, const cvs_plane* src, int n, uint8_t* const* dst, size_t dst_step, int dst_mem, bool minmax, float alpha, float beta)
{
    if (!h || !src || !dst || n < 1) return CVS_E_BADARG;
    if (dst_mem != CVS_MEM_HOST && dst_mem != CVS_MEM_DEVICE) return fail(h, CVS_E_BADARG, "dst_mem");
    int rc;
    bool regular = true;
    const ptrdiff_t stride = n > 1 ? src[1].data - src[0].data : 0;
    for (int i = 0; i < n; ++i) {
        if ((rc = check_plane(h, &src[i], "src"))) return rc;
        if (!dst[i]) return fail(h, CVS_E_BADARG, "dst");
        regular = regular && src[i].mem == CVS_MEM_DEVICE && src[i].rows == src[0].rows && src[i].cols == src[0].cols && src[i].step == src[0].step &&
                  src[i].data - src[0].data == stride * i;
    }
    if (dst_step < (size_t)src[0].cols) return fail(h, CVS_E_SIZE, "dst_step");
    regular = regular && stride >= 0 && (size_t)src[0].rows * src[0].step <= (size_t)0x7ffffff0;
    if (!regular) {
        for (int i = 0; i < n; ++i)
            if ((rc = to_u8(h, &src[i], dst[i], dst_step, dst_mem, minmax, alpha, beta))) return rc;
        return CVS_OK;
    }
    const int rows = src[0].rows, cols = src[0].cols;
    HIP_TRY(h, hipSetDevice(h->device));
    h->used = true;
    // scratch: 2n floats of min / max, and (host destinations) n staged byte planes
    // host destinations that lie back to back ([n][rows][dst_step], the usual block) are staged in exactly that layout
    // and come down as ONE linear copy (a pitched 2-D copy of the same bytes runs at a third of the link rate)
    bool packed = dst_mem == CVS_MEM_HOST && dst_step == (size_t)cols;  // padded rows keep their padding: copied row by row
    for (int i = 1; i < n && packed; ++i) packed = dst[i] == dst[0] + (size_t)i * rows * dst_step;
    const size_t dpitch = packed ? dst_step : round_up((size_t)cols, 256), plane_b = dpitch * rows;
    const size_t mm_elems = round_up((size_t)2 * n, 64);
    const size_t stage_elems = dst_mem == CVS_MEM_HOST ? round_up(plane_b * n / 4 + 64, 64) : 0;
    if ((rc = arena_reserve(h, mm_elems + stage_elems))) return rc;
    h->arena_used = 0;
    float* mm = arena_take(h, mm_elems);
    if (dst_mem == CVS_MEM_HOST) {
        uint8_t* stage = reinterpret_cast<uint8_t*>(arena_take(h, stage_elems));
        HIP_TRY(h, launch_to_u8_n(src[0].data, (size_t)stride, src[0].step / sizeof(float), rows, cols, n, minmax, mm, alpha, beta, stage, plane_b, dpitch, h->stream));
        if (packed) {
            HIP_TRY(h, hipMemcpyAsync(dst[0], stage, plane_b * n, hipMemcpyDeviceToHost, h->stream));
        } else {
            for (int i = 0; i < n; ++i)
                HIP_TRY(h, copy_rows(dst[i], dst_step, stage + (size_t)i * plane_b, dpitch, (size_t)cols, rows, hipMemcpyDeviceToHost, h->stream));
        }
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return CVS_OK;
    }
    // device destinations: regular too?  then straight into them, else plane by plane
    const ptrdiff_t dstride = n > 1 ? dst[1] - dst[0] : 0;
    bool dreg = dstride >= 0;
    for (int i = 0; i < n && dreg; ++i) dreg = dst[i] - dst[0] == dstride * i;
    if (dreg) {
        HIP_TRY(h, launch_to_u8_n(src[0].data, (size_t)stride, src[0].step / sizeof(float), rows, cols, n, minmax, mm, alpha, beta, dst[0], (size_t)dstride, dst_step, h->stream));
        return CVS_OK;
    }
    for (int i = 0; i < n; ++i)
        if ((rc = to_u8(h, &src[i], dst[i], dst_step, dst_mem, minmax, alpha, beta))) return rc;
    return CVS_OK;
}

int cvs_normalize_u8_batch(cvs_handle h, const cvs_plane* src, int n, uint8_t* const* dst, size_t dst_step, int dst_mem)
{
    return to_u8_batch(h, src, n, dst, dst_step, dst_mem, true, 0.f, 0.f);
}

int cvs_convert_u8_batch(cvs_handle h, const cvs_plane* src, int n, float alpha, float beta, uint8_t* const* dst, size_t dst_step, int dst_mem)
{
    return to_u8_batch(h, src, n, dst, dst_step, dst_mem, false, alpha, beta);
}

int cvs_normalize_u8(cvs_handle h, const cvs_plane* src, uint8_t* dst, size_t dst_step, int dst_mem)
{
    return to_u8(h, src, dst, dst_step, dst_mem, true, 0.f, 0.f);
}

int cvs_convert_u8(cvs_handle h, const cvs_plane* src, float alpha, float beta, uint8_t* dst, size_t dst_step, int dst_mem)
{
    return to_u8(h, src, dst, dst_step, dst_mem, false, alpha, beta);
}

}  // extern "C"
