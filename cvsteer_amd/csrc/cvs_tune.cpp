// cvs_tune.cpp -- how a basis launch is configured: strip height, store policy, launch order.
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

int default_strip_rows(cvs_handle h, int rows, int cols, bool fresh_input)
{
    if (h->strip_rows > 0) return h->strip_rows;
    // Heights of k * (2W+1) - 2W rows waste no iteration of the unrolled row loop.  Small images: enough strips for about
    // 2048 waves.  Large ones:
    //   G2  10 rows (k = 2): ~14k waves per 4096^2 launch keep every CU's store queues busy, the extra halo rows are cache
    //       hits, and on a stream of new images vertically adjacent strips must run close in time for their shared rows to
    //       hit (8 rotating images, round 2: 10-row strips 66 %, 19-row strips 57 %).  19 rows (k = 3) remain the default on a
    //       placement window (planar planes) and for states the Infinity Cache holds; the tuner compares both heights.
    //   G4  40 rows (k = 4) for the half banks: the pair launch is close to SIMD-bound, and 27-row strips filter 44 % more
    //       rows than they write against 30 % (-3..-6 %, profiles/r04_order_probe.txt).
    const int nt = 2 * h->width + 1, halo = 2 * h->width;
    const long strips_x = (cols + 63) / 64;
    const double ideal = (double)rows * (double)strips_x / 2048.0;
    long k = std::lround((ideal + halo) / nt);
    const bool plain_block = !h->sb.vmm && h->sb.base != nullptr && h->num_frames == 1 &&
                             (size_t)rows * cols * sizeof(float) * (size_t)(h->nb + 5) >= ((size_t)256 << 20);   // states the Infinity Cache cannot hold
    const long kmax = h->kind == CVS_KIND_G4 ? 4 : (fresh_input || plain_block || (size_t)rows * cols >= ((size_t)32 << 20)) ? 2 : 3;
    if (k < 2) k = 2;
    if (k > kmax) k = kmax;
    return (int)(k * nt - halo);
}

// Streaming (nontemporal) stores: "1 plane in, 7 out" (tools/membench.hip, profiles/r01_membench.txt)
// reaches ~5.9 TB/s with nt stores vs ~4.0 TB/s with plain stores once the planes no longer fit
// the 256 MiB Infinity Cache.  Small frames whose whole state stays cache-resident keep plain
// stores so the next per-pixel kernel finds them on die.
int use_nt_stores(cvs_handle h, size_t npix)
{
    if (h->store_policy == 1) return 0;
    if (h->store_policy == 2) return 1;
    const size_t state_bytes = npix * sizeof(float) * (size_t)(h->nb + 5);
    return state_bytes > (size_t)96 << 20;
}

// ---------------------------------------------------------------------------------------------------------------------
// Launch configuration: defaults, and the ONLINE comparison of a few alternatives on the caller's own launches.
//
// What a basis launch leaves open is the order in which its tiles are dealt to the chip (block_order / XCD shares), the strip
// height and, for G4, the bank layout.  Which combination is fastest depends on the box and -- more -- on the PROCESS, i.e. on
// where the allocator put the planes (round 4, six processes on one box, profiles/r04_order_probe.txt: the XCD-weighted order
// +5 % for the 12- and 20-plane launches in two of them, -4..-7 % in the other four; the XCD-column order best for the basis
// pass everywhere and worst for the pipeline in some), so a short list is compared where the code runs.  Rounds 2-3 did that with a burst of
// ~250 extra launches on the second call of a shape -- 25-30 ms during which the caller's stream stalled and the caller's
// output planes were rewritten over and over.  Since round 4 NOTHING extra is launched: while a shape is being tuned, each of
// the caller's own calls runs one candidate, bracketed by a pair of events on the caller's stream; candidates take turns in
// blocks of kBlock consecutive calls (the first call of a block is not counted: a configuration's first launch after a
// change runs slower than the ones that follow it), kRounds times, and the default once more at the end; when every sample has been read back -- at some later
// call, never by waiting -- the candidate with the lowest MEDIAN is kept, the default unless a challenger beats it by 3 %.  Until then and for shapes
// seen once, the default runs.  Process-wide (the reference's callers build one object per image), keyed by device, kind,
// kernel variant, shape, batch size, layout and what the caller pinned.  cvs_launch_info.tuning_launches stays 0.
// ---------------------------------------------------------------------------------------------------------------------
struct Cand {
    int order, xw, strip, split;
    int ahead = 0;   // launches on new images: a pure-read pass over the image first (cvs_kernels_point.hip k_read_ahead)
    int merge = 0;   // G2 launches that write orientation planes: one 12-plane group instead of basis | orientation (cvs_handle.cpp layout_state)
    int cap = 0;     // workgroups per CU (0 = no cap).  NOT offered by build_candidates: as a second stage of this comparison (the winner
                     // against itself with one workgroup per CU less) it was picked where sustained launches then ran 2 % slower and
                     // skipped where they would have run 3 % faster -- candidates that take turns every few launches share one
                     // power / clock state, and what the cap changes is exactly that (profiles/r04_occupancy_probe.txt)
    bool operator==(const Cand& o) const { return order == o.order && xw == o.xw && strip == o.strip && split == o.split && merge == o.merge && ahead == o.ahead && cap == o.cap; }
};

struct TuneEntry {
    std::vector<Cand> cand;        // cand[0] = the default
    std::vector<float> best;       // MEDIAN of the counted launches of each candidate, ms (filled in when the entry is decided)
    std::vector<std::vector<float>> samples;
    std::vector<int> nsamp;
    int cur = 0, in_block = 0, round = 0;
    int pending = 0;               // samples recorded but not read back yet
    int chosen = -1;               // index into cand once decided
    bool done_issuing = false;
};

struct Sample {
    TuneEntry* entry;
    int cand;
    hipEvent_t e0, e1;
    int device;
};

typedef std::tuple<int, int, int, int, int, int, int> TuneKey;
static std::mutex g_tune_mutex;
static std::map<TuneKey, TuneEntry> g_tune;          // node-based: TuneEntry* stays valid
static std::vector<Sample> g_samples;                // in flight
static std::map<int, std::vector<hipEvent_t>> g_free_events;   // per device: timing events are recycled, never destroyed while the process lives

constexpr int kBlock = 3, kRounds = 2;

static void apply(BasisArgs& a, const Cand& c, bool xw_pinned)
{
    a.block_order = c.order;
    if (!xw_pinned) {
        a.xcd_even = c.xw / 100;
        a.xcd_odd = c.xw % 100;
    }
    a.strip_rows = c.strip;
    a.g4_split = c.split;
    a.merge_orient = c.merge;
    a.read_ahead = c.ahead;
    if (c.cap) a.wg_per_cu = c.cap;
}

// read back every sample whose launch has finished (never waits); decide entries that are complete.  g_tune_mutex held.
static void harvest()
{
    if (g_samples.empty()) return;
    RelaxedCapture relaxed;   // another thread's stream may be under (global-mode) capture: these queries concern none of its events
    for (size_t i = 0; i < g_samples.size();) {
        Sample& sm = g_samples[i];
        const hipError_t q = hipEventQuery(sm.e1);
        if (q == hipErrorNotReady) {
            ++i;
            continue;
        }
        float ms = 0.f;
        TuneEntry& e = *sm.entry;
        if (q == hipSuccess && hipEventElapsedTime(&ms, sm.e0, sm.e1) == hipSuccess && ms > 0.f) {
            e.samples[sm.cand].push_back(ms);
            ++e.nsamp[sm.cand];
        }
        (void)hipGetLastError();
        --e.pending;
        g_free_events[sm.device].push_back(sm.e0);
        g_free_events[sm.device].push_back(sm.e1);
        g_samples[i] = g_samples.back();
        g_samples.pop_back();
        if (e.done_issuing && e.pending == 0 && e.chosen < 0) {
            // The median of a candidate's samples, not the fastest one: a configuration whose launches vary more would win on
            // its luckiest sample (the three-workgroup cap once did, 2.4 % ahead on its best launch and 3 % behind sustained)
            for (size_t c = 0; c < e.cand.size(); ++c) {
                std::vector<float>& v = e.samples[c];
                if (v.empty()) continue;
                std::sort(v.begin(), v.end());
                e.best[c] = v.size() % 2 ? v[v.size() / 2] : 0.5f * (v[v.size() / 2 - 1] + v[v.size() / 2]);
            }
            int best = 0;
            for (int c = 1; c < (int)e.cand.size(); ++c)
                // a challenger must win by 3 % -- the read-ahead pass by 1.5 %: it is either a clear loss (the image was in the cache
                // already: -5..-12 %) or worth 4-6 % sustained, of which this comparison sees about half
                if (e.nsamp[c] > 0 && e.best[c] < e.best[best] * (best == 0 ? (e.cand[c].ahead ? 0.985f : 0.97f) : 1.0f)) best = c;
            if (e.nsamp[0] == 0) best = 0;
            e.chosen = best;
            if (std::getenv("CVS_TUNE_VERBOSE")) {
                std::fprintf(stderr, "[cvsteer] tuned on the caller's launches:");
                for (size_t c = 0; c < e.cand.size(); ++c)
                    std::fprintf(stderr, " (order %d, xcd %d, strip %d, split %d, merged %d, read-ahead %d) %.4f ms x%d", e.cand[c].order, e.cand[c].xw, e.cand[c].strip, e.cand[c].split, e.cand[c].merge, e.cand[c].ahead,
                                 e.nsamp[c] ? e.best[c] : 0.f, e.nsamp[c]);
                std::fprintf(stderr, " -> candidate %d\n", best);
            }
        }
    }
}

static hipEvent_t take_event(int device)   // the caller has made `device` current
{
    std::vector<hipEvent_t>& pool = g_free_events[device];
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return e;
}

// the engine's default configuration for this launch; true = the launch is of a kind whose alternatives are worth comparing
static bool default_config(cvs_handle h, BasisArgs& a, int variant, bool fresh_input)
{
    const int xw_pinned = h->xcd_weights;
    a.xcd_even = xw_pinned ? xw_pinned / 100 : 5;
    a.xcd_odd = xw_pinned ? xw_pinned % 100 : 4;
    a.g4_split = h->g4_split >= 0 ? h->g4_split : 2;
    a.wg_per_cu = h->wg_per_cu;
    // CVS_OPT_STATE_LAYOUT = 2 pins the grouping: launches that write orientation planes use ONE group of twelve planes
    a.merge_orient = (h->layout == 2 && h->kind == CVS_KIND_G2 && a.orient && !a.no_state && a.batch == 0) ? 1 : 0;
    const bool fast = basis_fast_path(h->kind, h->width, h->taps);
    const bool big = (size_t)a.rows * a.cols >= ((size_t)1 << 20);
    // the plain row-major order: with the row-interleaved state it is within a few per cent of the best order on every box
    // and every variant measured, resident image or fresh (profiles/r04_layout_probe.txt); what beats it is box-dependent
    a.block_order = h->block_order >= 0 ? h->block_order : 0;
    // Placement windows (the opt-in search; planar planes), resident image: what round 3's sweeps found best there --
    // 10-row strips, every XCD on its own range of column blocks (fused steer: odd XCDs leave a thirteenth to their even
    // neighbours), the pipeline in the plain order (profiles/r03_launch_config_sweeps.txt)
    if (h->sb.vmm && h->kind == CVS_KIND_G2 && fast && big && !fresh_input && a.batch == 0 && !a.no_state && h->block_order < 0 &&
        h->strip_rows <= 0 && (size_t)a.rows * a.cols < ((size_t)32 << 20)) {
        const int grid_x = ((a.cols + 63) / 64 + 3) / 4;
        a.strip_rows = 2 * (2 * h->width + 1) - 2 * h->width;
        if (!(variant & 4) && grid_x % 8 == 0) {
            a.block_order = kOrderXcdColumns;
            if (!xw_pinned && (variant & 2)) { a.xcd_even = 7; a.xcd_odd = 6; }
        }
    }
    if (a.block_order == kOrderXcdColumns && !xw_pinned && !(a.xcd_even == 7 && a.xcd_odd == 6)) a.xcd_even = a.xcd_odd = 1;
    return fast && big;
}

static void build_candidates(cvs_handle h, const BasisArgs& a, bool fresh_input, TuneEntry& e)
{
    const int xw_pinned = h->xcd_weights;
    const bool free_order = h->block_order < 0;
    const bool free_strip = h->strip_rows <= 0 && (a.batch == 0 || !a.no_state);
    const bool free_split = h->kind == CVS_KIND_G4 && h->g4_split < 0;
    const int xw0 = a.xcd_even * 100 + a.xcd_odd;
    Cand def{a.block_order, xw0, a.strip_rows, a.g4_split};
    def.merge = a.merge_orient;
    e.cand.assign(1, def);
    auto add = [&](Cand c) {
        const bool deals = c.order == 1 || c.order == kOrderXcdColumns;   // orders in which the even : odd shares matter
        if (c.order == kOrderDynamic && !a.tile_ctr) return;
        if (c.order != def.order && !free_order) return;
        if (c.strip != def.strip && !free_strip) return;
        if (c.split != def.split && !free_split) return;
        if (deals && xw_pinned) c.xw = xw0;
        if (!deals) c.xw = xw0;
        c.merge = def.merge;
        const int grid_x = ((a.cols + 63) / 64 + 3) / 4;
        if (c.order == kOrderXcdColumns && (grid_x % 8 != 0 || a.batch != 0)) return;
        for (const Cand& k : e.cand)
            if (k == c) return;
        if (e.cand.size() < 7) e.cand.push_back(c);
    };
    const int nt = 2 * h->width + 1, halo = 2 * h->width;
    const int sr_short = 2 * nt - halo, sr_tall = 3 * nt - halo;
    if (h->kind == CVS_KIND_G2) {
        if (fresh_input) {
            // a stream of new images: short strips are a must (the halo rows of vertically adjacent strips only hit in cache when
            // those strips run close in time), the weighted order loses 4-6 points, and the plain order, the XCD-column order and
            // the dynamic tail are within 1 % of each other in sustained runs (tools/ab_same.py AB_ROT=1 AB_STEPS=300) -- offered,
            // the latter two only displaced the read-ahead pass below, the one candidate that is worth 3-5 % here, in two runs of six.
            // Large images: the default against itself with the read-ahead pass, nothing else.
            if ((size_t)a.rows * a.cols < ((size_t)8 << 20)) {
                add({kOrderDynamic, xw0, sr_short, def.split});
                add({kOrderXcdColumns, 101, sr_short, def.split});
                add({0, xw0, sr_short, def.split});
            }
        } else {
            // Resident image.  What has beaten the default (plain order, 10-row strips) by more than 2 % in SUSTAINED side-by-side
            // runs on one handle (tools/ab_same.py AB_STEPS=300, tools/tuner_value_probe.py; profiles/r04_order_probe_sustained.txt):
            // for launches that also write the orientation planes (12 / 20 planes) the weighted order (+4-6 %), the dynamic tail
            // (+3 %) and the merged grouping (below) on handles whose plane groups lie badly; for the basis pass and the fused steer
            // NOTHING (all orders within 1 %, taller strips behind) -- and there the comparison itself did harm: candidates that
            // take turns every few launches share one power / clock state, and in one run in seven a configuration that is 4 %
            // slower sustained (19-row strips, the XCD-column order) was kept.  So large single-group launches are not tuned.
            const bool multi = a.orient != nullptr || a.pipe;
            const bool large = (size_t)a.rows * a.cols >= ((size_t)8 << 20);
            if (multi || !large) {
                add({kOrderDynamic, xw0, sr_short, def.split});     // the tail handed out from per-XCD queues: an XCD that is ahead helps the others
                add({1, 504, sr_short, def.split});                 // more tiles for the faster XCDs: wins where the XCDs differ
                add({0, xw0, sr_short, def.split});
            }
            if (!large) {   // smaller images (the default there is the 19-row strip): both heights, both leading orders
                add({kOrderXcdColumns, 101, sr_short, def.split});
                add({1, 403, sr_tall, def.split});
                add({0, xw0, sr_tall, def.split});
            }
        }
    } else {
        // G4: the dynamic tail (+6 % on one box, level on the others).  Not offered any more: the weighted order (3-4 % behind in every
        // sustained run, yet kept once by the comparison), the single 11-plane kernel (15 % behind everywhere), 53-row strips (+-1.5 %)
        // -- profiles/r04_g4_strips_probe.txt, r04_g4_bank_layouts.txt; CVS_OPT_G4_SPLIT / CVS_OPT_STRIP_ROWS still pin them
        add({kOrderDynamic, xw0, def.strip, def.split});
    }
    // Launches on NEW images (a handle's first call, or another image than last time) of a size the Infinity Cache holds beside
    // the launch's own traffic: the two leading configurations again with a pure-read pass over the image in front.  Where the
    // image really comes from HBM that wins 3-6 %; where it is in the cache already (the previous kernel made it) it loses
    // its 5-12 us and is dropped.
    const size_t in_bytes = (size_t)a.rows * a.cols * (a.in_u8 ? 1 : sizeof(float));
    static const bool no_read_ahead = std::getenv("CVS_NO_READ_AHEAD") != nullptr;   // A/B aid (tools/fresh_probe.py)
    if (fresh_input && a.batch == 0 && in_bytes >= ((size_t)4 << 20) && in_bytes <= ((size_t)128 << 20) && !no_read_ahead) {
        const size_t n0 = std::min<size_t>(e.cand.size(), 2);
        for (size_t i = 0; i < n0; ++i) {
            Cand m = e.cand[i];
            m.ahead = 1;
            e.cand.push_back(m);
        }
    }
    // G2 launches that write the orientation planes too (full setup, pipeline), row-interleaved state, single image: the same
    // three leading configurations with ALL twelve planes in one group -- steadier (0.81-0.82 for the full setup in every
    // process) where two groups are either faster (0.85) or slower (0.755) depending on where the block lies
    if (h->kind == CVS_KIND_G2 && a.orient && !a.no_state && a.batch == 0 && h->last.state_layout != 0 && h->layout == 1 &&
        state_merge_fits(h, a.rows, h->dense_pitch)) {
        const size_t n0 = std::min<size_t>(e.cand.size(), 3);
        for (size_t i = 0; i < n0 && e.cand.size() < 9; ++i) {
            Cand m = e.cand[i];
            m.merge = 1;
            e.cand.push_back(m);
        }
    }
    e.best.assign(e.cand.size(), std::numeric_limits<float>::max());
    e.samples.assign(e.cand.size(), std::vector<float>());
    e.nsamp.assign(e.cand.size(), 0);
}

int tune_begin(cvs_handle h, BasisArgs& a, int variant, bool fresh_input, TuneToken& tok)
{
    tok = TuneToken();
    const bool tunable = default_config(h, a, variant, fresh_input);
    if (!tunable || !h->autotune) return CVS_OK;
    const bool free_any = h->block_order < 0 || h->strip_rows <= 0 || (h->kind == CVS_KIND_G4 && h->g4_split < 0);
    if (!free_any) return CVS_OK;
    const int xw_pinned = h->xcd_weights;
    const int pins = (h->strip_rows > 0 ? 2 : 0) + (h->g4_split >= 0 ? 1 : 0) + (h->last.state_layout ? 4 : 0) + (h->layout << 3);
    const TuneKey key = std::make_tuple(h->device, variant | (fresh_input ? 256 : 0) | (h->sb.vmm ? 1024 : 0) | (h->kind << 12) | (pins << 16) | (a.in_u8 << 22) | (h->wg_per_cu << 24),
                                        a.rows, a.cols, xw_pinned, h->block_order, a.batch);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(h->stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
    (void)hipGetLastError();
    std::lock_guard<std::mutex> lock(g_tune_mutex);
    if (!capturing) harvest();
    if (!capturing && g_free_events.find(h->device) == g_free_events.end()) {
        // the first tunable launch on this device (it also pays for the state allocation): a few timing events ahead of need,
        // so that no later call creates one on its way to the launch
        std::vector<hipEvent_t>& pool = g_free_events[h->device];
        for (int i = 0; i < 8; ++i) {
            hipEvent_t ev = nullptr;
            if (hipEventCreate(&ev) == hipSuccess) pool.push_back(ev);
            else (void)hipGetLastError();
        }
    }
    TuneEntry& e = g_tune[key];
    if (e.cand.empty()) build_candidates(h, a, fresh_input, e);
    if (e.chosen >= 0) {
        apply(a, e.cand[e.chosen], xw_pinned != 0);
        return CVS_OK;
    }
    if (e.done_issuing || capturing || e.cand.size() < 2) return CVS_OK;   // waiting for the last samples / nothing to compare: the default runs
    const int c = e.cur;
    apply(a, e.cand[c], xw_pinned != 0);
    const bool counted = e.in_block > 0;    // a configuration's first launch after a change is not representative
    if (counted) {
        hipEvent_t e0 = take_event(h->device), e1 = take_event(h->device);
        if (e0 && e1 && hipEventRecord(e0, h->stream) == hipSuccess) {
            tok.e0 = e0;
            tok.e1 = e1;
            tok.entry = &e;
            tok.cand = c;
        } else {
            if (e0) g_free_events[h->device].push_back(e0);
            if (e1) g_free_events[h->device].push_back(e1);
            (void)hipGetLastError();
        }
    }
    if (++e.in_block == kBlock) {
        e.in_block = 0;
        if (e.round == kRounds) e.done_issuing = true;   // the closing block of the default has run
        else if (++e.cur == (int)e.cand.size()) {
            // After the last round the DEFAULT runs one more block.  A card that was idle before the first of these calls (the
            // state allocation of a new handle is enough) takes tens of launches to come back to its clock, and the launches
            // within reach of the VALU follow the clock: whoever is sampled later looks faster, the default -- first in every
            // round -- worst (a G4 handle once kept an order that is 10 % slower side by side).  With the default sampled at
            // both ends of the comparison that drift can only work against a challenger.
            e.cur = 0;
            ++e.round;
        }
    }
    return CVS_OK;
}

void tune_end(cvs_handle h, const TuneToken& tok)
{
    if (!tok.entry) return;
    std::lock_guard<std::mutex> lock(g_tune_mutex);
    TuneEntry* e = static_cast<TuneEntry*>(tok.entry);
    if (hipEventRecord(tok.e1, h->stream) == hipSuccess) {
        g_samples.push_back({e, tok.cand, tok.e0, tok.e1, h->device});
        ++e->pending;
    } else {
        (void)hipGetLastError();
        g_free_events[h->device].push_back(tok.e0);
        g_free_events[h->device].push_back(tok.e1);
    }
    if (e->done_issuing && e->pending == 0 && e->chosen < 0) e->chosen = 0;   // every sample failed to record: the default it is
}

void note_launch(cvs_handle h, const BasisArgs& a)
{
    h->last.block_order = a.block_order;
    h->last.xcd_weights = a.xcd_even * 100 + a.xcd_odd;
    h->last.strip_rows = a.strip_rows;
    h->last.nt_stores = a.nt_stores;
    h->last.g4_split = a.g4_split;
    h->last.read_ahead = a.read_ahead;
    h->last.wg_per_cu = a.wg_per_cu;
    h->last.tuning_launches = h->tuning_launches;
}

// Frame batches (cvs_pipeline_batch with state kept; BASELINE config 4), opt-in with CVS_OPT_PLACEMENT_SEARCH = 1: which
// plain block the batch state lives in decides the launch's speed by 7-9 % (profiles/r03_c4_modes_probe.txt: eight blocks of 3.2 GB
// allocated one after the other in one process, the same frames and outputs -- blocks 0 and 5..7 run the launch at 0.73 of
// the HBM roofline, blocks 1..4 at 0.67-0.68, the same in every process: runs of the VRAM allocator again, see cvs_state.cpp).
// The per-plane windows of cvs_state.cpp do not fit a batch (hundreds of small planes), but the question can be put to the
// launch itself: up to kCand plain candidate blocks are allocated, the REAL launch is timed on each (it rewrites the same
// outputs with the same values), the fastest block is kept and the others are freed.  Once per (handle, block size), never
// under stream capture, bounded by the free memory; results do not depend on it.
int batch_block_search(cvs_handle h, BasisArgs& a)
{
    constexpr int kCand = 6;
    const size_t elems = h->state_elems, bytes = elems * sizeof(float);
    if (h->placement != 1 || h->sb.vmm || a.no_state || h->batch_searched_elems == elems || bytes < ((size_t)256 << 20)) return CVS_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return CVS_OK;
    h->batch_searched_elems = elems;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return CVS_OK;
    int ncand = 1;
    while (ncand < kCand && (size_t)ncand * bytes + ((size_t)8 << 30) < free_b) ++ncand;   // candidate 0 is the block the handle has
    if (ncand < 2) return CVS_OK;
    if (!h->ev0) {   // before anything is allocated: a failure here leaves nothing behind
        HIP_TRY(h, hipEventCreate(&h->ev0));
        HIP_TRY(h, hipEventCreate(&h->ev1));
    }
    const auto t_start = std::chrono::steady_clock::now();
    std::vector<StateBlock> cand(ncand);
    cand[0] = h->sb;
    int have = 1;
    for (; have < ncand; ++have)
        if (state_block_alloc_plain(h->device, elems, cand[have]) != hipSuccess) { (void)hipGetLastError(); break; }
    const ptrdiff_t orient_off = a.orient - a.basis;
    std::vector<float> t(have, std::numeric_limits<float>::max());
    hipError_t e = hipSuccess;
    for (int round = 0; round < 2 && e == hipSuccess; ++round)          // round 0 = first touch of the fresh blocks
        for (int c = 0; c < have && e == hipSuccess; ++c) {
            a.basis = cand[c].base;
            a.orient = cand[c].base + orient_off;
            e = hipEventRecord(h->ev0, h->stream);
            for (int k = 0; k < 2 && e == hipSuccess; ++k, ++h->tuning_launches) e = launch_basis(h->kind, h->width, h->taps, a, nullptr, h->stream);
            if (e == hipSuccess) e = hipEventRecord(h->ev1, h->stream);
            if (e == hipSuccess) e = hipEventSynchronize(h->ev1);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, h->ev0, h->ev1);
            if (round > 0 && e == hipSuccess) t[c] = ms / 2;
        }
    int best = 0;
    if (e == hipSuccess)
        for (int c = 1; c < have; ++c)
            if (t[c] < t[best] * 0.98f) best = c;     // a challenger must win by 2 %
    if (std::getenv("CVS_TUNE_VERBOSE")) {
        std::fprintf(stderr, "[cvsteer] batch block search, %d candidates of %zu MiB (ms per launch):", have, bytes >> 20);
        for (int c = 0; c < have; ++c) std::fprintf(stderr, " %.4f", t[c]);
        std::fprintf(stderr, " -> candidate %d\n", best);
    }
    (void)hipStreamSynchronize(h->stream);
    for (int c = 0; c < have; ++c)
        if (c != best) state_block_free(cand[c]);
    h->sb = cand[best];
    h->state = h->sb.base;
    h->state_elems = h->sb.elems;
    a.basis = h->state;
    a.orient = h->state + orient_off;
    h->window_found = best != 0;
    h->probe_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    if (e != hipSuccess) return fail_hip(h, e, "batch block search");
    return CVS_OK;
}

}  // namespace cvs
