// cvs_tune.cpp -- how a basis launch is configured: strip height, store policy, launch order.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <map>
#include <mutex>
#include <tuple>
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
    //       hit (8 rotating images, round 2: 10-row strips 66 %, 19-row strips 57 %).  19 rows (k = 3) remain the default for
    //       images below 3 Mpix whose state the Infinity Cache holds; the tuner compares both heights there.
    //   G4  40 rows (k = 4) for the half banks: the pair launch is close to SIMD-bound, and 27-row strips filter 44 % more
    //       rows than they write against 30 % (-3..-6 %, profiles/r04_order_probe.txt).
    const int nt = 2 * h->width + 1, halo = 2 * h->width;
    const long strips_x = (cols + 63) / 64;
    const double ideal = (double)rows * (double)strips_x / 2048.0;
    long k = std::lround((ideal + halo) / nt);
    const bool big_state = h->sb.base != nullptr && h->num_frames == 1 &&
                           (size_t)rows * cols * sizeof(float) * (size_t)(h->nb + 5) >= ((size_t)256 << 20);   // states the Infinity Cache cannot hold
    // (round 6: 10 rows from 3 Mpix on whatever the state's size -- same handle, tuner off, two processes: at 1536 x 2048 and 2048^2 the full
    // setup, the fused steer and the caller pipeline run 6-17 % faster with 10 rows than with 19, the basis pass -2 ... +12 %; at 1080p the two
    // heights are level, 19 rows 2-6 % ahead for the basis pass: profiles/r06_strip_heights_mid_size.txt)
    const long kmax = h->kind == CVS_KIND_G4 ? 4 : (fresh_input || big_state || (size_t)rows * cols >= ((size_t)3 << 20)) ? 2 : 3;
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
    if (const int forced = env_opts().nt_stores; forced >= 0) return forced;   // CVS_OPTS nt_stores=0|1: A/B and tests of the streaming instances on small shapes
    const size_t state_bytes = npix * sizeof(float) * (size_t)(h->nb + 5);
    return state_bytes > (size_t)96 << 20;
}

// ---------------------------------------------------------------------------------------------------------------------
// Launch configuration: defaults, and the ONLINE comparison of a few alternatives on the caller's own launches.
//
// What a basis launch leaves open is the order in which its tiles are dealt to the chip (plain row-major, the XCD-column
// order, the dynamic tail), the strip height and the number of workgroups per CU.  Which combination is fastest depends on the
// box and -- more -- on the PROCESS, i.e. on where the allocator put the planes, so a short list (at most four) is compared
// where the code runs.  NOTHING extra is launched: while a key is undecided each of the caller's own calls runs one candidate,
// every turn of a candidate bracketed by ONE pair of events on the caller's stream; the times are read back later with
// hipEventQuery, never waited for.
//
// Round 6: candidates are compared in SUSTAINED turns.  Rounds 4-5 let the candidates take turns in blocks of five calls and
// decided with Welch's t on the pooled samples.  That was precise about the wrong quantity: a configuration that wins five-call
// blocks interleaved with other configurations need not win when it runs for good -- the dynamic tail won the interleaved
// samples of the 32 x 1080p batch in 8 of 22 bench processes and then ran 6 % behind the plain order, a five-workgroups-per-CU
// challenger for the full setup won in 13 of 13 and ran 3 % behind (profiles/r05_bench_lines*.jsonl).  Now a
// candidate's turn is a run of kTurnMin..kTurnMax consecutive calls (about kTurnMs of GPU time) timed as a WHOLE behind its first
// kLead calls, the first ROUND is burn-in (see evaluate), and a challenger replaces the default only if, after at least two
// counted rounds, its median turn is at least kGain (3 %) faster than the default's median turn and its turns -- all of two,
// all but one of three or four -- are ahead of the default's median turn by half that; challengers that are not ahead by even
// half the margin on the median are dropped after two rounds; after kMaxRounds counted rounds the default stays.  The default
// leads every round.  What no turn of a few milliseconds can see is how the card's power management answers a configuration
// that runs for seconds (build_candidates: the strip heights that were withdrawn): the margin and the short list of
// challengers with effects of 5 % and more are the protection against that.
//
// Keys are by pixel-count BUCKET (half octaves), not by exact shape: the reference's callers build one object per image of
// whatever size comes along (example/steer.cpp:86).  Samples are kept as time per pixel so that shapes of one bucket pool.  A
// candidate that does not fit the shape of the call whose turn it would be (the XCD-column order needs the column blocks to
// divide by 8) passes its turn; after kMaxUnfit passed turns it is dropped, so that a bucket of mixed shapes still comes to a
// decision (it used to stay undecided for ever).  Everything is bounded: at most kMaxRounds rounds of at most four
// turns of at most kTurnMax calls.
// Process-wide, keyed by device, kind, kernel variant, bucket, batch size, layout and what the caller pinned.
// ---------------------------------------------------------------------------------------------------------------------
struct Cand {
    int order, strip;
    int merge = 0;   // the default's grouping of the G2 state planes (default_config), carried along: never varied any more
    int wg = -1;     // workgroups per CU; -1 = what default_config's rule says
    bool operator==(const Cand& o) const { return order == o.order && strip == o.strip && merge == o.merge && wg == o.wg; }
};

struct TuneEntry {
    std::vector<Cand> cand;        // cand[0] = the default
    std::vector<std::vector<float>> turn;      // per candidate: ns per pixel of its turn of the round in progress (one value)
    std::vector<std::vector<float>> medians;   // per candidate: ns per pixel of every counted turn
    std::vector<char> dropped;     // challengers out of the race
    std::vector<int> unfit;        // turns a candidate has passed because the call's shape did not fit it
    int cur = 0, in_turn = 0, round = 0;
    int turn_len = 0;              // calls per turn in this round
    int pending = 0;               // turns recorded but not read back yet
    int chosen = -1;               // index into cand once decided
    bool round_complete = false;   // a whole round has been issued since the last evaluation
    double ms_per_call = 0;        // time per call of the default's last turn (sets the length of the next round's turns)
    // the turn in progress: ONE event pair around its counted calls (see tune_begin)
    hipEvent_t t_e0 = nullptr;
    hipStream_t t_stream = nullptr;
    double t_npix = 0;
    int t_calls = 0, t_device = 0;
    bool t_valid = false;
};

struct Sample {   // one TURN: the events around its counted calls
    TuneEntry* entry;
    int cand;
    double npix;       // pixels of all the calls between the events
    int calls;
    hipEvent_t e0, e1;
    int device;
};

typedef std::tuple<int, int, int, int, int> TuneKey;
static std::mutex g_tune_mutex;
static std::map<TuneKey, TuneEntry> g_tune;          // node-based: TuneEntry* stays valid
static std::vector<Sample> g_samples;                // in flight
static std::map<int, std::vector<hipEvent_t>> g_free_events;   // per device: timing events are recycled, never destroyed while the process lives

constexpr int kLead = 5;                        // calls at the head of a turn that are not counted
constexpr int kTurnMin = 20, kTurnMax = 100;    // calls per turn
constexpr double kTurnMs = 12.0;                // ... about this much GPU time
constexpr int kBurnIn = 1, kMinRounds = 2, kMaxRounds = 4, kMaxUnfit = 3;   // rounds: burn-in, then at least / at most this many that count
constexpr float kGain = 0.03f;   // (2 % until the strip-height challengers came: a 9-row strip won its turns by 2.1 % -- per-call event time -- and ran 1-1.6 % BEHIND
                                 // back to back, where its 11 % more workgroups also cost between the launches: profiles/r06_tuner_value_probe.txt, session 24)

static void apply(BasisArgs& a, const Cand& c)
{
    a.block_order = c.order;
    a.strip_rows = c.strip;
    a.merge_orient = c.merge;
    if (c.wg >= 0 && env_opts().wgcap < 0) a.wg_per_cu = c.wg;   // (a cap forced through CVS_OPTS wins over a remembered challenger)
}

// a configuration decided for the bucket must fit THIS launch
static bool fits(const BasisArgs& a, const Cand& c)
{
    const int grid_x = ((a.cols + 63) / 64 + 3) / 4;
    if (c.order == kOrderXcdColumns && (grid_x % 8 != 0 || a.batch != 0)) return false;
    if (c.order == kOrderDynamic && !a.tile_ctr) return false;
    return true;
}

static float median_of(std::vector<float> v)
{
    std::sort(v.begin(), v.end());
    return v.size() % 2 ? v[v.size() / 2] : 0.5f * (v[v.size() / 2 - 1] + v[v.size() / 2]);
}

// a round is complete and all its samples are in: close the turns, then decide or ask for another round.  g_tune_mutex held.
static void evaluate(TuneEntry& e)
{
    for (size_t c = 0; c < e.cand.size(); ++c) {
        // Round 1 is burn-in and does not count: a configuration's FIRST turn in a process runs 15-25 % slower than its later ones for
        // tens of launches (verbose logs of four bench processes, profiles/r06_tuner_first_turn.txt: default 0.01663 then 0.01419 ns/pix,
        // dynamic tail 0.01000 then 0.00803) -- counted, it made a cold default lose to a warm challenger, or a cold challenger drop out
        if (e.round > kBurnIn && !e.turn[c].empty()) e.medians[c].push_back(median_of(e.turn[c]));   // (one value per turn; a turn that could not be timed does not count)
        e.turn[c].clear();
    }
    const int counted = e.round - kBurnIn;   // rounds that count
    int decision = -1;
    if (counted < 1) return;
    if (e.medians[0].empty()) {   // every sample of the default failed to record: nothing to hold a challenger against
        if (counted >= kMaxRounds) decision = 0;
    } else if (counted >= kMinRounds) {
        const float d_mid = median_of(e.medians[0]);
        int best = -1, live = 0;
        float mbest = std::numeric_limits<float>::max();
        for (size_t c = 1; c < e.cand.size(); ++c) {
            if (e.dropped[c]) continue;
            if ((int)e.medians[c].size() < kMinRounds) {   // passed or lost its turns
                if (counted >= kMaxRounds) e.dropped[c] = 1;
                else ++live;
                continue;
            }
            const float m = median_of(e.medians[c]);
            if (m > d_mid * (1.0f - 0.5f * kGain)) { e.dropped[c] = 1; continue; }   // not ahead by even half the margin on the median: out
            ++live;
            // kept: its median turn kGain ahead of the default's median turn, and its turns -- all of two, all but one of three or four (a
            // turn that ran into somebody else's burst) -- ahead of the default's median turn by half that
            int wins = 0;
            for (float v : e.medians[c]) wins += v <= d_mid * (1.0f - 0.5f * kGain) ? 1 : 0;
            const int nt_ = (int)e.medians[c].size();
            if (m <= d_mid * (1.0f - kGain) && wins >= nt_ - (nt_ >= 3 ? 1 : 0) && m < mbest) { mbest = m; best = (int)c; }
        }
        if (best >= 0) decision = best;
        else if (live == 0 || counted >= kMaxRounds) decision = 0;
    }
    if (decision >= 0) {
        e.chosen = decision;
        if (env_opts().verbose) {
            std::fprintf(stderr, "[cvsteer] tuned on the caller's launches (ns per pixel of each sustained turn):");
            for (size_t c = 0; c < e.cand.size(); ++c) {
                std::fprintf(stderr, " (order %d, strip %d, merged %d, wg %d%s)", e.cand[c].order, e.cand[c].strip, e.cand[c].merge, e.cand[c].wg, e.dropped[c] ? ", dropped" : "");
                for (float m : e.medians[c]) std::fprintf(stderr, " %.5f", m);
            }
            std::fprintf(stderr, " -> candidate %d after %d rounds of %d-call turns\n", decision, e.round, e.turn_len);
        }
    }
}

// read back every sample whose launch has finished (never waits); evaluate entries whose round is complete.  g_tune_mutex held.
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
            if (e.turn[sm.cand].size() < (size_t)kTurnMax) e.turn[sm.cand].push_back((float)(ms * 1e6 / sm.npix));
            if (sm.cand == 0) e.ms_per_call = ms / std::max(1, sm.calls);
        }
        (void)hipGetLastError();
        --e.pending;
        g_free_events[sm.device].push_back(sm.e0);
        g_free_events[sm.device].push_back(sm.e1);
        g_samples[i] = g_samples.back();
        g_samples.pop_back();
        if (e.round_complete && e.pending == 0 && e.chosen < 0) {
            e.round_complete = false;
            evaluate(e);
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
static bool default_config(cvs_handle h, BasisArgs& a, int variant)
{
    // Workgroups per CU.  With the input rows going straight into LDS and the packed arithmetic (round 5) the G2 kernels need 75-90
    // VGPRs and six workgroups would share a CU; every resident wave is one more write front in HBM, and the launches stream faster
    // with fewer (two processes on one box after the packed arithmetic, profiles/r05_occupancy_cap.txt, fraction of the HBM roofline):
    //   basis pass          four per CU   0.821 -> 0.845 resident, 0.722 -> 0.741 new images (4096^2); 8192^2 left alone (0.86-0.88)
    //   full setup          three         0.74 -> 0.797, new images 0.70 -> 0.745
    //   fused steer         three         0.81-0.82 (four, five) -> 0.833; 8192^2: four (new images 0.732 -> 0.757 and 0.728 -> 0.75 on two boxes;
    //                                     resident +2 % on one box, -2.5 % on the next)
    //   caller pipeline     three         0.71 -> 0.73
    //   pyramid level that also emits the next level: three at any size (five-level pyramid of 8192^2: 0.607 -> 0.67)
    // G4 does not care (0.70-0.71 at two to five), frame batches are left alone (+-1.5 %).  CVS_OPTS wgcap=N overrides (0 = none).
    const size_t npix_cfg = (size_t)a.rows * a.cols;
    a.wg_per_cu = 0;
    if (h->kind == CVS_KIND_G2 && a.batch == 0 && npix_cfg >= ((size_t)2 << 20)) {
        const bool mid = npix_cfg < ((size_t)32 << 20);
        if (a.pyr_out) a.wg_per_cu = 3;
        else if (variant == 0) a.wg_per_cu = mid ? 4 : 0;
        else if (variant == 2) a.wg_per_cu = mid ? 3 : 4;
        else if (variant == 1 || variant == 5) a.wg_per_cu = mid ? 3 : 0;
    }
    // Grouping of the G2 state planes.  Single-image launches that write the orientation planes too (full setup, caller pipeline) put all
    // TWELVE planes into one row-interleaved group -- one write sweep instead of two: round 6, six processes on one box, same handle,
    // sustained (profiles/r06_m4_layout_ab.txt): full setup 0.765 (two groups, three workgroups per CU) -> 0.82 (one group, uncapped) in 6
    // of 6, caller pipeline 0.744 -> 0.786 (one group, three per CU) in 6 of 6; round 5 over twelve processes on two boxes: +6-7 % in 7, never
    // worse than -0.6 % (profiles/r05_tuner_value_probe.txt).  With one group the full setup wants no occupancy cap (0.768 at three, 0.81 at
    // four, 0.82 at five = uncapped), the pipeline keeps three (0.786 against 0.745 / 0.738 at four / five).  CVS_OPT_STATE_LAYOUT = 3 pins
    // the two-group form.  Basis-only launches, batches and G4 always use the groups of layout_state.
    a.merge_orient = ((h->layout == 1 || h->layout == 2) && h->kind == CVS_KIND_G2 && a.orient && !a.no_state && a.batch == 0 &&
                      state_merge_fits(h, a.rows, round_up((size_t)a.cols, 64))) ? 1 : 0;
    if (a.merge_orient && !a.pipe && a.wg_per_cu == 3 && !a.pyr_out) a.wg_per_cu = 0;
    if (const int forced = env_opts().wgcap; forced >= 0) a.wg_per_cu = forced;
    const bool fast = basis_fast_path(h->kind, h->width, h->taps);
    const bool big = (size_t)a.rows * a.cols >= ((size_t)1 << 20);
    // the plain row-major order: with the row-interleaved state it is within a few per cent of the best order on every box
    // and every variant measured, resident image or fresh (profiles/r04_layout_probe.txt); what beats it is box-dependent.
    // (Round 6 held the dynamic tail against it for SECONDS -- regions of 1.5 s, same handle, alternating -- for single images of 8 Mpix and
    // more whose launch writes the orientation planes too: ahead on six boxes, full setup at 4096^2 +1.4 ... +3.4 %, pipeline +0.9 ... +3.5 %,
    // on new images 0 ... +1 % (profiles/r06_long_regions.txt); made the default, the seventh box read 0.786 with it where the plain order
    // reads 0.80 and its tuner went back to the plain order.  It stays a challenger.)
    a.block_order = h->block_order >= 0 ? h->block_order : 0;
    return fast && big;
}

static void build_candidates(cvs_handle h, const BasisArgs& a, bool fresh_input, TuneEntry& e)
{
    const bool free_order = h->block_order < 0;
    const bool free_strip = h->strip_rows <= 0 && (a.batch == 0 || !a.no_state);
    Cand def{a.block_order, a.strip_rows};
    def.merge = a.merge_orient;
    e.cand.assign(1, def);
    auto add = [&](Cand c) {
        if (c.order == kOrderDynamic && !a.tile_ctr) return;
        if (c.order != def.order && !free_order) return;
        if (c.strip != def.strip && !free_strip) return;
        c.merge = def.merge;
        if (!fits(a, c)) return;
        for (const Cand& k : e.cand)
            if (k == c) return;
        if (e.cand.size() < 4) e.cand.push_back(c);
    };
    const int nt = 2 * h->width + 1, halo = 2 * h->width;
    const int sr_short = 2 * nt - halo, sr_tall = 3 * nt - halo;
    // The caller pipeline runs capped at three workgroups per CU by the rule; whether three or five is faster flips from box to box
    // (0.737 / 0.711 on one, 0.820 / 0.868 on the next; profiles/r05_occupancy_cap.txt) -- so five is its first challenger, kept in 4 of
    // 13 bench processes at 0.80-0.82 where the others sit at 0.76-0.77.  NOT for the full setup: there the challenger won the
    // interleaved samples in 13 of 13 bench processes and then ran 3 % behind the rule in the 8 of them whose plane groups lie badly
    // (0.74 against 0.765; profiles/r05_bench_lines*.jsonl), although it is 5-6 % ahead sustained in processes with nothing else
    // allocated (profiles/r05_tuner_value_probe.txt).  The basis pass and the fused steer keep their rule on every box measured.
    if (h->kind == CVS_KIND_G2 && a.batch == 0 && a.wg_per_cu == 3 && a.pipe && env_opts().wgcap < 0) {
        Cand c = def;
        c.wg = 5;
        e.cand.push_back(c);
    }
    if (h->kind == CVS_KIND_G2) {
        const bool multi = a.orient != nullptr || a.pipe;
        const bool large = (size_t)a.rows * a.cols >= ((size_t)8 << 20);
        // Shorter strips than the 10-row default (round 6).  Which height streams best depends on the SHAPE -- the row pitch decides which
        // addresses the ~64 row bands in flight write at the same time -- and flips between shapes of one size class: same handle, tuner
        // off, two processes (profiles/r06_strip_heights_fine.txt): launches that write the orientation planes too run 4-12 % faster with
        // 7 rows at 1536 x 2048, 2048^2, 2160 x 3840, 3000 x 4000 and 4000 x 6000 and 8 % SLOWER at 4096^2.  No rule in sight: a challenger.
        // NOT for the basis pass and the fused steer: their 9- and 8-row strips gain 3-11 % at some shapes on some boxes and were offered for
        // a while -- until an 8-row strip that was 3 % ahead over its 12 ms turns took the HEADLINE launch from 0.835 to 0.798 in one bench
        // process of eight: run for seconds it draws the card's clock down (1.88 instead of 2.07 GHz under the same 1400 W), which no turn
        // of a few milliseconds can see (profiles/r06_bench_lines_earlier_in_the_round.jsonl, the last line).  Margins of a few per cent on
        // a launch that runs at the power cap are not decidable on the caller's own calls; margins of 6-10 % (7 rows, below) are.
        if (a.batch == 0 && multi && def.strip == sr_short) add({def.order, 7});
        if (fresh_input) {
            // a stream of new images: short strips are a must (the halo rows of vertically adjacent strips only hit in cache when
            // those strips run close in time); the plain order, the XCD-column order and the dynamic tail are within 1 % of each
            // other in sustained runs on large images, so only smaller ones compare them.  (Rounds 4 also offered a pure-read pass
            // over the image in front of the launch; the first waves of the launch do that themselves now: BasisArgs::warm_k.)
            if (!large) {
                add({kOrderDynamic, sr_short});
                add({kOrderXcdColumns, sr_short});
            }
        } else {
            // Resident image.  What has beaten the default (plain order, 10-row strips) by more than 2 % in SUSTAINED side-by-side
            // runs on one handle (profiles/r04_order_probe_sustained.txt): for launches that also write the orientation planes
            // (12 / 20 planes) the dynamic tail (+3 %); for the basis pass and the fused steer on a large image NOTHING (all orders within
            // 1 %; strip heights: see above), so those are not tuned at all.
            if (multi || !large) add({kOrderDynamic, sr_short});
            if (!large) {   // smaller images (below 3 Mpix the default is the 19-row strip): both heights, the column order
                add({0, sr_short});
                add({0, sr_tall});
                add({kOrderXcdColumns, sr_short});
            }
        }
    } else {
        // G4: the dynamic tail (+6 % on one box, level on the others)
        add({kOrderDynamic, def.strip});
    }
    e.turn.assign(e.cand.size(), std::vector<float>());
    e.medians.assign(e.cand.size(), std::vector<float>());
    e.dropped.assign(e.cand.size(), 0);
    e.unfit.assign(e.cand.size(), 0);
    e.turn_len = kTurnMin;
}

// half-octave bucket of the pixel count
static int pix_bucket(int rows, int cols) { return (int)std::floor(2.0 * std::log2((double)rows * (double)cols)); }

int tune_begin(cvs_handle h, BasisArgs& a, int variant, bool fresh_input, TuneToken& tok)
{
    tok = TuneToken();
    h->last.tuned = 0;
    h->last.tune_state = 0;
    const bool tunable = default_config(h, a, variant);
    if (!tunable || !h->autotune) return CVS_OK;
    if (h->block_order >= 0 && h->strip_rows > 0) return CVS_OK;   // everything pinned
    const int pins = (h->strip_rows > 0 ? 2 : 0) + (h->last.state_layout ? 4 : 0) + (h->layout << 3) + ((h->block_order + 1) ? 32 : 0);
    const TuneKey key = std::make_tuple(h->device, variant | (fresh_input ? 256 : 0) | (h->kind << 12) | (pins << 16) | (a.in_u8 << 24),
                                        pix_bucket(a.rows, a.cols), h->block_order, a.batch);
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
    if (e.round_complete && e.pending == 0 && e.chosen < 0) {   // the round's last sample was read back by an earlier call
        e.round_complete = false;
        evaluate(e);
    }
    if (e.cand.size() >= 2 && e.chosen >= 0) {
        h->last.tune_state = 2;
        if (e.chosen > 0 && fits(a, e.cand[e.chosen])) {
            apply(a, e.cand[e.chosen]);
            h->last.tuned = 1;
        }
        return CVS_OK;
    }
    if (e.cand.size() < 2) return CVS_OK;   // nothing to compare for this kind of launch: the default, tune_state 0
    h->last.tune_state = 1;
    if (capturing || e.round_complete) return CVS_OK;   // under capture / waiting for the round's last samples: the default runs
    // whose turn is it?  A candidate that does not fit this call's shape passes (at the head of its turn only; in mid-turn the call
    // runs the default uncounted); after kMaxUnfit passed turns it is out of the race.
    auto next_cand = [&]() {
        e.in_turn = 0;
        do {
            ++e.cur;
        } while (e.cur < (int)e.cand.size() && e.dropped[e.cur]);
        if (e.cur >= (int)e.cand.size()) {   // the round is complete; the default leads the next one
            e.cur = 0;
            ++e.round;
            e.round_complete = true;
        }
    };
    while (e.in_turn == 0 && e.cur > 0 && !fits(a, e.cand[e.cur])) {
        if (++e.unfit[e.cur] >= kMaxUnfit) e.dropped[e.cur] = 1;
        next_cand();
        if (e.round_complete) return CVS_OK;
    }
    if (e.cur == 0 && e.in_turn == 0)   // a new round: its turns last about kTurnMs of GPU time each
        e.turn_len = e.ms_per_call > 0 ? std::max(kTurnMin, std::min(kTurnMax, kLead + (int)std::ceil(kTurnMs / e.ms_per_call))) : kTurnMin;
    const int c = e.cur;
    const bool fit = fits(a, e.cand[c]);
    if (fit) apply(a, e.cand[c]);
    // ONE event pair per turn, around its counted calls (everything behind the first kLead: a configuration's first launches after a change
    // are not representative).  Until late in round 6 every call had its own pair; that measures how long a kernel ISSUES, not what a run of
    // them delivers: the stores a kernel leaves behind drain in the gap the events make, and a configuration that issues faster but leaves
    // more behind (shorter strips: more workgroups) won per call -- 2-3 % -- and ran 1-6 % SLOWER back to back (the headline launch itself, once:
    // profiles/r06_tuner_value_probe.txt, session 28).  A turn whose calls come from more than one stream, or that met a shape its candidate
    // does not fit, cannot be timed this way and does not count.
    if (e.in_turn >= kLead) {
        if (e.in_turn == kLead) {   // the first counted call opens the turn
            if (e.t_e0) g_free_events[e.t_device].push_back(e.t_e0);
            e.t_e0 = take_event(h->device);
            e.t_device = h->device;
            e.t_stream = h->stream;
            e.t_npix = 0;
            e.t_calls = 0;
            e.t_valid = fit && e.t_e0 && hipEventRecord(e.t_e0, h->stream) == hipSuccess;
            if (!e.t_valid) (void)hipGetLastError();
        }
        if (!fit || h->stream != e.t_stream) e.t_valid = false;
        e.t_npix += (double)a.rows * (double)a.cols * (double)(a.batch > 0 ? a.batch : 1);
        ++e.t_calls;
        if (e.in_turn == e.turn_len - 1) {   // the last call closes it: tune_end records the second event behind this launch
            hipEvent_t e1 = e.t_valid ? take_event(h->device) : nullptr;
            if (e1) {
                tok.e0 = e.t_e0;
                tok.e1 = e1;
                tok.entry = &e;
                tok.cand = c;
                tok.npix = e.t_npix;
                tok.calls = e.t_calls;
                e.t_e0 = nullptr;
            } else if (e.t_e0) {
                g_free_events[e.t_device].push_back(e.t_e0);
                e.t_e0 = nullptr;
            }
        }
    }
    if (++e.in_turn >= e.turn_len) next_cand();
    return CVS_OK;
}

void tune_end(cvs_handle h, const TuneToken& tok)
{
    if (!tok.entry) return;
    std::lock_guard<std::mutex> lock(g_tune_mutex);
    TuneEntry* e = static_cast<TuneEntry*>(tok.entry);
    if (hipEventRecord(tok.e1, h->stream) == hipSuccess) {
        g_samples.push_back({e, tok.cand, tok.npix, tok.calls, tok.e0, tok.e1, h->device});
        ++e->pending;
    } else {
        (void)hipGetLastError();
        g_free_events[h->device].push_back(tok.e0);
        g_free_events[h->device].push_back(tok.e1);
    }
}

void note_launch(cvs_handle h, const BasisArgs& a)
{
    h->last.block_order = a.block_order;
    h->last.strip_rows = a.strip_rows;
    h->last.nt_stores = a.nt_stores;
    h->last.warm = a.warm_k;
    h->last.wg_per_cu = a.wg_per_cu;
    h->last.tuning_launches = 0;   // nothing is ever launched beyond the caller's own calls
}

}  // namespace cvs
