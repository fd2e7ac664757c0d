// host_logic_san.cpp -- the host logic of libcvsteer_hip.so that needs no device, under AddressSanitizer +
// UndefinedBehaviorSanitizer: argument checks (check_plane), the overlap rules of include/cvsteer_hip.h (planes_overlap against
// a byte-for-byte model on random views), the CVS_OPTS parser on hostile strings, the state layout arithmetic (layout_state on
// every kind / size / grouping: offsets inside the block, no two planes sharing an element) and the tap generator.  A cvs_context is
// a plain struct: it is built here without a HIP call; no entry point that touches the device is called.
// Built and run by tools/run_sanitizers.sh and tests/test_sanitizers_cpu.py:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -Iinclude -Icvsteer_amd/csrc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
//       tests/cpp/host_logic_san.cpp cvsteer_amd/csrc/{cvs_handle,cvs_tune,cvs_state,cvs_taps}.cpp -L/opt/rocm/lib -lamdhip64 -lpthread
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "cvs_context.h"

namespace cvs {
// the two kernel-side symbols the host objects refer to (never reached here)
hipError_t launch_u8_to_f32(const uint8_t*, size_t, int, int, float*, size_t, hipStream_t) { return hipErrorUnknown; }
bool basis_fast_path(int, int, const float (*)[kMaxTaps]) { return true; }
}  // namespace cvs

static unsigned long long rs = 0x2545f4914f6cdd1dull;
static unsigned rnd()
{
    rs ^= rs << 13;
    rs ^= rs >> 7;
    rs ^= rs << 17;
    return (unsigned)(rs >> 20);
}
#define REQUIRE(c)                                                          \
    do {                                                                    \
        if (!(c)) {                                                         \
            std::fprintf(stderr, "host_logic_san: %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

int main()
{
    using namespace cvs;
    cvs_context ctx;
    cvs_handle h = &ctx;
    // ---- check_plane ----
    alignas(16) static float buf[64 * 64];
    cvs_plane ok{buf, 8, 8, 8 * sizeof(float), CVS_MEM_HOST};
    REQUIRE(check_plane(h, &ok, "p") == CVS_OK);
    REQUIRE(check_plane(h, nullptr, "p") == CVS_E_BADARG);
    cvs_plane p = ok; p.rows = 0; REQUIRE(check_plane(h, &p, "p") == CVS_E_SIZE);
    p = ok; p.cols = -3; REQUIRE(check_plane(h, &p, "p") == CVS_E_SIZE);
    p = ok; p.data = nullptr; REQUIRE(check_plane(h, &p, "p") == CVS_E_BADARG);
    p = ok; p.step = 7 * sizeof(float); REQUIRE(check_plane(h, &p, "p") == CVS_E_SIZE);
    p = ok; p.step = 8 * sizeof(float) + 2; REQUIRE(check_plane(h, &p, "p") == CVS_E_SIZE);
    p = ok; p.mem = 7; REQUIRE(check_plane(h, &p, "p") == CVS_E_BADARG);
    p = ok; p.mem = CVS_MEM_HOST | CVS_DEPTH_U8; REQUIRE(check_plane(h, &p, "p") == CVS_E_BADARG && check_plane(h, &p, "p", true) == CVS_OK);
    p = ok; p.mem = CVS_MEM_HOST | 0x1000; REQUIRE(check_plane(h, &p, "p") == CVS_E_BADARG);
    p = ok; p.data = reinterpret_cast<float*>(reinterpret_cast<char*>(buf) + 2); REQUIRE(check_plane(h, &p, "p") == CVS_E_BADARG);
    p = ok; p.mem = CVS_MEM_HOST | CVS_DEPTH_U8; p.step = 7; REQUIRE(check_plane(h, &p, "p", true) == CVS_E_SIZE);

    // ---- planes_overlap against a byte model: random f32 / 8-bit views into one 4 KiB arena ----
    static unsigned char arena[4096];
    for (int it = 0; it < 20000; ++it) {
        cvs_plane v[2];
        std::set<size_t> bytes[2];
        for (int k = 0; k < 2; ++k) {
            const bool u8 = rnd() & 1;
            const int es = u8 ? 1 : 4;
            const int rows = 1 + rnd() % 6, cols = 1 + rnd() % 6;
            const size_t step = (size_t)(cols + rnd() % 5) * es;
            const size_t span = (size_t)(rows - 1) * step + (size_t)cols * es;
            const size_t off = (rnd() % (sizeof arena - span)) / es * es;
            v[k] = cvs_plane{reinterpret_cast<float*>(arena + off), rows, cols, step, CVS_MEM_HOST | (u8 ? CVS_DEPTH_U8 : 0)};
            for (int r = 0; r < rows; ++r)
                for (size_t b = 0; b < (size_t)cols * es; ++b) bytes[k].insert(off + r * step + b);
        }
        bool share = false;
        for (size_t b : bytes[0]) share = share || bytes[1].count(b);
        const bool said = planes_overlap(&v[0], &v[1]);
        REQUIRE(said == planes_overlap(&v[1], &v[0]));
        if (share) REQUIRE(said);                                   // a shared byte is never missed
        if (v[0].step == v[1].step && !share) REQUIRE(!said);       // equal steps are compared exactly
    }
    const cvs_plane* outs[3] = {&ok, nullptr, &ok};
    REQUIRE(check_no_overlap(h, nullptr, outs, 3) == CVS_E_BADARG);   // an output given twice
    REQUIRE(check_point_overlaps(h, {&ok}, {&ok}) == CVS_OK);          // in place: an output may BE an input

    // ---- CVS_OPTS: hostile strings ----
    const char* opts[] = {"", ",", "=", "autotune", "autotune=", "=3", "autotune=0,layout=9,pyr_strip=-4,batch_ways=99999999999999999999,read_ahead=x",
                          "nt_stores=1,,,,verbose=1,pool_mb=-5", "unknown=1", "autotune=0,autotune=1", ",,,,=,=,=", "layout=2\n", "pool_mb=18446744073709551616"};
    for (const char* o : opts) {
        setenv("CVS_OPTS", o, 1);
        const EnvOpts e = env_opts();
        REQUIRE(e.layout >= -1 && e.layout <= 3 && e.autotune >= -1 && e.autotune <= 1 && e.batch_ways >= -1);
    }
    for (int it = 0; it < 3000; ++it) {
        std::string s;
        const char alphabet[] = "autonelyprsbchwdv_=,0123456789-x \n";
        for (int k = rnd() % 48; k > 0; --k) s += alphabet[rnd() % (sizeof alphabet - 1)];
        setenv("CVS_OPTS", s.c_str(), 1);
        (void)env_opts();
    }
    unsetenv("CVS_OPTS");

    // ---- state layout: every plane inside the block, no two planes sharing an element ----
    for (int kind : {CVS_KIND_G2, CVS_KIND_G4})
        for (int layout : {0, 1, 2})
            for (int merged : {0, 1})
                for (int it = 0; it < 40; ++it) {
                    cvs_context c;
                    c.kind = kind;
                    c.nb = kind == CVS_KIND_G2 ? 7 : 11;
                    c.width = kind == CVS_KIND_G2 ? 4 : 6;
                    c.layout = layout;
                    c.rows = 1 + rnd() % 40;
                    c.cols = 1 + rnd() % 300;
                    c.dense_pitch = round_up((size_t)c.cols, 64);
                    c.layout_stride = round_up(c.dense_pitch * c.rows, 64);
                    const size_t elems = c.layout_stride * (size_t)(c.nb + 5);
                    std::vector<float> block(elems);
                    c.state = block.data();
                    c.state_elems = elems;
                    layout_state(&c, merged != 0);
                    REQUIRE(c.frame_stride <= elems);
                    std::vector<unsigned char> used(elems, 0);
                    for (int idx = 0; idx < c.nb + 5; ++idx) {
                        const PlaneRef r = state_ref(&c, idx);
                        REQUIRE(r.p >= block.data());
                        for (int y = 0; y < c.rows; ++y)
                            for (int x = 0; x < c.cols; ++x) {
                                const size_t o = (size_t)(r.p - block.data()) + (size_t)y * r.pitch + x;
                                REQUIRE(o < elems);
                                REQUIRE(!used[o]);
                                used[o] = 1;
                            }
                    }
                    BasisArgs a{};
                    fill_state_args(&c, a, true);
                    REQUIRE(a.basis == block.data() && a.state_bytes == c.frame_stride * sizeof(float));
                    c.strip_rows = 0;
                    const int sr = default_strip_rows(&c, 1 + rnd() % 9000, 1 + rnd() % 9000, rnd() & 1);
                    REQUIRE(sr >= 2 * (2 * c.width + 1) - 2 * c.width && sr <= 4 * (2 * c.width + 1));
                }

    // ---- taps ----
    float t[kMaxTaps];
    REQUIRE(host_make_taps(CVS_KIND_G2, 0, 4, 0.67f, t) == 0 && t[4] < 0.f && t[0] == t[8]);
    REQUIRE(host_make_taps(CVS_KIND_G4, 10, 6, 0.5f, t) == 0);
    REQUIRE(host_make_taps(7, 0, 4, 0.67f, t) != 0 && host_make_taps(CVS_KIND_G2, 7, 4, 0.67f, t) != 0 && host_make_taps(CVS_KIND_G2, 0, kMaxWidth + 1, 0.67f, t) != 0);
    float w[kMaxBasis];
    for (float th : {0.f, 0.3f, -1.2f, 3.1415927f}) REQUIRE(host_steer_weights(CVS_KIND_G2, th, w) == 0 && host_steer_weights(CVS_KIND_G4, th, w) == 0);
    std::printf("host_logic_san: argument checks, 20000 overlap cases, CVS_OPTS fuzz, 480 state layouts, taps: no sanitizer report\n");
    return 0;
}
