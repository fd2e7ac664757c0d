// tests/cpp/test_protected.cpp -- the protected statics of fa::SteerableFilters, reached the way user code
// reaches them in the reference: through a subclass (cvsteer/SteerableFilters.h:41-50 declares `create` and
// `wrap` protected; SteerableFiltersG2/G4 are such subclasses).
//   create: k[i + width] = f(float(i) * spacing), i = -width..width      (SteerableFilters.cpp:33-42)
//   wrap:   output = angle > pi ? angle - 2 pi : angle                    (SteerableFilters.cpp:46-51)
// wrap runs on the GPU (cvs_wrap); create is host math, as in the reference.
#include <cvsteer/SteerableFilters.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <stdexcept>

namespace {

float gauss(float x) { return std::exp(-x * x); }
float odd_cubic(float x) { return 0.5f * x * x * x - x; }

// a user-defined filter family: nothing but the base class surface
class Probe : public fa::SteerableFilters {
public:
    Probe() : fa::SteerableFilters(2 /* CVS_KIND_G2 */, 4, 0.67f, 0) {}
    void setup(const fa::Mat1f&) {}
    void steer(float, fa::Mat1f&, fa::Mat1f&) {}
    static fa::Mat1f make(int width, float spacing, KernelType f) { return create(width, spacing, f); }
    static void wrapped(const fa::Mat1f& a, fa::Mat1f& o) { wrap(a, o); }
};

bool same_bits(float a, float b) { return std::memcmp(&a, &b, sizeof(float)) == 0; }

}  // namespace

int main()
{
    int failures = 0;
#define EXPECT(cond)                                                          \
    do {                                                                      \
        if (!(cond)) { std::printf("FAILED: %s (line %d)\n", #cond, __LINE__); ++failures; } \
    } while (0)
    try {
        // ---- create ----
        const int widths[3] = {4, 6, 1};
        const float spacings[3] = {0.67f, 0.5f, 1.25f};
        for (int t = 0; t < 3; ++t) {
            const int w = widths[t];
            fa::Mat1f k = Probe::make(w, spacings[t], gauss), o = Probe::make(w, spacings[t], odd_cubic);
            EXPECT(k.rows == 1 && k.cols == 2 * w + 1 && o.rows == 1 && o.cols == 2 * w + 1);
            for (int i = -w; i <= w; ++i) {
                EXPECT(same_bits(k(0, i + w), gauss(float(i) * spacings[t])));
                EXPECT(same_bits(o(0, i + w), odd_cubic(float(i) * spacings[t])));
            }
            EXPECT(same_bits(k(0, w), 1.0f));
        }
        // ---- wrap ----
        const float pi = 3.14159274f, two_pi = 6.2831855f;  // M_PI, 2 M_PI narrowed to f32 (what the Mat expression uses)
        const float nan = std::numeric_limits<float>::quiet_NaN(), inf = std::numeric_limits<float>::infinity();
        const float vals[] = {0.f, -0.f, pi, std::nextafter(pi, 4.f), std::nextafter(pi, 0.f), -pi, two_pi, std::nextafter(two_pi, 7.f),
                              3.f * pi, -3.f * pi, 1.0f, -1.0f, 4.0f, 5.5f, 1e30f, -1e30f, inf, -inf, nan};
        const int n = (int)(sizeof(vals) / sizeof(vals[0]));
        fa::Mat1f a(3, n), out;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < n; ++c) a(r, c) = r == 2 ? vals[n - 1 - c] : vals[c];
        Probe::wrapped(a, out);
        EXPECT(out.rows == 3 && out.cols == n);
        for (int r = 0; r < 3 && out.rows == 3; ++r)
            for (int c = 0; c < n; ++c) {
                const float v = a(r, c), want = v > pi ? v - two_pi : v;
                const float got = out(r, c);
                EXPECT((want != want && got != got) || same_bits(got, want));
            }
        // aliased call, as the reference makes it (wrap(m_theta, m_theta), SteerableFiltersG2.cpp:98)
        fa::Mat1f b = a.clone();
        Probe::wrapped(b, b);
        for (int c = 0; c < n; ++c) EXPECT((b(0, c) != b(0, c) && out(0, c) != out(0, c)) || same_bits(b(0, c), out(0, c)));
        // the object itself is usable (the base ctor made a handle)
        Probe p;
        EXPECT(p.handle() != 0 && p.device() == 0);
        p.synchronize();
    } catch (const std::exception& ex) {
        std::printf("EXCEPTION: %s\n", ex.what());
        return 1;
    }
    std::printf(failures ? "cvsteer.protected FAILED (%d)\n" : "cvsteer.protected OK\n", failures);
    return failures ? 1 : 0;
}
