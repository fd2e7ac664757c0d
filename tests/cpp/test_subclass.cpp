// tests/cpp/test_subclass.cpp -- code that EXTENDS the reference, written the way the reference's own headers are
// written: inside _STEER_BEGIN / _STEER_END (cvsteer/cvsteer.h:12-15), deriving from SteerableFiltersG2 / G4 and reading
// the protected plane members by their reference names (m_g2a..m_h2d, m_c1..m_c3, m_theta: SteerableFiltersG2.h:62-66;
// m_g4a..m_h4f: SteerableFiltersG4.h:53-54).  With the facade those members are host copies of the GPU state:
// filled after every setup() on a subclass object, or by syncMembers() inside a subclass constructor.
#include <cvsteer/SteerableFiltersG2.h>
#include <cvsteer/SteerableFiltersG4.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

_STEER_BEGIN

// a downstream class in the style of the reference: oriented energy from the first basis pair, read from the members
class LocalEnergyG2 : public SteerableFiltersG2
{
public:
    LocalEnergyG2(const Mat1f& image) : SteerableFiltersG2(image)
    {
        // the base constructor ran setup() while this object was still a plain SteerableFiltersG2: ask for the members
        syncMembers();
        m_rowsAtConstruction = m_g2a.rows;
    }
    float energy(int r, int c) const { return m_g2a(r, c) * m_g2a(r, c) + m_h2a(r, c) * m_h2a(r, c); }
    const Mat1f& member(int i) const
    {
        const Mat1f* all[12] = {&m_g2a, &m_g2b, &m_g2c, &m_h2a, &m_h2b, &m_h2c, &m_h2d, &m_c1, &m_c2, &m_c3, &m_theta, &m_orientationStrength};
        return *all[i];
    }
    bool unusedMembersEmpty() const { return m_dx.empty() && m_dy.empty(); }
    int tapsCols() const { return m_g1.cols; }
    int rowsAtConstruction() const { return m_rowsAtConstruction; }
    void keepMembers(bool on) { setMemberSync(on); }

private:
    int m_rowsAtConstruction;
};

class ProbeG4 : public SteerableFiltersG4
{
public:
    ProbeG4(const Mat1f& image) : SteerableFiltersG4(image) { syncMembers(); }
    const Mat1f& member(int i) const
    {
        const Mat1f* all[11] = {&m_g4a, &m_g4b, &m_g4c, &m_g4d, &m_g4e, &m_h4a, &m_h4b, &m_h4c, &m_h4d, &m_h4e, &m_h4f};
        return *all[i];
    }
    bool orientationMembersEmpty() const { return m_c1.empty() && m_c2.empty() && m_c3.empty() && m_theta.empty() && m_orientationStrength.empty(); }
};

_STEER_END

namespace {

fa::Mat1f make_image(int rows, int cols, float phase)
{
    fa::Mat1f m(rows, cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c)
            m(r, c) = 0.5f + 0.3f * std::sin(0.21f * c + 0.05f * r + phase) + 0.2f * std::cos(0.13f * (r - c)) + (c > cols / 2 ? 0.25f : 0.f);
    return m;
}

bool same(const fa::Mat1f& a, const fa::Mat1f& b)
{
    if (a.rows != b.rows || a.cols != b.cols || a.empty()) return false;
    for (int r = 0; r < a.rows; ++r)
        if (std::memcmp(a.ptr(r), b.ptr(r), (size_t)a.cols * sizeof(float)) != 0) return false;
    return true;
}

}  // namespace

int main()
{
    int failures = 0;
#define EXPECT(cond)                                                          \
    do {                                                                      \
        if (!(cond)) { std::printf("FAILED: %s (line %d)\n", #cond, __LINE__); ++failures; } \
    } while (0)
    try {
        const fa::Mat1f img1 = make_image(97, 131, 0.f), img2 = make_image(97, 131, 1.3f);
        fa::LocalEnergyG2 f(img1);
        EXPECT(f.rowsAtConstruction() == 97 && f.tapsCols() == 9 && f.unusedMembersEmpty());
        fa::SteerableFiltersG2 plain(img1);  // same image through the unextended class
        fa::Mat1f want, c1, c2, c3;
        for (int i = 0; i < 7; ++i) {
            plain.getBasis(i, want);
            EXPECT(same(f.member(i), want));
        }
        plain.getCoefficients(c1, c2, c3);
        EXPECT(same(f.member(7), c1) && same(f.member(8), c2) && same(f.member(9), c3));
        EXPECT(same(f.member(10), plain.getDominantOrientationAngle()) && same(f.member(11), plain.getDominantOrientationStrength()));
        plain.getBasis(0, want);
        fa::Mat1f h2a;
        plain.getBasis(3, h2a);
        EXPECT(f.energy(40, 60) == want(40, 60) * want(40, 60) + h2a(40, 60) * h2a(40, 60));
        // a later setup() through the base interface refreshes the members by itself (the object IS a subclass now)
        fa::SteerableFilters* base = &f;
        base->setup(img2);
        fa::SteerableFiltersG2 plain2(img2);
        for (int i = 0; i < 7; ++i) {
            plain2.getBasis(i, want);
            EXPECT(same(f.member(i), want));
        }
        EXPECT(same(f.member(10), plain2.getDominantOrientationAngle()));
        // ... unless the subclass says it never reads them
        f.keepMembers(false);
        base->setup(img1);
        plain2.getBasis(2, want);
        EXPECT(same(f.member(2), want));   // still the planes of img2
        fa::Mat1f g, h;
        f.steer(0.3f, g, h);               // the engine itself is on img1
        fa::Mat1f g1, h1;
        plain.steer(0.3f, g1, h1);
        EXPECT(same(g, g1) && same(h, h1));
        // G4: the eleven basis members; the orientation members stay empty as in the reference (G4.h:55, never assigned)
        fa::ProbeG4 f4(img1);
        fa::SteerableFiltersG4 plain4(img1);
        for (int i = 0; i < 11; ++i) {
            plain4.getBasis(i, want);
            EXPECT(same(f4.member(i), want));
        }
        EXPECT(f4.orientationMembersEmpty());
    } catch (const std::exception& ex) {
        std::printf("EXCEPTION: %s\n", ex.what());
        return 1;
    }
    std::printf(failures ? "cvsteer.subclass FAILED (%d)\n" : "cvsteer.subclass OK\n", failures);
    return failures ? 1 : 0;
}
