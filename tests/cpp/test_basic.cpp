// tests/cpp/test_basic.cpp -- the reference's only test (test/test.cpp:70-108), written against
// the drop-in facade exactly as a user of cvsteer would write it:
//     fa::SteerableFiltersG2 filters2(fish, 4, 0.67f);
//     filters2.steer(filters2.getDominantOrientationAngle(), g2, h2, e, magnitude, phase);
//     filters2.findEdges(magnitude, phase, edges); ...
// OpenCV's imdecode / normalize / imencode are not available in this image, so the fish comes
// from the pre-decoded fixture (tests/golden/fish_u8.npy) and the three result planes are
// written as raw f32 for the pytest wrapper (tests/test_gpu_facade.py), which does the
// normalise -> JPEG recode -> mean-L1 <= 1.0 comparison against the reference's golden JPEGs.
// Usage: test_basic <fish_u8.npy> <outdir>
#include <cvsteer/SteerableFiltersG2.h>
#include <cvsteer/SteerableFiltersG4.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

static std::vector<unsigned char> read_npy_u8(const char* path, int& rows, int& cols)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    unsigned char hdr[10];
    if (std::fread(hdr, 1, 10, f) != 10 || std::memcmp(hdr, "\x93NUMPY", 6) != 0) throw std::runtime_error("not an npy file");
    const int hlen = hdr[8] | (hdr[9] << 8);
    std::string h(hlen, '\0');
    if (std::fread(&h[0], 1, hlen, f) != (size_t)hlen) throw std::runtime_error("short npy header");
    const size_t s = h.find("'shape': (");
    if (s == std::string::npos || h.find("|u1") == std::string::npos) throw std::runtime_error("expected a 2-D |u1 array");
    if (std::sscanf(h.c_str() + s + 10, "%d, %d", &rows, &cols) != 2) throw std::runtime_error("bad shape");
    std::vector<unsigned char> d((size_t)rows * cols);
    if (std::fread(d.data(), 1, d.size(), f) != d.size()) throw std::runtime_error("short npy data");
    std::fclose(f);
    return d;
}

static void write_raw(const std::string& path, const fa::Mat1f& m)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    for (int r = 0; r < m.rows; ++r) std::fwrite(&m(r, 0), sizeof(float), m.cols, f);
    std::fclose(f);
}

#define EXPECT(cond)                                                          \
    do {                                                                      \
        if (!(cond)) { std::printf("FAILED: %s (line %d)\n", #cond, __LINE__); ++failures; } \
    } while (0)

int main(int argc, char** argv)
{
    if (argc < 3) { std::printf("usage: %s fish_u8.npy outdir\n", argv[0]); return 2; }
    int failures = 0;
    try {
        int rows = 0, cols = 0;
        std::vector<unsigned char> u8 = read_npy_u8(argv[1], rows, cols);
        // test.cpp:85: cv::Mat1f(fish) -- u8 -> f32, unscaled 0..255
        fa::Mat1f fish(rows, cols);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) fish(r, c) = (float)u8[(size_t)r * cols + c];

        // ---- the reference test body, verbatim in structure (test.cpp:84-90) ----
        fa::Mat1f g2, h2, e, magnitude, phase, edges, linesDark, linesBright;
        fa::SteerableFiltersG2 filters2(fish, 4, 0.67f);
        filters2.steer(filters2.getDominantOrientationAngle(), g2, h2, e, magnitude, phase);

        filters2.findEdges(magnitude, phase, edges);
        filters2.findDarkLines(magnitude, phase, linesDark);
        filters2.findBrightLines(magnitude, phase, linesBright);

        const std::string out(argv[2]);
        write_raw(out + "/edges.f32", edges);
        write_raw(out + "/linesDark.f32", linesDark);
        write_raw(out + "/linesBright.f32", linesBright);
        write_raw(out + "/theta.f32", filters2.getDominantOrientationAngle());
        write_raw(out + "/magnitude.f32", magnitude);
        write_raw(out + "/phase.f32", phase);
        EXPECT(edges.rows == rows && edges.cols == cols);

        // ---- rest of the surface: overloads agree with each other ----
        // theta passed as a *copy* (upload path) == theta passed as the getter's own Mat (device path)
        fa::Mat1f thetaCopy = filters2.getDominantOrientationAngle().clone();
        fa::Mat1f g2b, h2b;
        filters2.steer(thetaCopy, g2b, h2b);
        double d = 0;
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) d = std::fmax(d, std::fabs(g2b(r, c) - g2(r, c)) + std::fabs(h2b(r, c) - h2(r, c)));
        EXPECT(d == 0.0);

        // scalar steer: image overload vs single-point overload (G2.cpp:137-145 vs :115-123)
        fa::Mat1f gs, hs, es, ms, ps;
        filters2.steer(0.3f, gs, hs, es, ms, ps);
        const fa::Point pts[3] = {fa::Point(0, 0), fa::Point(cols - 1, rows - 1), fa::Point(100, 50)};
        for (const fa::Point& p : pts) {
            float pg, ph, pe, pm, pp;
            filters2.steer(p, 0.3f, pg, ph, pe, pm, pp);
            const float tol = 1e-5f * 400.f;  // planes on the 0..255 fish reach ~330
            EXPECT(std::fabs(pg - gs(p)) <= tol && std::fabs(ph - hs(p)) <= tol && std::fabs(pe - es(p)) <= 1e-5f * 2e5f);
            EXPECT(std::fabs(pm - ms(p)) <= tol);
            float pg2, ph2;
            filters2.steer(p, 0.3f, pg2, ph2);
            EXPECT(pg2 == pg && ph2 == ph);
        }

        // computeMagnitudeAndPhase + static phaseWeights reproduce findEdges (G2.cpp:194-204)
        fa::Mat1f mag2, ph2, lambda;
        filters2.computeMagnitudeAndPhase(g2, h2, mag2, ph2);
        fa::SteerableFiltersG2::phaseWeights(ph2, lambda, (float)M_PI_2, false, 2.0f);
        d = 0;
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) d = std::fmax(d, std::fabs(mag2(r, c) * lambda(r, c) - edges(r, c)));
        EXPECT(d <= 1e-3);

        // fused pipeline == the stepwise sequence
        fa::Mat1f q[8];
        fa::SteerableFiltersG2 filtersP(fa::Mat1f(), 4, 0.67f, 0);
        filtersP.pipeline(fish, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]);
        d = 0;
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c)
                d = std::fmax(d, std::fabs(q[5](r, c) - edges(r, c)) + std::fabs(q[6](r, c) - linesDark(r, c)) + std::fabs(q[7](r, c) - linesBright(r, c)));
        EXPECT(d == 0.0);

        // G4: constructed, steered both ways; getters are empty; computeMagnitudeAndPhase is a no-op
        fa::SteerableFiltersG4 filters4(fish);
        fa::Mat1f g4, h4, g4m, h4m, untouched;
        filters4.steer(0.3f, g4, h4);
        fa::Mat1f th(rows, cols);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) th(r, c) = 0.3f;
        filters4.steer(th, g4m, h4m);
        d = 0;
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) d = std::fmax(d, std::fabs(g4(r, c) - g4m(r, c)) + std::fabs(h4(r, c) - h4m(r, c)));
        EXPECT(d <= 1e-5 * 2000.0);
        EXPECT(filters4.getDominantOrientationAngle().empty());
        filters4.computeMagnitudeAndPhase(g4, h4, untouched, untouched);
        EXPECT(untouched.empty());
        write_raw(out + "/g4.f32", g4);
        write_raw(out + "/h4.f32", h4);

        // error behaviour: the reference throws from inside OpenCV on bad input
        bool threw = false;
        try { fa::Mat1f bad(3, 3), o1, o2; filters2.steer(bad, o1, o2); } catch (const std::exception&) { threw = true; }
        EXPECT(threw);
    } catch (const std::exception& ex) {
        std::printf("EXCEPTION: %s\n", ex.what());
        return 1;
    }
    std::printf(failures ? "cvsteer.basic FAILED (%d)\n" : "cvsteer.basic OK\n", failures);
    return failures ? 1 : 0;
}
