// ref_taps_driver.cpp -- dumps the tap vectors produced by the REFERENCE's own tap functions.
//
// The reference as a whole cannot be built here (every .cpp includes OpenCV, which this image
// lacks).  Its file-static 1-D tap functions (cvsteer/SteerableFiltersG2.cpp:35-42,
// cvsteer/SteerableFiltersG4.cpp:34-45) need only <cmath>, so oracle/ref_taps.mk slices
// exactly those lines out of the sources where they lie under /root/reference into
// oracle/_ref/ (git-ignored, never committed) and compiles them with this driver.  The
// driver applies create()'s sampling rule (SteerableFilters.cpp:33-42) and prints JSON.
// Output is committed as tests/golden/taps_ref.json: data, not source.
#define _USE_MATH_DEFINES
#include <cmath>
#include <math.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

namespace ref {
#include "g2_tap_lines.inc"
#include "g4_tap_lines.inc"
}

typedef float (*fn)(float);
static void dump(const char* name, fn f, int width, float spacing, bool last)
{
    std::printf("  \"%s\": [", name);
    for (int i = -width; i <= width; i++) {
        float v = f(float(i) * spacing);
        uint32_t u; std::memcpy(&u, &v, 4);
        std::printf("\"%08x\"%s", u, i == width ? "" : ", ");
    }
    std::printf("]%s\n", last ? "" : ",");
}

int main()
{
    using namespace ref;
    std::printf("{\n \"g2\": {\"width\": 4, \"spacing_hex\": \"3f2b851f\", \"order\": [\"G21\",\"G22\",\"G23\",\"H21\",\"H22\",\"H23\",\"H24\"],\n");
    fn g2[] = {G21, G22, G23, H21, H22, H23, H24};
    const char* n2[] = {"G21","G22","G23","H21","H22","H23","H24"};
    for (int i = 0; i < 7; i++) dump(n2[i], g2[i], 4, 0.67f, i == 6);
    std::printf(" },\n \"g4\": {\"width\": 6, \"spacing_hex\": \"3f000000\", \"order\": [\"G41\",\"G42\",\"G43\",\"G44\",\"G45\",\"H41\",\"H42\",\"H43\",\"H44\",\"H45\",\"H46\"],\n");
    fn g4[] = {G41, G42, G43, G44, G45, H41, H42, H43, H44, H45, H46};
    const char* n4[] = {"G41","G42","G43","G44","G45","H41","H42","H43","H44","H45","H46"};
    for (int i = 0; i < 11; i++) dump(n4[i], g4[i], 6, 0.5f, i == 10);
    // a second, non-default (width, spacing) per kind so the formula (not a table) is pinned
    std::printf(" },\n \"g2_w6_s05\": {\"width\": 6, \"spacing_hex\": \"3f000000\",\n");
    for (int i = 0; i < 7; i++) dump(n2[i], g2[i], 6, 0.5f, i == 6);
    std::printf(" },\n \"g4_w8_s04\": {\"width\": 8, \"spacing_hex\": \"3ecccccd\",\n");
    for (int i = 0; i < 11; i++) dump(n4[i], g4[i], 8, 0.4f, i == 10);
    std::printf(" }\n}\n");
    return 0;
}
