// fuzz_readers.cpp -- the image readers of the batch driver (cvsteer_amd/facade/cvsteer_run.cpp: binary PGM and .npy, files
// from anywhere) under AddressSanitizer + UndefinedBehaviorSanitizer: N mutated files per format -- truncated headers and
// rasters, maxval > 255, negative / zero / overflowing sizes, 2 GiB claims on a few bytes, stray comments, flipped bytes.
// Every case must end in an Image whose buffers match its header, or in a std::exception.  The reference's CI runs its test under
// ASan and LSan (.travis.yml:48-51); this is the host-side counterpart for the code that parses untrusted input.
//   g++ -std=c++11 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -Iinclude tests/cpp/fuzz_readers.cpp -o fuzz_readers
//   ./fuzz_readers 10000
#define CVSTEER_RUN_NO_MAIN
#include "../../cvsteer_amd/facade/cvsteer_run.cpp"

#include <cstdint>

namespace {
uint64_t rng_state = 0x9e3779b97f4a7c15ull;
uint32_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}
const char* kNumbers[] = {"0", "-1", "1", "7", "255", "256", "65535", "2147483647", "2147483648", "4294967295", "4294967296", "99999999999999999999",
                          "1048576", "1048577", "46341", "65536", "", " ", "1e9", "0x10", "+5", "3.5"};
std::string pick_number() { return kNumbers[rnd() % (sizeof kNumbers / sizeof *kNumbers)]; }

std::string valid_pgm(int rows, int cols, bool comment)
{
    std::string d = "P5\n";
    if (comment) d += "# a comment\n";
    d += std::to_string(cols) + " " + std::to_string(rows) + "\n255\n";
    for (int i = 0; i < rows * cols; ++i) d += (char)(rnd() & 0xff);
    return d;
}
std::string valid_npy(int rows, int cols, bool f4, int major)
{
    std::string dict = std::string("{'descr': '") + (f4 ? "<f4" : "|u1") + "', 'fortran_order': False, 'shape': (" + std::to_string(rows) + ", " + std::to_string(cols) + "), }";
    const size_t pre = major == 1 ? 10 : 12;
    while ((pre + dict.size() + 1) % 64) dict += ' ';
    dict += '\n';
    std::string d = "\x93NUMPY";
    d += (char)major;
    d += (char)0;
    d += (char)(dict.size() & 0xff);
    d += (char)((dict.size() >> 8) & 0xff);
    if (major != 1) { d += (char)0; d += (char)0; }
    d += dict;
    for (int i = 0; i < rows * cols * (f4 ? 4 : 1); ++i) d += (char)(rnd() & 0xff);
    return d;
}
void replace_first(std::string& d, const std::string& what, const std::string& with)
{
    const size_t p = d.find(what);
    if (p != std::string::npos) d.replace(p, what.size(), with);
}
std::string mutate(std::string d, int rows, int cols)
{
    switch (rnd() % 9) {
        case 0: d.resize(rnd() % (d.size() + 1)); break;                                   // truncated anywhere
        case 1: replace_first(d, std::to_string(cols), pick_number()); break;               // a size field gone wrong
        case 2: replace_first(d, std::to_string(rows), pick_number()); break;
        case 3: replace_first(d, "255", pick_number()); break;                             // maxval
        case 4: for (int k = 0; k < 1 + (int)(rnd() % 4); ++k) if (!d.empty()) d[rnd() % std::min<size_t>(d.size(), 96)] = (char)(rnd() & 0xff); break;   // flipped header bytes
        case 5: replace_first(d, std::to_string(rows), "46341"); replace_first(d, std::to_string(cols), "46341"); break;   // 2 GiB claimed, a few bytes there
        case 6: d.insert(rnd() % std::min<size_t>(d.size(), 20) , "#\n"); break;
        case 7: if (d.size() > 9) { d[8] = (char)0xff; d[9] = (char)0xff; } break;         // .npy header length beyond the file
        case 8: break;                                                                      // untouched: must parse
    }
    return d;
}
}  // namespace

int main(int argc, char** argv)
{
    const long n = argc > 1 ? std::atol(argv[1]) : 10000;
    long ok = 0, rejected = 0;
    for (long i = 0; i < n; ++i) {
        const int rows = 1 + (int)(rnd() % 9), cols = 1 + (int)(rnd() % 9);
        const int kind = (int)(rnd() % 3);
        const std::string good = kind == 0 ? valid_pgm(rows, cols, rnd() & 1) : valid_npy(rows, cols, kind == 2, (rnd() & 1) ? 1 : 2);
        const std::string d = mutate(good, rows, cols);
        try {
            const Image im = kind == 0 ? parse_pgm(d, "fuzz.pgm") : parse_npy(d, "fuzz.npy");
            const size_t px = (size_t)im.rows * (size_t)im.cols;
            if (im.rows <= 0 || im.cols <= 0 || (im.u8 ? im.bytes.size() : im.pix.size()) != px) {
                std::fprintf(stderr, "case %ld: accepted an image whose buffer does not match its header\n", i);
                return 1;
            }
            ++ok;
        } catch (const std::exception&) {
            ++rejected;
        }
    }
    std::printf("fuzz_readers: %ld cases, %ld parsed, %ld rejected, no sanitizer report\n", n, ok, rejected);
    return ok > 0 && rejected > 0 ? 0 : 1;
}
