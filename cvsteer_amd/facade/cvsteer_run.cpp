// cvsteer_run.cpp -- `cvsteer-run` for MI355X: the reference's batch driver (example/steer.cpp:59-173) in C++ over
// the C ABI (include/cvsteer_hip.h).  Host side only: no HIP call is made here.
//
//   cvsteer-run --input <image | list.txt> --output <dir> [--gain G] [--gpus N | --devices a,b,..] [--ext .pgm|.npy] [--verbose]
//
// Per image the reference's per-file body (steer.cpp:69-124): gray f32 (unscaled 0..255) -> SteerableFiltersG2(gray, 4,
// 0.67) -> steer at the dominant orientation -> findEdges / findDarkLines / findBrightLines on the magnitude -> 8-bit
// via normalize(0, 255, MINMAX) or convertTo(gain) -> <base>_edges, <base>_lines_dark, <base>_lines_bright.
// Where the reference runs cv::parallel_for_ over the files (steer.cpp:169), this driver hands runs of equally sized
// images to cvs_batch_run as HOST planes: every GPU uploads its own block of frames over its own link, filters it in
// one fused launch per chunk, turns the three feature maps into bytes on the device (8-bit host output planes of
// cvs_batch_run), and only those come back -- 1 byte per pixel and map instead of 4 -- while the next chunk goes up.
// OpenCV's imgcodecs are not available: inputs are binary PGM (P5, maxval <= 255) or .npy (2-D, |u1 or <f4, C order),
// outputs PGM or .npy.  Differences from the reference, on purpose: --gain is honoured (steer.cpp:167-168 passes
// `verbose` as the gain); single-channel inputs work (steer.cpp:79-82 leaves `gray` empty for them); unreadable files
// are reported and make the exit status non-zero.
#include <cvsteer_hip.h>

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Image {
    std::string path;
    int rows = 0, cols = 0;
    bool u8 = true;              // 8-bit sources stay bytes: they cross the host link as bytes and are widened on the GPU
    std::vector<uint8_t> bytes;  // gray, 0..255
    std::vector<float> pix;      // gray, unscaled (Mat1f(gray), steer.cpp:84) -- float sources only
};

std::string base_name(const std::string& path)
{
    const size_t slash = path.find_last_of('/');
    std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
    const size_t dot = name.find_last_of('.');
    return dot == std::string::npos ? name : name.substr(0, dot);
}

std::string lower_ext(const std::string& path)
{
    const size_t dot = path.find_last_of('.');
    std::string e = dot == std::string::npos ? "" : path.substr(dot);
    for (char& c : e) c = (char)std::tolower((unsigned char)c);
    return e;
}

std::string slurp(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(path + ": cannot open");
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// a decimal header field: digits only, at most 9 of them (no sign, no overflow); -1 = anything else
long header_number(const std::string& t)
{
    if (t.empty() || t.size() > 9) return -1;
    long v = 0;
    for (char c : t) {
        if (c < '0' || c > '9') return -1;
        v = v * 10 + (c - '0');
    }
    return v;
}

constexpr long kMaxDim = 1 << 20;   // rows / columns this driver accepts (the engine itself takes larger planes)

// binary PGM: "P5" ws width ws height ws maxval single-ws raster ('#' comments allowed in the header).  `d` = the file's bytes:
// untrusted input -- every size is checked against the bytes that are really there before anything is copied.
Image parse_pgm(const std::string& d, const std::string& path)
{
    size_t pos = 0;
    auto token = [&]() {
        for (;;) {
            while (pos < d.size() && std::isspace((unsigned char)d[pos])) ++pos;
            if (pos < d.size() && d[pos] == '#') {
                while (pos < d.size() && d[pos] != '\n') ++pos;
                continue;
            }
            break;
        }
        const size_t b = pos;
        while (pos < d.size() && !std::isspace((unsigned char)d[pos])) ++pos;
        return d.substr(b, pos - b);
    };
    if (token() != "P5") throw std::runtime_error(path + ": not a binary PGM (P5)");
    Image im;
    im.path = path;
    const long cols = header_number(token()), rows = header_number(token()), maxval = header_number(token());
    if (rows <= 0 || cols <= 0 || rows > kMaxDim || cols > kMaxDim || maxval <= 0 || maxval > 255) throw std::runtime_error(path + ": unsupported PGM header");
    if (pos >= d.size()) throw std::runtime_error(path + ": truncated raster");
    ++pos;  // the single whitespace byte after maxval
    const size_t n = (size_t)rows * (size_t)cols;
    if (d.size() - pos < n) throw std::runtime_error(path + ": truncated raster");
    im.rows = (int)rows;
    im.cols = (int)cols;
    im.bytes.assign(reinterpret_cast<const uint8_t*>(d.data()) + pos, reinterpret_cast<const uint8_t*>(d.data()) + pos + n);
    return im;
}

Image read_pgm(const std::string& path) { return parse_pgm(slurp(path), path); }

// NumPy .npy version 1.x / 2.x, 2-D, '|u1' or '<f4', fortran_order False (`d` = the file's bytes, untrusted)
Image parse_npy(const std::string& d, const std::string& path)
{
    if (d.size() < 12 || std::memcmp(d.data(), "\x93NUMPY", 6) != 0) throw std::runtime_error(path + ": not an .npy file");
    const int major = (unsigned char)d[6];
    size_t hlen, off;
    if (major == 1) { hlen = (unsigned char)d[8] | ((size_t)(unsigned char)d[9] << 8); off = 10; }
    else { hlen = (unsigned char)d[8] | ((size_t)(unsigned char)d[9] << 8) | ((size_t)(unsigned char)d[10] << 16) | ((size_t)(unsigned char)d[11] << 24); off = 12; }
    if (d.size() - off < hlen) throw std::runtime_error(path + ": truncated header");
    const std::string h = d.substr(off, hlen);
    const bool u1 = h.find("'|u1'") != std::string::npos, f4 = h.find("'<f4'") != std::string::npos;
    if (!u1 && !f4) throw std::runtime_error(path + ": dtype must be uint8 or little-endian float32");
    if (h.find("'fortran_order': False") == std::string::npos) throw std::runtime_error(path + ": fortran order is not supported");
    const size_t sp = h.find("'shape': (");
    if (sp == std::string::npos) throw std::runtime_error(path + ": no shape");
    // "(rows, cols)": two decimal fields, nothing else
    size_t q = sp + 10;
    auto field = [&]() {
        while (q < h.size() && h[q] == ' ') ++q;
        const size_t b = q;
        while (q < h.size() && h[q] >= '0' && h[q] <= '9') ++q;
        return header_number(h.substr(b, q - b));
    };
    const long r = field();
    if (q >= h.size() || h[q] != ',') throw std::runtime_error(path + ": expected a 2-D array");
    ++q;
    const long c = field();
    while (q < h.size() && h[q] == ' ') ++q;
    if (q >= h.size() || h[q] != ')' || r <= 0 || c <= 0 || r > kMaxDim || c > kMaxDim) throw std::runtime_error(path + ": expected a 2-D array");
    Image im;
    im.path = path;
    im.rows = (int)r;
    im.cols = (int)c;
    const size_t n = (size_t)r * (size_t)c, data = off + hlen, elem = u1 ? 1 : 4;
    if ((d.size() - data) / elem < n) throw std::runtime_error(path + ": truncated data");
    im.u8 = u1;
    if (u1) im.bytes.assign(reinterpret_cast<const uint8_t*>(d.data()) + data, reinterpret_cast<const uint8_t*>(d.data()) + data + n);
    else {
        im.pix.resize(n);
        std::memcpy(im.pix.data(), d.data() + data, n * 4);
    }
    return im;
}

Image read_npy(const std::string& path) { return parse_npy(slurp(path), path); }

Image read_gray(const std::string& path)
{
    const std::string e = lower_ext(path);
    if (e == ".npy") return read_npy(path);
    if (e == ".pgm") return read_pgm(path);
    throw std::runtime_error(path + ": unsupported format (binary PGM or .npy; no imgcodecs in this build)");
}

void write_u8(const std::string& path, const std::vector<uint8_t>& u8, int rows, int cols)
{
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(path + ": cannot create");
    if (lower_ext(path) == ".npy") {
        std::string dict = "{'descr': '|u1', 'fortran_order': False, 'shape': (" + std::to_string(rows) + ", " + std::to_string(cols) + "), }";
        while ((10 + dict.size() + 1) % 64) dict += ' ';
        dict += '\n';
        const char head[10] = {'\x93', 'N', 'U', 'M', 'P', 'Y', 1, 0, (char)(dict.size() & 0xff), (char)(dict.size() >> 8)};
        f.write(head, 10);
        f.write(dict.data(), (std::streamsize)dict.size());
    } else {
        const std::string head = "P5\n" + std::to_string(cols) + " " + std::to_string(rows) + "\n255\n";
        f.write(head.data(), (std::streamsize)head.size());
    }
    f.write(reinterpret_cast<const char*>(u8.data()), (std::streamsize)u8.size());
    if (!f) throw std::runtime_error(path + ": write failed");
}

// steer.cpp:156-165: a .txt file (or a name without extension) is a list of files, else one image
std::vector<std::string> input_list(const std::string& arg)
{
    const std::string name = arg.substr(arg.find_last_of('/') == std::string::npos ? 0 : arg.find_last_of('/') + 1);
    if (lower_ext(arg) != ".txt" && name.find('.') != std::string::npos) return {arg};
    std::ifstream f(arg);
    if (!f) throw std::runtime_error(arg + ": cannot open the file list");
    std::vector<std::string> out;
    std::string ln;
    while (std::getline(f, ln)) {
        while (!ln.empty() && std::isspace((unsigned char)ln.back())) ln.pop_back();
        size_t b = 0;
        while (b < ln.size() && std::isspace((unsigned char)ln[b])) ++b;
        if (b < ln.size()) out.push_back(ln.substr(b));
    }
    return out;
}

#ifndef CVSTEER_RUN_NO_MAIN   // (tests/cpp/fuzz_readers.cpp includes this file for the readers above)
// the status first, THEN the error text: written as `check(call(&batch), ..., cvs_batch_last_error(batch))` the two
// arguments are evaluated in unspecified order and the text may be read before the call has run (or created `batch`)
#define CHECK_BATCH(batch, call, what)                         \
    do {                                                       \
        const int rc__ = (call);                               \
        check(rc__, what, cvs_batch_last_error(batch));        \
    } while (0)

void check(int rc, const char* what, const char* detail)
{
    if (rc != CVS_OK) throw std::runtime_error(std::string(what) + ": " + cvs_status_string(rc) + (detail && *detail ? std::string(" -- ") + detail : ""));
}

#endif
}  // namespace

#ifndef CVSTEER_RUN_NO_MAIN
int main(int argc, char** argv)
{
    std::string input, output, ext = ".pgm";
    float gain = 0.f;
    int gpus = 1, max_batch = 64;
    std::vector<int> device_list;
    bool verbose = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); }
            return argv[++i];
        };
        if (a == "--input" || a == "-i") input = val();
        else if (a == "--output" || a == "-o") output = val();
        else if (a == "--gain" || a == "-g") gain = (float)std::atof(val().c_str());
        else if (a == "--gpus") gpus = std::atoi(val().c_str());
        else if (a == "--devices") {  // explicit list, e.g. 0,1,2,3; a device listed twice = rehearsal of a larger world on a smaller box
            device_list.clear();
            std::stringstream ss(val());
            std::string tok;
            while (std::getline(ss, tok, ',')) device_list.push_back(std::atoi(tok.c_str()));
            gpus = (int)device_list.size();
        }
        else if (a == "--ext") ext = val();
        else if (a == "--max-batch") max_batch = std::max(1, std::atoi(val().c_str()));
        else if (a == "--verbose" || a == "-v") verbose = true;
        else if (a == "--help" || a == "-h") {
            std::printf("usage: cvsteer-run --input <image | list.txt> --output <dir> [--gain G] [--gpus N | --devices a,b,..] [--ext .pgm|.npy] [--verbose]\n");
            return 0;
        } else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (input.empty() || output.empty() || gpus < 1) { std::fprintf(stderr, "cvsteer-run: --input and --output are required\n"); return 2; }
    int failures = 0;
    cvs_batch batch = nullptr;
    try {
        const std::vector<std::string> files = input_list(input);
        std::vector<int> devices(gpus);
        for (int d = 0; d < gpus; ++d) devices[d] = device_list.empty() ? d : device_list[d];
        CHECK_BATCH(batch, cvs_batch_create_local(CVS_KIND_G2, 4, 0.67f, gpus, devices.data(), &batch), "cvs_batch_create_local");
        CHECK_BATCH(batch, cvs_batch_set_option(batch, CVS_OPT_PERSIST_STATE, 0), "cvs_batch_set_option");  // only the three maps are kept

        size_t next = 0;
        while (next < files.size()) {
            // a run of consecutive images of one size = one batch (frames of a batch share their geometry)
            std::vector<Image> run;
            while (next < files.size() && (int)run.size() < max_batch) {
                Image im;
                try {
                    im = read_gray(files[next]);
                } catch (const std::exception& e) {
                    std::fprintf(stderr, "cvsteer-run: %s\n", e.what());
                    ++failures;
                    ++next;
                    continue;
                }
                if (!run.empty() && (im.rows != run[0].rows || im.cols != run[0].cols || im.u8 != run[0].u8)) break;  // starts the next run
                run.push_back(std::move(im));
                ++next;
            }
            if (run.empty()) continue;
            const int rows = run[0].rows, cols = run[0].cols, n = (int)run.size();
            std::vector<cvs_plane> in(n);
            for (int f = 0; f < n; ++f)
                in[f] = run[f].u8 ? cvs_plane{reinterpret_cast<float*>(run[f].bytes.data()), rows, cols, (size_t)cols, CVS_MEM_HOST | CVS_DEPTH_U8}
                                  : cvs_plane{run[f].pix.data(), rows, cols, (size_t)cols * sizeof(float), CVS_MEM_HOST};
            cvs_batch_cfg cfg;
            std::memset(&cfg, 0, sizeof cfg);
            cfg.rows = rows;
            cfg.cols = cols;
            cfg.n_frames = n;
            cfg.outputs = (1u << 5) | (1u << 6) | (1u << 7);  // edges, dark lines, bright lines
            cfg.root = 0;
            cfg.gather = 1;
            // 8-bit HOST output planes: every GPU turns its maps into bytes itself -- normalize(0, 255, MINMAX) per map
            // (steer.cpp:98-104) or convertTo(gain) (steer.cpp:92-97) -- and only bytes come back, chunk by chunk, while
            // the next chunk of frames is uploaded and filtered (cvs_batch_run on host planes; every rank over its own link)
            const size_t plane = (size_t)rows * cols;
            std::vector<uint8_t> u8((size_t)n * 3 * plane);
            std::vector<cvs_plane> outp((size_t)n * 8);
            std::memset(outp.data(), 0, outp.size() * sizeof(cvs_plane));
            for (int f = 0; f < n; ++f)
                for (int j = 0; j < 3; ++j)
                    outp[(size_t)f * 8 + 5 + j] = cvs_plane{reinterpret_cast<float*>(u8.data() + ((size_t)f * 3 + j) * plane), rows, cols, (size_t)cols,
                                                            CVS_MEM_HOST | CVS_DEPTH_U8};
            CHECK_BATCH(batch, cvs_batch_set_u8_gain(batch, gain > 0.f ? gain : 0.f), "cvs_batch_set_u8_gain");
            cvs_batch_timing t;
            CHECK_BATCH(batch, cvs_batch_run(batch, &cfg, in.data(), outp.data(), &t), "cvs_batch_run");
            if (verbose) std::printf("batch of %d x %dx%d: upload %.2f ms, download %.2f ms, span %.2f ms\n", n, rows, cols, t.scatter_ms, t.gather_ms, t.compute_ms);
            static const char* suffix[3] = {"_edges", "_lines_dark", "_lines_bright"};
            const int frame = n;
            std::vector<uint8_t> one(plane);
            for (int f = 0; f < frame; ++f)
                for (int j = 0; j < 3; ++j) {
                    const std::string dst = output + "/" + base_name(run[f].path) + suffix[j] + ext;
                    one.assign(u8.begin() + ((size_t)f * 3 + j) * plane, u8.begin() + ((size_t)f * 3 + j + 1) * plane);
                    write_u8(dst, one, rows, cols);
                    if (verbose) std::printf("%s\n", dst.c_str());
                }
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "cvsteer-run: %s\n", e.what());
        failures = failures ? failures : 1;
    }
    if (batch) cvs_batch_destroy(batch);
    return failures ? 1 : 0;
}
#endif
