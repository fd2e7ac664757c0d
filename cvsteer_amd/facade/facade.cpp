// facade.cpp -- fa::SteerableFilters / SteerableFiltersG2 / SteerableFiltersG4 on top of the C ABI
// (include/cvsteer_hip.h).  Builds libcvsteer.so: the library name the reference installs
// (cvsteer/CMakeLists.txt: target `cvsteer`), so an existing `-lcvsteer` link line picks up the
// MI355X engine.  No arithmetic on image data happens here: every method marshals fa::Mat1f
// views into cvs_plane structs and calls the HIP library.  Errors become std::runtime_error
// (the reference surfaces errors as cv::Exception thrown from inside OpenCV).
#include <cvsteer/SteerableFiltersG2.h>
#include <cvsteer/SteerableFiltersG4.h>

#include <mutex>
#include <stdexcept>
#include <string>
#include <typeinfo>

#include "cvsteer_hip.h"

namespace fa {

namespace {

cvs_plane view(const Mat1f& m)
{
    cvs_plane p;
    p.data = const_cast<float*>(reinterpret_cast<const float*>(m.data));
    p.rows = m.rows;
    p.cols = m.cols;
    p.step = (size_t)m.step;
    if (m.rows == 1 || p.step == 0) p.step = (size_t)m.cols * sizeof(float);
    p.mem = CVS_MEM_HOST;
    return p;
}

// (re)allocate an output the way cv::Mat::create does: no-op when the size already matches
cvs_plane out_view(Mat1f& m, int rows, int cols)
{
    m.create(rows, cols);
    return view(m);
}

void throw_status(cvs_handle h, int status, const char* where)
{
    std::string msg = std::string("cvsteer: ") + where + ": " + cvs_status_string(status);
    if (h) {
        const char* d = cvs_last_error(h);
        if (d && *d) msg += std::string(" (") + d + ")";
    }
    throw std::runtime_error(msg);
}

// handle for the static, object-less entry points (phaseWeights, wrap)
std::mutex g_static_mutex;
cvs_handle static_handle()
{
    static cvs_handle h = 0;
    if (!h) {
        int rc = cvs_create(CVS_KIND_G2, 4, 0.67f, 0, &h);
        if (rc != CVS_OK) throw_status(0, rc, "cvs_create (static helper handle; no HIP device?)");
    }
    return h;
}

Mat1f taps_of(cvs_handle h, int idx, int width)
{
    Mat1f k(1, 2 * width + 1);
    cvs_taps(h, idx, reinterpret_cast<float*>(k.data));
    return k;
}

}  // namespace

// --------------------------------------------------------------------------- base
SteerableFilters::SteerableFilters(int kind, int width, float spacing, int device) : m_memberSync(true), m_handle(0), m_device(device)
{
    int rc = cvs_create(kind, width, spacing, device, &m_handle);
    if (rc != CVS_OK) throw_status(0, rc, "cvs_create (is a HIP device present? there is no CPU fallback)");
}

SteerableFilters::~SteerableFilters()
{
    if (m_handle) cvs_destroy(m_handle);
}

void SteerableFilters::check(int status, const char* where) const
{
    if (status != CVS_OK) throw_status(m_handle, status, where);
}

// Only a subclass can read the protected plane members, so only objects whose dynamic type is NOT the facade class
// itself get them filled after setup().  (Inside a base-class constructor typeid(*this) is the base class: a subclass
// constructor that wants them calls syncMembers() itself.)
bool SteerableFilters::memberSyncWanted(const void* exact_type_info) const
{
    return m_memberSync && typeid(*this) != *static_cast<const std::type_info*>(exact_type_info);
}

void SteerableFilters::synchronize() { check(cvs_sync(m_handle), "cvs_sync"); }

void SteerableFilters::setExactAtan(bool on) { check(cvs_set_option(m_handle, CVS_OPT_ATAN_MODE, on ? 1 : 0), "cvs_set_option"); }

void SteerableFilters::fetch(int which, Mat1f& dst) const
{
    int rows = 0, cols = 0;
    check(cvs_shape(m_handle, &rows, &cols), "cvs_shape");
    if (rows == 0 || cols == 0) throw_status(m_handle, CVS_E_STATE, "fetch");
    cvs_plane p = out_view(dst, rows, cols);
    check(cvs_read_state(m_handle, which, &p), "cvs_read_state");
}

Mat1f SteerableFilters::create(int width, float spacing, KernelType f)
{
    Mat1f kernel(1, width * 2 + 1);
    float* k = reinterpret_cast<float*>(kernel.data);
    for (int i = -width; i <= width; i++) k[i + width] = f(float(i) * spacing);
    return kernel;
}

void SteerableFilters::wrap(const Mat1f& angle, Mat1f& output)
{
    std::lock_guard<std::mutex> lock(g_static_mutex);
    cvs_handle h = static_handle();
    Mat1f src = (angle.data == output.data) ? angle.clone() : angle;  // the reference is called aliased
    cvs_plane a = view(src), o = out_view(output, src.rows, src.cols);
    int rc = cvs_wrap(h, &a, &o);
    if (rc != CVS_OK) throw_status(h, rc, "cvs_wrap");
}

// --------------------------------------------------------------------------- G2
SteerableFiltersG2::SteerableFiltersG2(const Mat1f& image, int width, float spacing)
    : SteerableFilters(CVS_KIND_G2, width, spacing, 0), m_thetaValid(false), m_strengthValid(false)
{
    init(image);
}

SteerableFiltersG2::SteerableFiltersG2(const Mat1f& image, int width, float spacing, int device)
    : SteerableFilters(CVS_KIND_G2, width, spacing, device), m_thetaValid(false), m_strengthValid(false)
{
    init(image);
}

void SteerableFiltersG2::init(const Mat1f& image)
{
    int kind = 0, width = 0;
    cvs_kind(m_handle, &kind, &width, 0);
    Mat1f* taps[7] = {&m_g1, &m_g2, &m_g3, &m_h1, &m_h2, &m_h3, &m_h4};
    for (int i = 0; i < 7; ++i) *taps[i] = taps_of(m_handle, i, width);
    if (!image.empty()) setup(image);
}

void SteerableFiltersG2::setup(const Mat1f& image)
{
    m_thetaValid = m_strengthValid = false;
    cvs_plane p = view(image);
    check(cvs_setup(m_handle, &p, CVS_SETUP_FULL), "cvs_setup");
    if (memberSyncWanted(&typeid(SteerableFiltersG2))) syncMembers();
}

// SteerableFiltersG2.h:64-66 of the reference: m_g2a..m_h2d, m_c1..m_c3, m_theta, m_orientationStrength
void SteerableFiltersG2::syncMembers()
{
    Mat1f* basis[7] = {&m_g2a, &m_g2b, &m_g2c, &m_h2a, &m_h2b, &m_h2c, &m_h2d};
    for (int i = 0; i < 7; ++i) fetch(CVS_PLANE_BASIS0 + i, *basis[i]);
    fetch(CVS_PLANE_C1, m_c1);
    fetch(CVS_PLANE_C2, m_c2);
    fetch(CVS_PLANE_C3, m_c3);
    (void)getDominantOrientationAngle();
    (void)getDominantOrientationStrength();
}

const Mat1f& SteerableFiltersG2::getDominantOrientationAngle() const
{
    if (!m_thetaValid) {
        fetch(CVS_PLANE_THETA, m_theta);
        m_thetaValid = true;
    }
    return m_theta;
}

const Mat1f& SteerableFiltersG2::getDominantOrientationStrength() const
{
    if (!m_strengthValid) {
        fetch(CVS_PLANE_STRENGTH, m_orientationStrength);
        m_strengthValid = true;
    }
    return m_orientationStrength;
}

void SteerableFiltersG2::getBasis(int index, Mat1f& dst) const { fetch(CVS_PLANE_BASIS0 + index, dst); }

void SteerableFiltersG2::getCoefficients(Mat1f& c1, Mat1f& c2, Mat1f& c3) const
{
    fetch(CVS_PLANE_C1, c1);
    fetch(CVS_PLANE_C2, c2);
    fetch(CVS_PLANE_C3, c3);
}

void SteerableFiltersG2::steer(const Point& p, float theta, float& g2, float& h2)
{
    float out[5];
    check(cvs_steer_point(m_handle, p.x, p.y, theta, out), "cvs_steer_point");
    g2 = out[0];
    h2 = out[1];
}

void SteerableFiltersG2::steer(const Point& p, float theta, float& g2, float& h2, float& e, float& magnitude, float& phase)
{
    float out[5];
    check(cvs_steer_point(m_handle, p.x, p.y, theta, out), "cvs_steer_point");
    g2 = out[0];
    h2 = out[1];
    e = out[2];
    magnitude = out[3];
    phase = out[4];
}

// the callers pass getDominantOrientationAngle() straight back in (test/test.cpp:86): then the
// device-resident theta plane is used and nothing is uploaded
bool SteerableFiltersG2::isOwnTheta(const Mat1f& theta) const { return m_thetaValid && theta.data == m_theta.data; }

void SteerableFiltersG2::steer(float theta, Mat1f& g2, Mat1f& h2)
{
    int rows = 0, cols = 0;
    check(cvs_shape(m_handle, &rows, &cols), "cvs_shape");
    cvs_plane g = out_view(g2, rows, cols), h = out_view(h2, rows, cols);
    check(cvs_steer_scalar(m_handle, theta, &g, &h, 0, 0, 0), "cvs_steer_scalar");
}

void SteerableFiltersG2::steer(const Mat1f& theta, Mat1f& g2, Mat1f& h2)
{
    cvs_plane t = view(theta);
    cvs_plane g = out_view(g2, theta.rows, theta.cols), h = out_view(h2, theta.rows, theta.cols);
    check(cvs_steer_map(m_handle, isOwnTheta(theta) ? 0 : &t, &g, &h, 0, 0, 0), "cvs_steer_map");
}

void SteerableFiltersG2::steer(float theta, Mat1f& g2, Mat1f& h2, Mat1f& e, Mat1f& magnitude, Mat1f& phase)
{
    int rows = 0, cols = 0;
    check(cvs_shape(m_handle, &rows, &cols), "cvs_shape");
    cvs_plane g = out_view(g2, rows, cols), h = out_view(h2, rows, cols), pe = out_view(e, rows, cols);
    cvs_plane pm = out_view(magnitude, rows, cols), pp = out_view(phase, rows, cols);
    check(cvs_steer_scalar(m_handle, theta, &g, &h, &pe, &pm, &pp), "cvs_steer_scalar");
}

void SteerableFiltersG2::steer(const Mat1f& theta, Mat1f& g2, Mat1f& h2, Mat1f& e, Mat1f& magnitude, Mat1f& phase)
{
    cvs_plane t = view(theta);
    const int rows = theta.rows, cols = theta.cols;
    cvs_plane g = out_view(g2, rows, cols), h = out_view(h2, rows, cols), pe = out_view(e, rows, cols);
    cvs_plane pm = out_view(magnitude, rows, cols), pp = out_view(phase, rows, cols);
    check(cvs_steer_map(m_handle, isOwnTheta(theta) ? 0 : &t, &g, &h, &pe, &pm, &pp), "cvs_steer_map");
}

void SteerableFiltersG2::computeMagnitudeAndPhase(const Mat1f& g2, const Mat1f& h2, Mat1f& magnitude, Mat1f& phase)
{
    cvs_plane g = view(g2), h = view(h2);
    cvs_plane pm = out_view(magnitude, g2.rows, g2.cols), pp = out_view(phase, g2.rows, g2.cols);
    check(cvs_mag_phase(m_handle, &g, &h, &pm, &pp), "cvs_mag_phase");
}

void SteerableFiltersG2::findEdges(const Mat1f& e, const Mat1f& phase, Mat1f& output, float)
{
    cvs_plane pe = view(e), pp = view(phase), po = out_view(output, e.rows, e.cols);
    check(cvs_find(m_handle, &pe, &pp, &po, 0, 0), "cvs_find");
}

void SteerableFiltersG2::findDarkLines(const Mat1f& e, const Mat1f& phase, Mat1f& output, float)
{
    cvs_plane pe = view(e), pp = view(phase), po = out_view(output, e.rows, e.cols);
    check(cvs_find(m_handle, &pe, &pp, 0, &po, 0), "cvs_find");
}

void SteerableFiltersG2::findBrightLines(const Mat1f& e, const Mat1f& phase, Mat1f& output, float)
{
    cvs_plane pe = view(e), pp = view(phase), po = out_view(output, e.rows, e.cols);
    check(cvs_find(m_handle, &pe, &pp, 0, 0, &po), "cvs_find");
}

void SteerableFiltersG2::phaseWeights(const Mat1f& phase, Mat1f& lambda, float phi, bool signum, float k)
{
    std::lock_guard<std::mutex> lock(g_static_mutex);
    cvs_handle h = static_handle();
    cvs_plane pp = view(phase), pl = out_view(lambda, phase.rows, phase.cols);
    int rc = cvs_phase_weights(h, &pp, &pl, phi, signum ? 1 : 0, k);
    if (rc != CVS_OK) throw_status(h, rc, "cvs_phase_weights");
}

void SteerableFiltersG2::pipeline(const Mat1f& image, Mat1f& g2, Mat1f& h2, Mat1f& e, Mat1f& magnitude, Mat1f& phase,
                                  Mat1f& edges, Mat1f& linesDark, Mat1f& linesBright)
{
    m_thetaValid = m_strengthValid = false;
    cvs_plane pi = view(image);
    Mat1f* outs[8] = {&g2, &h2, &e, &magnitude, &phase, &edges, &linesDark, &linesBright};
    cvs_plane planes[8];
    const cvs_plane* ptrs[8];
    for (int i = 0; i < 8; ++i) {
        planes[i] = out_view(*outs[i], image.rows, image.cols);
        ptrs[i] = &planes[i];
    }
    check(cvs_pipeline(m_handle, &pi, ptrs), "cvs_pipeline");
}

// --------------------------------------------------------------------------- G4
SteerableFiltersG4::SteerableFiltersG4(const Mat1f& image, int width, float spacing)
    : SteerableFilters(CVS_KIND_G4, width, spacing, 0)
{
    init(image);
}

SteerableFiltersG4::SteerableFiltersG4(const Mat1f& image, int width, float spacing, int device)
    : SteerableFilters(CVS_KIND_G4, width, spacing, device)
{
    init(image);
}

void SteerableFiltersG4::init(const Mat1f& image)
{
    int kind = 0, width = 0;
    cvs_kind(m_handle, &kind, &width, 0);
    Mat1f* taps[11] = {&m_g1, &m_g2, &m_g3, &m_g4, &m_g5, &m_h1, &m_h2, &m_h3, &m_h4, &m_h5, &m_h6};
    for (int i = 0; i < 11; ++i) *taps[i] = taps_of(m_handle, i, width);
    if (!image.empty()) setup(image);
}

void SteerableFiltersG4::setup(const Mat1f& image)
{
    cvs_plane p = view(image);
    check(cvs_setup(m_handle, &p, CVS_SETUP_BASIS), "cvs_setup");
    if (memberSyncWanted(&typeid(SteerableFiltersG4))) syncMembers();
}

// SteerableFiltersG4.h:53-54 of the reference: m_g4a..m_g4e, m_h4a..m_h4f
void SteerableFiltersG4::syncMembers()
{
    Mat1f* basis[11] = {&m_g4a, &m_g4b, &m_g4c, &m_g4d, &m_g4e, &m_h4a, &m_h4b, &m_h4c, &m_h4d, &m_h4e, &m_h4f};
    for (int i = 0; i < 11; ++i) fetch(CVS_PLANE_BASIS0 + i, *basis[i]);
}

void SteerableFiltersG4::steer(const Mat1f& theta, Mat1f& g4, Mat1f& h4)
{
    cvs_plane t = view(theta);
    cvs_plane g = out_view(g4, theta.rows, theta.cols), h = out_view(h4, theta.rows, theta.cols);
    check(cvs_steer_map(m_handle, &t, &g, &h, 0, 0, 0), "cvs_steer_map");
}

void SteerableFiltersG4::steer(float theta, Mat1f& g4, Mat1f& h4)
{
    int rows = 0, cols = 0;
    check(cvs_shape(m_handle, &rows, &cols), "cvs_shape");
    cvs_plane g = out_view(g4, rows, cols), h = out_view(h4, rows, cols);
    check(cvs_steer_scalar(m_handle, theta, &g, &h, 0, 0, 0), "cvs_steer_scalar");
}

// G4.cpp:88-90: empty body in the reference; outputs are left untouched
void SteerableFiltersG4::computeMagnitudeAndPhase(const Mat1f&, const Mat1f&, Mat1f&, Mat1f&) {}

void SteerableFiltersG4::getBasis(int index, Mat1f& dst) const { fetch(CVS_PLANE_BASIS0 + index, dst); }

}  // namespace fa
