// cvsteer/SteerableFilters.h -- abstract base of the facade.
// Mirrors reference cvsteer/SteerableFilters.h:41-50: pure-virtual setup(image) and
// steer(theta, g, h); protected statics create() and wrap() keep their names and meaning
// (SteerableFilters.cpp:33-42, :46-51) for code that subclasses the base.
#ifndef CVSTEER_AMD_STEERABLEFILTERS_H
#define CVSTEER_AMD_STEERABLEFILTERS_H

#include <cvsteer/Mat.h>
#include <cvsteer/cvsteer.h>

struct cvs_context;

namespace fa {

class SteerableFilters {
public:
    virtual ~SteerableFilters();
    virtual void setup(const Mat1f& image) = 0;
    virtual void steer(float theta, Mat1f& g2, Mat1f& h2) = 0;

    // -- additions of this build (not in the reference) --
    int device() const { return m_device; }
    void synchronize();          // wait for the handle's HIP stream
    void setExactAtan(bool on);  // default off: OpenCV-compatible fastAtan2 polynomial
    cvs_context* handle() const { return m_handle; }  // the C-ABI handle (include/cvsteer_hip.h)

protected:
    typedef float (*KernelType)(float x);
    // k[i + width] = f(float(i) * spacing), i in [-width, width]   (SteerableFilters.cpp:33-42)
    static Mat1f create(int width, float spacing, KernelType f);
    // output = angle > pi ? angle - 2*pi : angle                    (SteerableFilters.cpp:46-51)
    static void wrap(const Mat1f& angle, Mat1f& output);

    SteerableFilters(int kind, int width, float spacing, int device);
    void check(int status, const char* where) const;  // throws std::runtime_error
    // copy state plane `which` (CVS_PLANE_*) into a host Mat1f
    void fetch(int which, Mat1f& dst) const;

    cvs_context* m_handle;
    int m_device;

private:
    SteerableFilters(const SteerableFilters&);
    SteerableFilters& operator=(const SteerableFilters&);
};

}  // namespace fa

#endif
