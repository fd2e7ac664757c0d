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

    // The reference keeps every intermediate plane in protected cv::Mat1f members (m_g2a.., m_c1.., SteerableFiltersG2.h:62-66,
    // SteerableFiltersG4.h:50-56) that a subclass may read.  Here those planes live on the GPU; the members of the same names
    // exist and are host COPIES, filled
    //   * after every setup() that runs on an object whose dynamic type is a SUBCLASS of the facade classes (nobody but a
    //     subclass can read protected members, so plain G2 / G4 objects never pay for the download), and
    //   * on demand by syncMembers() -- for a subclass constructor that reads them right after the base constructor ran
    //     (inside the base constructor the object is not yet a subclass, as in any C++ class).
    // setMemberSync(false) switches the automatic copy off for subclasses that never read them (12 planes of 64 MiB per
    // 4096^2 image cross the host link otherwise).
    virtual void syncMembers() {}
    void setMemberSync(bool on) { m_memberSync = on; }
    bool memberSyncWanted(const void* exact_type_info) const;  // true when *this is a subclass and the copy is on
    bool m_memberSync;

    cvs_context* m_handle;
    int m_device;

private:
    SteerableFilters(const SteerableFilters&);
    SteerableFilters& operator=(const SteerableFilters&);
};

}  // namespace fa

#endif
