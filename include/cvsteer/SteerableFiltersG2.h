// cvsteer/SteerableFiltersG2.h -- G2/H2 quadrature pair, facade over libcvsteer_hip.so.
// Public surface = reference cvsteer/SteerableFiltersG2.h:38-60, same names, same default
// arguments (width = 4, spacing = 0.67, k = 2.0), same overload set.  Each method states the
// reference lines it replaces; the arithmetic runs in HIP kernels on the MI355X.
#ifndef CVSTEER_AMD_STEERABLEFILTERSG2_H
#define CVSTEER_AMD_STEERABLEFILTERSG2_H

#include <cvsteer/SteerableFilters.h>

namespace fa {

class SteerableFiltersG2 : public SteerableFilters {
public:
    // G2.cpp:44-58: build the 7 tap vectors, then setup(image)
    SteerableFiltersG2(const Mat1f& image, int width = 4, float spacing = 0.67f);
    // addition: choose the HIP device; image may be empty (call setup later)
    SteerableFiltersG2(const Mat1f& image, int width, float spacing, int device);

    // G2.h:40-41.  References stay valid until the next setup(); filled on first use.
    const Mat1f& getDominantOrientationAngle() const;
    const Mat1f& getDominantOrientationStrength() const;

    void setup(const Mat1f& image);  // G2.cpp:60-100

    // Steer filters at single point (G2.cpp:115-134); p.x = column, p.y = row
    void steer(const Point& p, float theta, float& g2, float& h2);
    void steer(const Point& p, float theta, float& g2, float& h2, float& e, float& magnitude, float& phase);
    void steer(const Mat1f& theta, Mat1f& g2, Mat1f& h2);  // G2.cpp:147-155

    // Processing on entire images
    void steer(float theta, Mat1f& g2, Mat1f& h2);  // G2.cpp:137-145
    void steer(float theta, Mat1f& g2, Mat1f& h2, Mat1f& e, Mat1f& magnitude, Mat1f& phase);          // G2.cpp:157-165
    void steer(const Mat1f& theta, Mat1f& g2, Mat1f& h2, Mat1f& e, Mat1f& magnitude, Mat1f& phase);   // G2.cpp:167-177
    void computeMagnitudeAndPhase(const Mat1f& g2, const Mat1f& h2, Mat1f& magnitude, Mat1f& phase);  // G2.cpp:107-112

    void findEdges(const Mat1f& e, const Mat1f& phase, Mat1f& output, float k = 2.0f);        // G2.cpp:201-204
    void findDarkLines(const Mat1f& e, const Mat1f& phase, Mat1f& output, float k = 2.0f);    // G2.cpp:205-208
    void findBrightLines(const Mat1f& e, const Mat1f& phase, Mat1f& output, float k = 2.0f);  // G2.cpp:209-212

    static void phaseWeights(const Mat1f& phase, Mat1f& lambda, float phi, bool signum, float k);  // G2.cpp:179-186

    // -- additions of this build --
    // the reference's protected basis planes m_g2a..m_h2d (index 0..6) and m_c1..m_c3
    void getBasis(int index, Mat1f& dst) const;
    void getCoefficients(Mat1f& c1, Mat1f& c2, Mat1f& c3) const;
    // the callers' whole sequence (test/test.cpp:85-90) in two kernel launches
    void pipeline(const Mat1f& image, Mat1f& g2, Mat1f& h2, Mat1f& e, Mat1f& magnitude, Mat1f& phase,
                  Mat1f& edges, Mat1f& linesDark, Mat1f& linesBright);

protected:
    // the reference's protected members, same names (SteerableFiltersG2.h:62-66).  m_g1..m_h4 are the 7 tap vectors; the
    // planes m_g2a..m_h2d, m_c1..m_c3 are host copies of the GPU state, filled for subclasses (see SteerableFilters.h:
    // after setup() on a subclass object, or by syncMembers()); m_dx / m_dy are declared and never used, as in the reference
    Mat1f m_dx, m_dy;
    Mat1f m_g1, m_g2, m_g3, m_h1, m_h2, m_h3, m_h4;
    Mat1f m_g2a, m_g2b, m_g2c, m_h2a, m_h2b, m_h2c, m_h2d;
    Mat1f m_c1, m_c2, m_c3;
    mutable Mat1f m_theta, m_orientationStrength;    // host copies, fetched lazily by the getters
    mutable bool m_thetaValid, m_strengthValid;
    void syncMembers();  // download m_g2a..m_h2d, m_c1..m_c3, m_theta, m_orientationStrength now

private:
    void init(const Mat1f& image);
    bool isOwnTheta(const Mat1f& theta) const;
};

}  // namespace fa

#endif
