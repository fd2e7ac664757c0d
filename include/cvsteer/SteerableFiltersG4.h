// cvsteer/SteerableFiltersG4.h -- G4/H4 quadrature pair, facade over libcvsteer_hip.so.
// Public surface = reference cvsteer/SteerableFiltersG4.h:38-48 (defaults width = 6,
// spacing = 0.5).  Like the reference: setup() builds the 11 basis planes only, steer() has the
// scalar and the per-pixel overloads, computeMagnitudeAndPhase() is an empty body
// (G4.cpp:88-90) and the two getters return Mats that are never filled (G4.h:40-41,55).
#ifndef CVSTEER_AMD_STEERABLEFILTERSG4_H
#define CVSTEER_AMD_STEERABLEFILTERSG4_H

#include <cvsteer/SteerableFilters.h>

namespace fa {

class SteerableFiltersG4 : public SteerableFilters {
public:
    SteerableFiltersG4(const Mat1f& image, int width = 6, float spacing = 0.5f);  // G4.cpp:47-65
    SteerableFiltersG4(const Mat1f& image, int width, float spacing, int device);

    const Mat1f& getDominantOrientationAngle() const { return m_theta; }
    const Mat1f& getDominantOrientationStrength() const { return m_orientationStrength; }

    void setup(const Mat1f& image);  // G4.cpp:67-81

    void steer(const Mat1f& theta, Mat1f& g4, Mat1f& h4);  // G4.cpp:92-112
    void steer(float theta, Mat1f& g4, Mat1f& h4);         // G4.cpp:114-122
    void computeMagnitudeAndPhase(const Mat1f& g4, const Mat1f& h4, Mat1f& magnitude, Mat1f& phase);  // G4.cpp:88-90: no-op

    // addition: the reference's protected basis planes m_g4a..m_h4f (index 0..10)
    void getBasis(int index, Mat1f& dst) const;

protected:
    // the reference's protected members, same names (SteerableFiltersG4.h:50-56): 11 tap vectors; the planes m_g4a..m_h4f
    // are host copies of the GPU state, filled for subclasses (see SteerableFilters.h); m_c1..m_c3, m_theta and
    // m_orientationStrength are declared and never assigned, as in the reference
    Mat1f m_g1, m_g2, m_g3, m_g4, m_g5;
    Mat1f m_h1, m_h2, m_h3, m_h4, m_h5, m_h6;
    Mat1f m_g4a, m_g4b, m_g4c, m_g4d, m_g4e;
    Mat1f m_h4a, m_h4b, m_h4c, m_h4d, m_h4e, m_h4f;
    Mat1f m_c1, m_c2, m_c3, m_theta, m_orientationStrength;
    void syncMembers();  // download m_g4a..m_h4f now

private:
    void init(const Mat1f& image);
};

}  // namespace fa

#endif
