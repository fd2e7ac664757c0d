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
    Mat1f m_g1, m_g2, m_g3, m_g4, m_g5;
    Mat1f m_h1, m_h2, m_h3, m_h4, m_h5, m_h6;
    Mat1f m_theta, m_orientationStrength;  // never assigned, as in the reference

private:
    void init(const Mat1f& image);
};

}  // namespace fa

#endif
