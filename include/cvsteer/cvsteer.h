// cvsteer/cvsteer.h -- namespace of the drop-in facade.
// Same namespace as the reference (cvsteer/cvsteer.h:12-15: "fa" = Freeman and Adelson), so
// `fa::SteerableFiltersG2` in existing sources resolves to the MI355X engine -- and the same two
// macros, so that downstream files written like the reference's own headers
// (`_STEER_BEGIN class MyFilters : public SteerableFiltersG2 { ... }; _STEER_END`) compile unchanged.
#ifndef CVSTEER_AMD_CVSTEER_H
#define CVSTEER_AMD_CVSTEER_H

#define CVSTEER_AMD 1
#define CVSTEER_NAMESPACE fa

// reference cvsteer/cvsteer.h:12-15
#define _STEER_BEGIN \
    namespace fa     \
    { // Freeman and Adelson
#define _STEER_END }

#endif
