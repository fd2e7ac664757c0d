// cvsteer/cvsteer.h -- namespace of the drop-in facade.
// Same namespace as the reference (cvsteer/cvsteer.h:12-15: "fa" = Freeman and Adelson), so
// `fa::SteerableFiltersG2` in existing sources resolves to the MI355X engine.
#ifndef CVSTEER_AMD_CVSTEER_H
#define CVSTEER_AMD_CVSTEER_H

#define CVSTEER_AMD 1
#define CVSTEER_NAMESPACE fa

#endif
