// edm/edm.h -- error convention of the EDM library (reference: lib/edm.h:5, lib/edm.cpp:4-7).
#ifndef EDM_EDM_H_
#define EDM_EDM_H_

#include <iostream>

namespace EDM {

// prints "[EDM:<location>] <error>" to stderr and abort()s, like the reference
void edm_error(const char* error, const char* location);

}  // namespace EDM
#endif
