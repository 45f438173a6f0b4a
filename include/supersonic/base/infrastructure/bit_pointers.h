// supersonic/base/infrastructure/bit_pointers.h -- the reference's include path for this header (/root/reference/supersonic/base/infrastructure/bit_pointers.h).  The MI355X-native mirror
// keeps the whole builder API of the path in one header (of this one: bool_ptr and BoolView, the skip vectors of DoEvaluate); this file only makes the reference's #include line resolve.
#ifndef SSGPU_FWD_SUPERSONIC_BASE_INFRASTRUCTURE_BIT_POINTERS_H_
#define SSGPU_FWD_SUPERSONIC_BASE_INFRASTRUCTURE_BIT_POINTERS_H_
#include "../../../supersonic_amd/supersonic.h"
#endif  // SSGPU_FWD_SUPERSONIC_BASE_INFRASTRUCTURE_BIT_POINTERS_H_
