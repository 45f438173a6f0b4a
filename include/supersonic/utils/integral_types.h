// supersonic/utils/integral_types.h -- the reference's include path for this header (/root/reference/supersonic/utils/integral_types.h).  The MI355X-native mirror
// keeps the whole builder API of the path in one header; this file only makes the reference's #include line resolve.
#ifndef SSGPU_FWD_SUPERSONIC_UTILS_INTEGRAL_TYPES_H_
#define SSGPU_FWD_SUPERSONIC_UTILS_INTEGRAL_TYPES_H_
#include "../../supersonic_amd/supersonic.h"
#endif  // SSGPU_FWD_SUPERSONIC_UTILS_INTEGRAL_TYPES_H_
