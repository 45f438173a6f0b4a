// supersonic/cursor/base/operation.h -- the reference's include path for this header (/root/reference/supersonic/cursor/base/operation.h).  The MI355X-native mirror
// keeps the whole builder API of the path in one header; this file only makes the reference's #include line resolve.
#ifndef SSGPU_FWD_SUPERSONIC_CURSOR_BASE_OPERATION_H_
#define SSGPU_FWD_SUPERSONIC_CURSOR_BASE_OPERATION_H_
#include "../../../supersonic_amd/supersonic.h"
#endif  // SSGPU_FWD_SUPERSONIC_CURSOR_BASE_OPERATION_H_
