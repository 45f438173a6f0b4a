// supersonic/cursor/core/sort.h -- the reference's include path for this header (/root/reference/supersonic/cursor/core/sort.h).  The MI355X-native mirror
// keeps the whole builder API of the path in one header; this file only makes the reference's #include line resolve.
#ifndef SSGPU_FWD_SUPERSONIC_CURSOR_CORE_SORT_H_
#define SSGPU_FWD_SUPERSONIC_CURSOR_CORE_SORT_H_
#include "../../../supersonic_amd/supersonic.h"
#endif  // SSGPU_FWD_SUPERSONIC_CURSOR_CORE_SORT_H_
