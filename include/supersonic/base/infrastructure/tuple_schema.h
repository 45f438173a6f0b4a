// supersonic/base/infrastructure/tuple_schema.h -- the reference's include path for this header (/root/reference/supersonic/base/infrastructure/tuple_schema.h).  The MI355X-native mirror
// keeps the whole builder API of the path in one header; this file only makes the reference's #include line resolve.
#ifndef SSGPU_FWD_SUPERSONIC_BASE_INFRASTRUCTURE_TUPLE_SCHEMA_H_
#define SSGPU_FWD_SUPERSONIC_BASE_INFRASTRUCTURE_TUPLE_SCHEMA_H_
#include "../../../supersonic_amd/supersonic.h"
#endif  // SSGPU_FWD_SUPERSONIC_BASE_INFRASTRUCTURE_TUPLE_SCHEMA_H_
