// supersonic/supersonic.h -- the reference's umbrella include path (/root/reference: supersonic/supersonic.h:20-74), so
// that code written against it -- `#include "supersonic/supersonic.h"`, `using supersonic::Operation;` ... -- compiles
// against the MI355X-native library with nothing but `-I<repo>/include -lssgpu`.  Everything lives in
// supersonic_amd/supersonic.h (the header-only mirror over the C ABI of ssgpu.h).
#ifndef SUPERSONIC_SUPERSONIC_H_
#define SUPERSONIC_SUPERSONIC_H_
#include "../supersonic_amd/supersonic.h"
#endif  // SUPERSONIC_SUPERSONIC_H_
