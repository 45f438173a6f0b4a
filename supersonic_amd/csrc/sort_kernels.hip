// sort_kernels.hip -- device radix sort for the Sort operator (placeholder TU,
// filled in by the sort milestone).
#include <hip/hip_runtime.h>
