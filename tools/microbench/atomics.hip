// Probe for the GroupAggregate sink: 12 64-bit atomics per row into acc[slot * 12 + s] with random
// slots (1e5 groups).  Variant A: one table, agent-scope atomics (what atomicAdd does).  Variant B:
// one table per XCD (index from HW_REG_XCC_ID), workgroup-scope atomics that complete in the XCD's
// own L2.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned long long u64;
typedef unsigned int u32;

__device__ __forceinline__ u32 xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF; }

template <int MODE>
__global__ __launch_bounds__(256) void k(const u32* __restrict__ slots, const double* __restrict__ v, long n, u64* table, long table_stride, u32* xcc_hist) {
  u64* T = table;
  if (MODE == 1) { const u32 x = xcc_id(); T = table + (long)x * table_stride; if (threadIdx.x == 0 && blockIdx.x < 4096) atomicAdd(&xcc_hist[x], 1u); }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const u32 s = slots[i];
    const double x = v[i];
    u64* A = T + (long)s * 12;
    if (MODE == 2) {   // SoA: accumulator j of all groups contiguous
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsafeAtomicAdd(reinterpret_cast<double*>(T + (long)(3 * j) * 100000 + s), x);
        atomicMin(T + (long)(3 * j + 1) * 100000 + s, (u64)__double_as_longlong(x));
        atomicMax(T + (long)(3 * j + 2) * 100000 + s, (u64)__double_as_longlong(x));
      }
      continue;
    }
    if (MODE == 3) {   // one atomic per row only
      unsafeAtomicAdd(reinterpret_cast<double*>(A), x);
      continue;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE == 0) {
        unsafeAtomicAdd(reinterpret_cast<double*>(A + 3 * j), x);
        atomicMin(A + 3 * j + 1, (u64)__double_as_longlong(x));
        atomicMax(A + 3 * j + 2, (u64)__double_as_longlong(x));
      } else {
        __hip_atomic_fetch_add(reinterpret_cast<double*>(A + 3 * j), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_min(A + 3 * j + 1, (u64)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_max(A + 3 * j + 2, (u64)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
}
__global__ void init(u32* slots, double* v, long n, u32 groups) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    u64 h = (u64)i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    slots[i] = (u32)(h % groups); v[i] = (double)(h & 1023);
  }
}
int main() {
  const long n = 100000000; const u32 groups = 100000; const long stride = (long)groups * 12;
  u32* slots; double* v; u64* table; u32* xh;
  hipMalloc(&slots, n * 4); hipMalloc(&v, n * 8); hipMalloc(&table, stride * 8 * 8); hipMalloc(&xh, 64);
  hipMemset(table, 0, stride * 8 * 8); hipMemset(xh, 0, 64);
  hipLaunchKernelGGL(init, dim3(4096), dim3(256), 0, 0, slots, v, n, groups);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 4; ++mode) for (int grid : {2048}) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, slots, v, n, table, stride, xh);
      else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, slots, v, n, table, stride, xh);
      else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, slots, v, n, table, stride, xh);
      else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, slots, v, n, table, stride, xh);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("mode=%d grid=%d: %.3f ms  %.1f G atomics/s  %.2f G rows/s\n", mode, grid, ms, n * 12.0 / ms / 1e6, n / ms / 1e6);
  }
  u32 h[16]; hipMemcpy(h, xh, 64, hipMemcpyDeviceToHost);
  printf("xcc histogram:"); for (int i = 0; i < 16; ++i) printf(" %u", h[i]); printf("\n");
  return 0;
}
