// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the GroupAggregate kernels use (the guide
// calibrates only 16 B/lane streaming reads: FETCH_SIZE reports half of them).  Each kernel moves exactly 1 GiB; run under
//   rocprofv3 --pmc FETCH_SIZE --output-format csv ...   and   rocprofv3 --pmc WRITE_SIZE ...
// and divide the known KiB by the reported counter: tools/pmc_calibrate.sh prints the factors.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
template <typename T> __global__ __launch_bounds__(256) void calib_read(const T* __restrict__ p, u64 n, unsigned* out) {
  unsigned acc = 0;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) { T v = p[i]; const unsigned* w = reinterpret_cast<const unsigned*>(&v); for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc ^= w[k]; }
  if (acc == 0x12345679u) out[0] = acc;
}
template <typename T> __global__ __launch_bounds__(256) void calib_write(T* __restrict__ p, u64 n) {
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) { T v; unsigned* w = reinterpret_cast<unsigned*>(&v); for (unsigned k = 0; k < sizeof(T) / 4; ++k) w[k] = (unsigned)i + k; p[i] = v; }
}
// 40-byte records read as five 8-byte words per lane (ssgpu_part_agg_kernel's record fetch) and written as consecutive 8-byte words
__global__ __launch_bounds__(256) void calib_read_rec40(const u64* __restrict__ p, u64 n_rec, unsigned* out) {
  u64 acc = 0;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n_rec; i += (u64)gridDim.x * 256) { const u64* r = p + i * 5; acc ^= r[0] ^ r[1] ^ r[2] ^ r[3] ^ r[4]; }
  if (acc == 0x12345679ull) out[0] = 1;
}
int main() {
  const u64 bytes = 1ull << 30;
  char* buf; unsigned* out; CHECK(hipMalloc(&buf, bytes + 64)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(buf, 1, bytes));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_read<unsigned>, dim3(4096), dim3(256), 0, 0, (const unsigned*)buf, bytes / 4, out);
    hipLaunchKernelGGL(calib_read<u64>, dim3(4096), dim3(256), 0, 0, (const u64*)buf, bytes / 8, out);
    hipLaunchKernelGGL(calib_read<uint4>, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(calib_read_rec40, dim3(4096), dim3(256), 0, 0, (const u64*)buf, bytes / 40, out);
    hipLaunchKernelGGL(calib_write<unsigned>, dim3(4096), dim3(256), 0, 0, (unsigned*)buf, bytes / 4);
    hipLaunchKernelGGL(calib_write<u64>, dim3(4096), dim3(256), 0, 0, (u64*)buf, bytes / 8);
    hipLaunchKernelGGL(calib_write<uint4>, dim3(4096), dim3(256), 0, 0, (uint4*)buf, bytes / 16);
  }
  CHECK(hipDeviceSynchronize());
  printf("done\n");
  return 0;
}
