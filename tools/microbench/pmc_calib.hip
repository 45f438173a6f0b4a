// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the GroupAggregate kernels use (the guide
// calibrates only 16 B/lane streaming reads: FETCH_SIZE reports half of them).  Each kernel moves exactly 1 GiB; run under
//   rocprofv3 --pmc FETCH_SIZE --output-format csv ...   and   rocprofv3 --pmc WRITE_SIZE ...
// and divide the known KiB by the reported counter: tools/pmc_calibrate.sh prints the factors.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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
// Random 64-byte record reads (ssgpu_sort_gather_rec_kernel's access pattern: four lanes fetch one record as 4 x 16 bytes; the
// record index is a bijection of the row, so every record of the window is read exactly once: known bytes = records x 64).
// `mask` + 1 = records in the window (a power of two): 64 M records = 4 GiB (far beyond L2 and the Infinity Cache),
// 64 K records = 4 MiB (fits one XCD's L2).
__global__ __launch_bounds__(256) void calib_gather_rec64(const uint4* __restrict__ recs, u64 n_rec, u64 mask, unsigned* out) {
  unsigned acc = 0;
  for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < n_rec * 4; j += (u64)gridDim.x * 256) {
    const u64 row = j >> 2, ch = j & 3;
    const u64 rec = (row * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & mask;   // odd multiplier: a bijection modulo 2^k
    const uint4 v = recs[rec * 4 + ch];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345679u) out[0] = acc;
}
// the same records, one lane per record (4 x 16 bytes from one lane: 64 lanes of a wave touch 64 different records per load)
__global__ __launch_bounds__(256) void calib_gather_rec64_lane(const uint4* __restrict__ recs, u64 n_rec, u64 mask, unsigned* out) {
  unsigned acc = 0;
  for (u64 row = (u64)blockIdx.x * 256 + threadIdx.x; row < n_rec; row += (u64)gridDim.x * 256) {
    const u64 rec = (row * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & mask;
    const uint4 a = recs[rec * 4], b = recs[rec * 4 + 1], c = recs[rec * 4 + 2], d = recs[rec * 4 + 3];
    acc ^= a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345679u) out[0] = acc;
}
// random 8-byte reads (the column-by-column gather): known bytes = n x 8, every element once
__global__ __launch_bounds__(256) void calib_gather_u64(const u64* __restrict__ p, u64 n, u64 mask, unsigned* out) {
  u64 acc = 0;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) acc ^= p[(i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & mask];
  if (acc == 0x12345679ull) out[0] = 1;
}
// random 128-byte (line-sized, line-aligned) reads: eight lanes fetch one line
__global__ __launch_bounds__(256) void calib_gather_rec128(const uint4* __restrict__ recs, u64 n_rec, u64 mask, unsigned* out) {
  unsigned acc = 0;
  for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < n_rec * 8; j += (u64)gridDim.x * 256) {
    const u64 row = j >> 3, ch = j & 7;
    const u64 rec = (row * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & mask;
    const uint4 v = recs[rec * 8 + ch];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345679u) out[0] = acc;
}
static double time_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b)); return ms; }
int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "gather")) {
    // rocprofv3 --pmc <counters> ... pmc_calib gather   (tools/pmc_calibrate.sh): every kernel 3 launches; prints durations too
    const u64 big = 4ull << 30;                                   // 4 GiB window
    char* buf; unsigned* out; CHECK(hipMalloc(&buf, big)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(buf, 1, big));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const u64 rec64 = big / 64, rec128 = big / 128, small64 = (4ull << 20) / 64;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(calib_gather_rec64, dim3(8192), dim3(256), 0, 0, (const uint4*)buf, rec64, rec64 - 1, out); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      printf("gather_rec64       4 GiB window: %8.3f ms  %6.2f G records/s\n", time_ms(e0, e1), rec64 / time_ms(e0, e1) / 1e6);
      CHECK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(calib_gather_rec64_lane, dim3(8192), dim3(256), 0, 0, (const uint4*)buf, rec64, rec64 - 1, out); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      printf("gather_rec64_lane  4 GiB window: %8.3f ms  %6.2f G records/s\n", time_ms(e0, e1), rec64 / time_ms(e0, e1) / 1e6);
      CHECK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(calib_gather_rec128, dim3(8192), dim3(256), 0, 0, (const uint4*)buf, rec128, rec128 - 1, out); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      printf("gather_rec128      4 GiB window: %8.3f ms  %6.2f G records/s\n", time_ms(e0, e1), rec128 / time_ms(e0, e1) / 1e6);
      CHECK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(calib_gather_u64, dim3(8192), dim3(256), 0, 0, (const u64*)buf, big / 8 / 4, big / 8 - 1, out); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      printf("gather_u64 (1 GiB of 8-byte reads over the 4 GiB window): %8.3f ms  %6.2f G reads/s\n", time_ms(e0, e1), (big / 32) / time_ms(e0, e1) / 1e6);
      // the same number of record reads from a 4 MiB window (every record 1024 times): L2-resident
      CHECK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(calib_gather_rec64, dim3(8192), dim3(256), 0, 0, (const uint4*)buf, rec64, small64 - 1, out); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      printf("gather_rec64       4 MiB window: %8.3f ms  %6.2f G records/s\n", time_ms(e0, e1), rec64 / time_ms(e0, e1) / 1e6);
    }
    // where the rate changes: the same 64 M record reads over windows from one XCD's L2 (4 MiB) through the Infinity Cache (256 MiB) to DRAM
    for (u64 mib = 2; mib <= 4096; mib *= 2) {
      const u64 wrec = (mib << 20) / 64;
      CHECK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(calib_gather_rec64, dim3(8192), dim3(256), 0, 0, (const uint4*)buf, rec64, wrec - 1, out); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      printf("window_sweep rec64 %5llu MiB: %8.3f ms  %6.2f G records/s\n", mib, time_ms(e0, e1), rec64 / time_ms(e0, e1) / 1e6);
    }
    CHECK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
  }

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
