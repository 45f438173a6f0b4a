// Microbenchmark (not product code), round 3: does a buffer that one kernel writes and the next kernel reads stay in the 256 MiB
// Infinity Cache?  Alternates write(buf) / read(buf) over buffers of 32 MB .. 2 GB and prints both rates; then the same with the
// producer and consumer working on TWO alternating buffers (ping-pong), as a chunked scatter -> aggregate pipeline would.
//   hipcc --offload-arch=gfx950 -O3 -o _bin/mall_pingpong mall_pingpong.hip && _bin/mall_pingpong
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void write_kernel(uint4* __restrict__ p, u64 n16, unsigned v) {
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += (u64)gridDim.x * 256) p[i] = uint4{v, v + 1, v + 2, (unsigned)i};
}
__global__ __launch_bounds__(256) void read_kernel(const uint4* __restrict__ p, u64 n16, unsigned* __restrict__ out) {
  unsigned acc = 0;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += (u64)gridDim.x * 256) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
// the source of a chunk: a large array streamed once (so that the cache also sees the input traffic of a real pipeline)
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, u64 n16) {
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += (u64)gridDim.x * 256) dst[i] = src[i];
}

int main() {
  const u64 big = 4ull << 30;
  char *src, *a, *b; unsigned* out;
  CHECK(hipMalloc(&src, big)); CHECK(hipMalloc(&a, 2ull << 30)); CHECK(hipMalloc(&b, 2ull << 30)); CHECK(hipMalloc(&out, 64));
  hipLaunchKernelGGL(write_kernel, dim3(4096), dim3(256), 0, 0, (uint4*)src, big / 16, 7u);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1, e2; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
  printf("-- write(buf) then read(buf), same buffer\n");
  for (u64 mb : {32ull, 64ull, 96ull, 128ull, 192ull, 256ull, 512ull, 2048ull}) {
    const u64 bytes = mb << 20; float w = 0, r = 0; const int reps = 8;
    for (int i = 0; i < reps + 2; ++i) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(write_kernel, dim3(2048), dim3(256), 0, 0, (uint4*)a, bytes / 16, (unsigned)i);
      CHECK(hipEventRecord(e1));
      hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(256), 0, 0, (const uint4*)a, bytes / 16, out);
      CHECK(hipEventRecord(e2)); CHECK(hipEventSynchronize(e2));
      float t1, t2; CHECK(hipEventElapsedTime(&t1, e0, e1)); CHECK(hipEventElapsedTime(&t2, e1, e2));
      if (i >= 2) { w += t1; r += t2; }
    }
    printf("%5llu MB: write %6.2f TB/s   read-back %6.2f TB/s\n", mb, bytes / (w / reps) / 1e9, bytes / (r / reps) / 1e9);
  }
  printf("-- pipeline: chunk k: copy(src slice -> buf[k&1]) ; read(buf[k&1])  (input streamed from a 4 GB array)\n");
  for (u64 mb : {32ull, 64ull, 96ull, 128ull, 256ull, 1024ull}) {
    const u64 bytes = mb << 20; const int chunks = (int)(big / bytes);
    CHECK(hipEventRecord(e0));
    for (int k = 0; k < chunks; ++k) {
      char* buf = (k & 1) ? b : a;
      hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, (const uint4*)(src + (u64)k * bytes), (uint4*)buf, bytes / 16);
      hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
    }
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float t; CHECK(hipEventElapsedTime(&t, e0, e1));
    printf("chunks of %5llu MB: 4 GB through copy + read in %.3f ms  (a plain 4 GB read at 6.3 TB/s = 0.68 ms; copy + re-read from HBM = 12 GB = 2.0 ms)\n", mb, t);
  }
  return 0;
}
