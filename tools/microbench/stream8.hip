// Ceiling probe for the pipeline kernel's input path: NC 8-byte columns, each wave keeps the
// NEXT tile's 16 B/lane x NC loads in flight in registers while it "computes" on the current
// one (a configurable amount of dependent ALU work), persistent workgroups striding tiles.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

#define LD(p) (NT ? __builtin_nontemporal_load(p) : *(p))
template <int NC, int WPE, int NT, int CHUNK>
__global__ __launch_bounds__(256, WPE) void stream_kernel(const char* const* cols, long n_rows, int n_tiles, int work, u64* out) {
  const int t = threadIdx.x;
  u32x4 pf[NC];
  const int per = (n_tiles + gridDim.x - 1) / gridDim.x;
  int tile = CHUNK ? blockIdx.x * per : blockIdx.x;
  const int step = CHUNK ? 1 : gridDim.x;
  const int end = CHUNK ? min(n_tiles, (int)(blockIdx.x + 1) * per) : n_tiles;
  n_tiles = end;
  if (tile < n_tiles)
#pragma unroll
    for (int c = 0; c < NC; ++c) pf[c] = LD(reinterpret_cast<const u32x4*>(cols[c] + ((long)tile * 512 + 2 * t) * 8));
  u64 acc = 0;
  for (; tile < n_tiles; tile += step) {
    u32x4 cur[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cur[c] = pf[c];
    const int nt = tile + step;
    if (nt < n_tiles)
#pragma unroll
      for (int c = 0; c < NC; ++c) pf[c] = LD(reinterpret_cast<const u32x4*>(cols[c] + ((long)nt * 512 + 2 * t) * 8));
#pragma unroll
    for (int c = 0; c < NC; ++c) acc += ((u64)cur[c][0] | ((u64)cur[c][1] << 32)) + ((u64)cur[c][2] | ((u64)cur[c][3] << 32));
    for (int w = 0; w < work; ++w) { acc = acc * 6364136223846793005ull + 1442695040888963407ull; asm volatile("" : "+v"(acc)); }
  }
  for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
  if ((t & 63) == 0) atomicAdd(out, acc);
}

template <int NC, int WPE, int NT, int CHUNK>
static void run(const char* const* dcols, long rows, int wg_per_cu, int work, u64* dout) {
  const int n_tiles = (int)(rows / 512);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((stream_kernel<NC, WPE, NT, CHUNK>), dim3(grid), dim3(256), 0, 0, dcols, rows, n_tiles, work, dout);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<NC, WPE, NT, CHUNK>), dim3(grid), dim3(256), 0, 0, dcols, rows, n_tiles, work, dout);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("nt=%d chunk=%d NC=%d wpe=%d wg/cu=%d work=%d: %.3f ms  %.2f TB/s\n", NT, CHUNK, NC, WPE, wg_per_cu, work, ms, (double)rows * NC * 8 / ms / 1e9);
}

int main(int argc, char** argv) {
  const long rows = 100000000 / 512 * 512;
  const int NC = 8;
  std::vector<const char*> h(NC);
  for (int c = 0; c < NC; ++c) { void* p; if (hipMalloc(&p, rows * 8) != hipSuccess) { printf("alloc failed\n"); return 1; } hipMemset(p, c + 1, rows * 8); h[c] = (const char*)p; }
  const char** dcols; hipMalloc((void**)&dcols, NC * sizeof(char*)); hipMemcpy(dcols, h.data(), NC * sizeof(char*), hipMemcpyHostToDevice);
  u64* dout; hipMalloc((void**)&dout, 8); hipMemset(dout, 0, 8);
  for (int work : {0, 100}) {
    for (int wg : {2, 3, 4}) { run<8, 4, 0, 0>(dcols, rows, wg, work, dout); run<8, 4, 1, 0>(dcols, rows, wg, work, dout); run<8, 4, 0, 1>(dcols, rows, wg, work, dout); run<8, 4, 1, 1>(dcols, rows, wg, work, dout); }
  }
  return 0;
}
