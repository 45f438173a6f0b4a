// Microbenchmark (not product code), round 3: the first pass of an MSD radix sort that moves WHOLE rows -- 8 columns of 8 bytes
// read once, one 64-byte record per row appended to the bucket of the key's top bits, over (bucket, XCD) segments with one
// global atomic per (tile, bucket) run (the partition scatter's shape, group_scatter_kernel.hip).  How close to a plain copy
// of the same 12.8 GB does it get, for 256 / 512 / 1024 buckets?
//   hipcc --offload-arch=gfx950 -O3 -o _bin/rec_partition rec_partition.hip && _bin/rec_partition
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32;
typedef unsigned long long u64;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void init_kernel(u64* c0, u64* c1, u64* c2, u64* c3, u64* c4, u64* c5, u64* c6, u64* c7, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    u64 h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    c0[i] = h; c1[i] = i; c2[i] = h + 1; c3[i] = h + 2; c4[i] = h + 3; c5[i] = h + 4; c6[i] = h + 5; c7[i] = h + 6;
  }
}
struct Cols { const u64* c[8]; };

template <int THREADS, int R, int NSEG>
__global__ __launch_bounds__(THREADS) void part_kernel(Cols C, u64 n, u32 NB, u32 shift, u32 cap, u64* __restrict__ out, u32* __restrict__ counts, u32* __restrict__ overflow) {
  constexpr int T = THREADS * R;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32* cnt = reinterpret_cast<u32*>(smem);
  u32* gbase = cnt + NB;
  u32* start = gbase + NB;
  u32* grec = start + NB + 2;
  u64* stage = reinterpret_cast<u64*>(grec + T);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const u32 xcd = NSEG == 8 ? (blockIdx.x & 7u) : 0u;
  for (u32 i = t; i < NB; i += THREADS) cnt[i] = 0;
  __syncthreads();
  const u64 n_tiles = (n + T - 1) / T;
  for (u64 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const u64 base = tile * T;
    u64 v[R][8]; u32 pt[R], pos[R]; bool ok[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const u64 row = base + (u64)j * THREADS + t;
      ok[j] = row < n;
      if (ok[j]) {
#pragma unroll
        for (int f = 0; f < 8; ++f) v[j][f] = C.c[f][row];
        pt[j] = (u32)(v[j][0] >> shift); pos[j] = atomicAdd(&cnt[pt[j]], 1u);
      }
    }
    __syncthreads();
    if (wave == 0) {
      const u32 per = (NB + 63) / 64;
      u32 s = 0;
      for (u32 i = 0; i < per; ++i) { const u32 q = lane * per + i; if (q < NB) s += cnt[q]; }
      u32 inc = s;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d); if (lane >= d) inc += o; }
      u32 ex = inc - s;
      for (u32 i = 0; i < per; ++i) { const u32 q = lane * per + i; if (q < NB) { start[q] = ex; ex += cnt[q]; } }
      if (lane == 63) start[NB] = ex;
    } else {
      for (u32 i = t - 64; i < NB; i += THREADS - 64) { const u32 c = cnt[i]; gbase[i] = c ? atomicAdd(&counts[i * NSEG + xcd], c) : 0u; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < R; ++j) {
      if (!ok[j]) continue;
      const u32 s = start[pt[j]] + pos[j], g = gbase[pt[j]] + pos[j];
      if (g < cap) grec[s] = (pt[j] * NSEG + xcd) * cap + g; else { grec[s] = 0xFFFFFFFFu; *overflow = 1u; }
#pragma unroll
      for (int f = 0; f < 8; ++f) stage[(size_t)s * 8 + f] = v[j][f];
    }
    for (u32 i = t; i < NB; i += THREADS) cnt[i] = 0;
    __syncthreads();
    const u32 words = start[NB] * 8u;
    for (u32 w = t; w < words; w += THREADS) {
      const u32 g = grec[w >> 3];
      if (g != 0xFFFFFFFFu) out[(u64)g * 8u + (w & 7u)] = stage[w];
    }
  }
}

__global__ void copy_kernel(Cols C, u64 n, u64* __restrict__ out) {   // the floor: same bytes, sequential
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
#pragma unroll
    for (int f = 0; f < 8; ++f) out[(u64)f * n + i] = C.c[f][i];
  }
}

template <int THREADS, int R, int NSEG>
static void run(const Cols& C, u64 n, u32 bits, u32 G) {
  constexpr int T = THREADS * R;
  const u32 NB = 1u << bits;
  const size_t lds = (size_t)(3 * NB + 2 + T) * 4 + 16 + (size_t)T * 64;
  if (lds > 160 * 1024) { printf("NB=%u T=%d: LDS %zu too large\n", NB, T, lds); return; }
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&part_kernel<THREADS, R, NSEG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const double expect = (double)n / (NB * (double)NSEG);
  const u32 cap = (u32)(expect * 1.1 + 8.0 * __builtin_sqrt(expect) + 64.0);
  u64* out; u32* counts; u32* overflow;
  CHECK(hipMalloc(&out, (size_t)NB * NSEG * cap * 64)); CHECK(hipMalloc(&counts, (size_t)NB * NSEG * 4)); CHECK(hipMalloc(&overflow, 4));
  CHECK(hipMemset(overflow, 0, 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipMemsetAsync(counts, 0, (size_t)NB * NSEG * 4));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((part_kernel<THREADS, R, NSEG>), dim3(G), dim3(THREADS), lds, 0, C, n, NB, 64 - bits, cap, out, counts, overflow);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  u32 ov; CHECK(hipMemcpy(&ov, overflow, 4, hipMemcpyDeviceToHost));
  printf("%s partition threads=%4d T=%4d buckets=%4u grid=%4u lds=%6zu : %.3f ms  (%.2f TB/s of 12.8 GB)%s\n", NSEG == 8 ? "per-XCD segs " : "shared bucket", THREADS, T, NB, G, lds, best, 12.8 / best, ov ? "  (overflow)" : "");
  fflush(stdout);
  CHECK(hipFree(out)); CHECK(hipFree(counts)); CHECK(hipFree(overflow));
}

// random 64-byte record reads inside windows of W bytes (consecutive windows, each read completely, in random order), output written sequentially
__global__ __launch_bounds__(256) void gather_kernel(const uint4* __restrict__ recs, uint4* __restrict__ out, u64 n, u64 win_recs) {
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n * 4; i += (u64)gridDim.x * 256) {
    const u64 r = i >> 2, q = i & 3;                 // 4 lanes per record, 16 bytes each
    const u64 w = r / win_recs, k = r - w * win_recs;
    u64 h = k * 0x9E3779B97F4A7C15ull; h ^= h >> 31;
    u64 src = w * win_recs + (h % win_recs);
    if (src >= n) src = n - 1;
    out[i] = recs[src * 4 + q];
  }
}

int main() {
  const u64 n = 100000000ull;
  Cols C; u64* c[8];
  for (int f = 0; f < 8; ++f) { CHECK(hipMalloc(&c[f], n * 8)); C.c[f] = c[f]; }
  hipLaunchKernelGGL(init_kernel, dim3(2048), dim3(256), 0, 0, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], n);
  CHECK(hipDeviceSynchronize());
  { u64* out; CHECK(hipMalloc(&out, n * 64)); hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) { CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(copy_kernel, dim3(4096), dim3(256), 0, 0, C, n, out); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 0 && ms < best) best = ms; }
    printf("plain copy of the 8 columns: %.3f ms (%.2f TB/s)\n", best, 12.8 / best); CHECK(hipFree(out)); }
  if (getenv("PART")) for (u32 bits : {8u, 9u}) { run<512, 2, 8>(C, n, bits, 512); run<512, 2, 1>(C, n, bits, 512); run<1024, 2, 1>(C, n, bits, 256); }
  { uint4 *recs, *out; CHECK(hipMalloc(&recs, n * 64)); CHECK(hipMalloc(&out, n * 64));
    hipLaunchKernelGGL(copy_kernel, dim3(4096), dim3(256), 0, 0, C, n, (u64*)recs); CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (u64 wmb : {4ull, 16ull, 25ull, 64ull, 128ull, 512ull, 6104ull}) {
      const u64 win_recs = (wmb << 20) / 64 < n ? (wmb << 20) / 64 : n;
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) { CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(gather_kernel, dim3(8192), dim3(256), 0, 0, recs, out, n, win_recs); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 0 && ms < best) best = ms; }
      printf("random 64-B record gather inside %5llu MB windows: %.3f ms  (%.1f G records/s)\n", wmb, best, n / best / 1e6);
    }
  }
  return 0;
}
