// Microbenchmark (not product code): what a partition scatter that is NOT a tile-VM program could reach.
//
// The GroupAggregate scatter of config #3 writes one 40-byte record (packed key + 4 DOUBLE values) per row into the
// segment of its (hash partition, workgroup); as a VM program its tile is 512 rows (the VM's LDS registers leave no room
// for more at three workgroups per CU), so a partition gets ~0.5 records per tile and every record is a scattered store
// (3.5 ms for 100 M rows; runs of >= 8 records would cost 2.3 ms).  Here the same work as a kernel of its
// own: rows straight from the columns into a partition-ordered LDS staging area of T rows, flushed with consecutive
// lanes on consecutive words.  Prints the time for T = 1024 / 2048 and 128 ... 1024 partitions, sorted and unsorted.
//
//   hipcc --offload-arch=gfx950 -O3 -o _bin/part_scatter part_scatter.hip && _bin/part_scatter
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u32;
typedef unsigned long long u64;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ u32 hash_key(u64 k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33;
  return (u32)k * 0x2C1B3C6Du;
}

__global__ void init_kernel(int* k1, int* k2, double* d0, double* d1, double* d2, double* d3, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    const u64 h = (i * 0x9E3779B97F4A7C15ull) >> 20;
    const u32 g = (u32)(h % 100000ull);
    k1[i] = (int)(g / 317u); k2[i] = (int)(g % 317u);
    d0[i] = (double)(h & 1023); d1[i] = (double)((h >> 10) & 1023); d2[i] = (double)((h >> 20) & 63); d3[i] = (double)((h >> 26) & 63);
  }
}

// T rows per tile (T / 256 per thread), NP partitions, SORTED: records staged in partition order
template <int T, bool SORTED>
__global__ __launch_bounds__(256) void scatter_kernel(const int* __restrict__ k1, const int* __restrict__ k2, const double* __restrict__ d0,
                                                      const double* __restrict__ d1, const double* __restrict__ d2, const double* __restrict__ d3,
                                                      u64 n, u32 NP, u32 cap, u64* __restrict__ out, u32* __restrict__ counts, u32* __restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32* fill = reinterpret_cast<u32*>(smem);
  u32* fill0 = fill + NP;
  u32* start = fill0 + NP;
  u32* grec = start + NP;
  u64* stage = reinterpret_cast<u64*>(grec + T);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const u32 G = gridDim.x, wg = blockIdx.x;
  constexpr int R = T / 256;
  for (u32 i = t; i < NP; i += 256) fill[i] = 0;
  __syncthreads();
  const u64 n_tiles = (n + T - 1) / T;
  for (u64 tile = wg; tile < n_tiles; tile += G) {
    const u64 base = tile * T;
    u64 key[R]; double v[R][4]; u32 pt[R], pos[R]; bool ok[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const u64 row = base + (u64)j * 256 + t;
      ok[j] = row < n;
      if (ok[j]) {
        key[j] = (u64)(u32)k1[row] | ((u64)(u32)k2[row] << 32);
        v[j][0] = d0[row]; v[j][1] = d1[row]; v[j][2] = d2[row]; v[j][3] = d3[row];
      }
    }
    if (SORTED) { for (u32 i = t; i < NP; i += 256) fill0[i] = fill[i]; __syncthreads(); }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      if (ok[j]) { pt[j] = (u32)(((u64)hash_key(key[j]) * NP) >> 32); pos[j] = atomicAdd(&fill[pt[j]], 1u); }
    }
    u32 n_staged;
    if (SORTED) {
      __syncthreads();
      if (wave == 0) {
        const u32 per = (NP + 63) / 64;
        u32 s = 0;
        for (u32 i = 0; i < per; ++i) { const u32 q = lane * per + i; if (q < NP) s += fill[q] - fill0[q]; }
        u32 inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d); if (lane >= d) inc += o; }
        u32 ex = inc - s;
        for (u32 i = 0; i < per; ++i) { const u32 q = lane * per + i; if (q < NP) { start[q] = ex; ex += fill[q] - fill0[q]; } }
      }
      __syncthreads();
    }
    const u64 rows_here = n - base < (u64)T ? n - base : (u64)T;
    n_staged = (u32)rows_here;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      if (!ok[j]) continue;
      const u32 s = SORTED ? start[pt[j]] + (pos[j] - fill0[pt[j]]) : (u32)j * 256u + (u32)t;
      if (pos[j] < cap) grec[s] = (pt[j] * G + wg) * cap + pos[j]; else { grec[s] = 0xFFFFFFFFu; *overflow = 1u; }
      u64* r = stage + (size_t)s * 5;
      r[0] = key[j]; r[1] = __double_as_longlong(v[j][0]); r[2] = __double_as_longlong(v[j][1]); r[3] = __double_as_longlong(v[j][2]); r[4] = __double_as_longlong(v[j][3]);
    }
    __syncthreads();
    const u32 words = n_staged * 5u;
    for (u32 w = t; w < words; w += 256) {
      const u32 j = w / 5u, f = w - j * 5u;
      const u32 g = grec[j];
      if (g != 0xFFFFFFFFu) out[(u64)g * 5u + f] = stage[w];
    }
    __syncthreads();
  }
  for (u32 i = t; i < NP; i += 256) counts[(u64)i * G + wg] = fill[i];
}

template <int T, bool SORTED>
static float run(const int* k1, const int* k2, const double* d0, const double* d1, const double* d2, const double* d3, u64 n, u32 NP, int wgs_per_cu) {
  const size_t lds = (size_t)NP * 12 + (size_t)T * 4 + (size_t)T * 40 + 64;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&scatter_kernel<T, SORTED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const u32 G = 256u * (u32)wgs_per_cu;
  const double expect = (double)n / ((double)NP * G);
  const u32 cap = (u32)(expect * 1.3 + 8.0 * __builtin_sqrt(expect) + 32.0);
  u64* out; u32* counts; u32* overflow;
  CHECK(hipMalloc(&out, (size_t)NP * G * cap * 40)); CHECK(hipMalloc(&counts, (size_t)NP * G * 4)); CHECK(hipMalloc(&overflow, 4));
  CHECK(hipMemset(overflow, 0, 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((scatter_kernel<T, SORTED>), dim3(G), dim3(256), lds, 0, k1, k2, d0, d1, d2, d3, n, NP, cap, out, counts, overflow);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  u32 ov; CHECK(hipMemcpy(&ov, overflow, 4, hipMemcpyDeviceToHost));
  printf("T=%4d %-8s NP=%4u wgs/cu=%d lds=%6zu cap=%5u : %.3f ms%s\n", T, SORTED ? "sorted" : "unsorted", NP, wgs_per_cu, lds, cap, best, ov ? "  (a segment overflowed)" : "");
  CHECK(hipFree(out)); CHECK(hipFree(counts)); CHECK(hipFree(overflow));
  return best;
}

int main() {
  const u64 n = 100000000ull;
  int *k1, *k2; double *d0, *d1, *d2, *d3;
  CHECK(hipMalloc(&k1, n * 4)); CHECK(hipMalloc(&k2, n * 4));
  CHECK(hipMalloc(&d0, n * 8)); CHECK(hipMalloc(&d1, n * 8)); CHECK(hipMalloc(&d2, n * 8)); CHECK(hipMalloc(&d3, n * 8));
  hipLaunchKernelGGL(init_kernel, dim3(2048), dim3(256), 0, 0, k1, k2, d0, d1, d2, d3, n);
  CHECK(hipDeviceSynchronize());
  for (u32 NP : {128u, 256u, 1024u}) {
    run<1024, true>(k1, k2, d0, d1, d2, d3, n, NP, 3);
    run<1024, false>(k1, k2, d0, d1, d2, d3, n, NP, 3);
  }
  run<2048, true>(k1, k2, d0, d1, d2, d3, n, 128, 1);
  run<2048, true>(k1, k2, d0, d1, d2, d3, n, 256, 1);
  run<512, true>(k1, k2, d0, d1, d2, d3, n, 64, 4);
  run<512, false>(k1, k2, d0, d1, d2, d3, n, 1024, 4);
  return 0;
}
