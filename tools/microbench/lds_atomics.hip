// Probe: throughput of LDS read-modify-write on a CU -- 64-bit atomics (add f64 / min u64 / returning add),
// 32-bit atomics, and plain (non-atomic) 64-bit read + write pairs -- at random table slots, 1024 threads per
// workgroup, 1 or 2 workgroups per CU.  Decides how the partitioned GroupAggregate folds records into its table.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o lds_atomics lds_atomics.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned long long u64;
typedef unsigned int u32;
extern __shared__ u64 table[];

template <int MODE>
__global__ __launch_bounds__(1024) void k(u32 slots, int iters, u64* out) {
  for (u32 i = threadIdx.x; i < slots; i += 1024) table[i] = 0;
  __syncthreads();
  u32 h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  u64 acc = 0;
  for (int it = 0; it < iters; ++it) {
    h = h * 1664525u + 1013904223u;
    const u32 s = __umulhi(h, slots);
    if (MODE == 0) unsafeAtomicAdd(reinterpret_cast<double*>(&table[s]), 1.0);
    else if (MODE == 1) atomicMin(&table[s], (u64)h);
    else if (MODE == 2) acc += (u64)unsafeAtomicAdd(reinterpret_cast<double*>(&table[s]), 1.0);
    else if (MODE == 3) atomicAdd(reinterpret_cast<u32*>(&table[s]), 1u);
    else if (MODE == 4) acc += atomicAdd(reinterpret_cast<u32*>(&table[s]), 1u);
    else if (MODE == 5) { u64 v = table[s]; table[s] = v + h; }          // plain RMW (racy: throughput only)
    else if (MODE == 6) atomicAdd(&table[s], (u64)h);
    else if (MODE >= 7) {
      // the GroupAggregate row: 12 atomics on words of one random 17-word entry (MODE 7: the DOUBLE adds return)
      const u32 e = __umulhi(h, slots / 17u) * 17u + 1u;
      const double x = (double)(h & 1023u);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (MODE == 7) { const double old = unsafeAtomicAdd(reinterpret_cast<double*>(&table[e + 4 * c]), x); if (old + x - old != x) unsafeAtomicAdd(reinterpret_cast<double*>(&table[e + 4 * c + 1]), 1.0); }
        else if (MODE == 8) { unsafeAtomicAdd(reinterpret_cast<double*>(&table[e + 4 * c]), x); }
        else atomicAdd(&table[e + 4 * c], (u64)h);
        atomicMin(&table[e + 4 * c + 2], (u64)h);
        atomicMax(&table[e + 4 * c + 3], (u64)h);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = table[0] + acc;
}

template <int MODE> void run(const char* name, int wgs_per_cu, u32 slots, u64* out) {
  const int iters = 4096, grid = 256 * wgs_per_cu;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  k<MODE><<<grid, 1024, slots * 8, 0>>>(slots, 16, out);
  hipEventRecord(a);
  k<MODE><<<grid, 1024, slots * 8, 0>>>(slots, iters, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double ops = (double)grid * 1024 * iters * (MODE >= 7 ? 12 : 1);
  printf("%-28s wgs/CU %d slots %6u: %7.3f ms  %8.1f G ops/s  = %.2f lane-ops per clock per CU (2.4 GHz)\n", name, wgs_per_cu, slots, ms,
         ops / ms / 1e6, ops / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
  u64* out; hipMalloc(&out, 8 * 1024);
  for (int w = 1; w <= 2; ++w) {
    const u32 slots = w == 1 ? 16384 : 8192;
    run<0>("ds_add_f64 (no return)", w, slots, out);
    run<1>("ds_min_u64 (no return)", w, slots, out);
    run<6>("ds_add_u64 (no return)", w, slots, out);
    run<2>("ds_add_rtn_f64", w, slots, out);
    run<3>("ds_add_u32 (no return)", w, slots, out);
    run<4>("ds_add_rtn_u32", w, slots, out);
    run<5>("plain read + write b64", w, slots, out);
  }
  for (int w = 1; w <= 2; ++w) {
    run<7>("row: 4 x (rtn f64 add, min, max)", w, 17 * 600, out);
    run<8>("row: 4 x (f64 add, min, max)", w, 17 * 600, out);
    run<9>("row: 4 x (u64 add, min, max)", w, 17 * 600, out);
  }
  run<0>("ds_add_f64, 256 slots", 1, 256, out);
  run<3>("ds_add_u32, 256 slots", 1, 256, out);
  return 0;
}
