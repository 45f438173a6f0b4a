// sort_kernels.hip -- device kernels of the Sort and AggregateClusters operators.
//
// Sort (supersonic/cursor/core/sort.cc:781-805 SortPermutation + :182-238, gather through
// ViewCursorWithSelectionVector, cursor/infrastructure/view_cursor.cc:97-118): the reference
// sorts an int64 permutation with std::sort and gathers rows 1024 at a time.  Here: an LSD
// radix sort of (64-bit order-preserving key, row id) pairs, 8 bits per pass, stable, one
// pass sequence per key column from the least to the most significant key; NULL ordering
// (first for ASCENDING, last for DESCENDING) is one extra 1-bit pass per nullable key.
// Every pass is HBM-bound: histogram reads the keys once, scatter reads and writes
// (key, id) once; payload columns are gathered exactly once at the end.
//
// AggregateClusters (cursor/core/aggregate_clusters.cc:97-122,286-315): cluster boundary
// flags + exclusive scan give every row its output row ("segment id").
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include "launch.h"

typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int u32;
typedef unsigned char u8;

#define SORT_THREADS 256
#define SORT_ITEMS 16                       /* keys per thread */
#define SORT_TILE (SORT_THREADS * SORT_ITEMS) /* 4096 keys per workgroup */

// ---- key construction -------------------------------------------------------------------
// kind: 0 unsigned, 1 signed, 2 float32, 3 float64, 4 bool
__device__ __forceinline__ u64 order_key(const void* col, u32 width, int kind, u64 row) {
  u64 k;
  if (width == 8) k = reinterpret_cast<const u64*>(col)[row];
  else if (width == 4) k = reinterpret_cast<const u32*>(col)[row];
  else k = reinterpret_cast<const u8*>(col)[row];
  switch (kind) {
    case 1: k = width == 8 ? (k ^ 0x8000000000000000ull) : (u64)((u32)k ^ 0x80000000u); break;
    // -0.0 compares equal to +0.0 (ThreeWayCompare, types_infrastructure.h:238-246): one key for both, so
    // that the stable sort leaves their relative order alone, as the reference's comparison sort does
    case 2: { u32 b = (u32)k; if (b == 0x80000000u) b = 0u; b = (b & 0x80000000u) ? ~b : (b | 0x80000000u); k = b; } break;
    case 3: if (k == 0x8000000000000000ull) k = 0ull; k = (k & 0x8000000000000000ull) ? ~k : (k | 0x8000000000000000ull); break;
    default: break;
  }
  return k;
}

__global__ void ssgpu_sort_iota_kernel(u32* __restrict__ idx, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = (u32)i;
}

// keys[i] = transform(col[idx[i]]) (descending: complemented), or the NULL-order bit.
// Also folds OR / AND of all keys into bits[0] / bits[1]: a radix digit whose bits agree in
// every key (OR == AND there) needs no pass at all -- real keys (dates, small integers, ids)
// rarely use all 64 bits.
#define LOAD_KEYS_PER_THREAD 8
__global__ __launch_bounds__(256) void ssgpu_sort_load_keys_kernel(u64* __restrict__ keys, const u32* __restrict__ idx, const void* __restrict__ col,
                                            const u8* __restrict__ nulls, u32 width, int kind, int descending,
                                            int null_pass, u64 n, unsigned long long* __restrict__ bits) {
  __shared__ u64 red[2][4];
  u64 vor = 0, vand = ~0ull;
  const u64 base = (u64)blockIdx.x * (256u * LOAD_KEYS_PER_THREAD) + threadIdx.x;
#pragma unroll
  for (int j = 0; j < LOAD_KEYS_PER_THREAD; ++j) {
    const u64 i = base + (u64)j * 256u;
    if (i >= n) break;
    const u64 row = idx ? idx[i] : i;   // no row ids: the keys-only sort of a single column
    u64 k;
    if (null_pass) {
      const bool isnull = nulls && nulls[row];
      k = descending ? (isnull ? 1ull : 0ull) : (isnull ? 0ull : 1ull);   // NULLs first ASC, last DESC
    } else {
      k = order_key(col, width, kind, row);
      if (descending) k = ~k;
      if (nulls && nulls[row]) k = 0;   // value of a NULL row is unspecified: make ties deterministic
    }
    keys[i] = k;
    vor |= k; vand &= k;
  }
  for (int o = 32; o; o >>= 1) { vor |= __shfl_xor(vor, o); vand &= __shfl_xor(vand, o); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = vor; red[1][wave] = vand; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicOr(&bits[0], red[0][0] | red[0][1] | red[0][2] | red[0][3]);
    atomicAnd(&bits[1], red[1][0] & red[1][1] & red[1][2] & red[1][3]);
  }
}

// per-tile digit histogram -> hist[digit * n_tiles + tile]
__global__ __launch_bounds__(SORT_THREADS) void ssgpu_sort_hist_kernel(const u64* __restrict__ keys, u32 shift, u64 n,
                                                                       u32 n_tiles, u32* __restrict__ hist) {
  __shared__ u32 h[256];
  const int t = threadIdx.x;
  h[t] = 0;
  __syncthreads();
  const u64 base = (u64)blockIdx.x * SORT_TILE;
#pragma unroll
  for (int j = 0; j < SORT_ITEMS; ++j) {
    const u64 i = base + (u64)j * SORT_THREADS + t;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 0xFF], 1u);
  }
  __syncthreads();
  hist[(u64)t * n_tiles + blockIdx.x] = h[t];
}

// stable scatter: element order inside a tile is (wave, step, lane); ranks of equal digits by
// wave-level match (8 ballots) on top of running per-wave digit counters in LDS.  The tile is
// first reordered by digit INSIDE LDS, then written out: a digit's keys of this tile form one
// contiguous run in the output (4096 / 256 = 16 keys = 128 B on average), so the global writes
// are coalesced instead of 64 random 8-byte stores per wave step.
// HAS_IDX = false: keys only (a single-column sort whose output is the key column itself: no row ids
// travel with the keys and no gather follows).
template <bool HAS_IDX>
__global__ __launch_bounds__(SORT_THREADS) void ssgpu_sort_scatter_kernel(
    const u64* __restrict__ keys_in, const u32* __restrict__ idx_in, u64* __restrict__ keys_out, u32* __restrict__ idx_out,
    u32 shift, u64 n, u32 n_tiles, const u32* __restrict__ offsets) {
  __shared__ u32 wave_cnt[4][256];   // running tile-local positions per wave and digit
  __shared__ u32 scanbuf[256];
  __shared__ u32 goff[256];          // global offset of digit d minus its tile-local start
  __shared__ u64 lk[SORT_TILE];
  __shared__ u32 li[HAS_IDX ? SORT_TILE : 1];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const u64 tile_base = (u64)blockIdx.x * SORT_TILE;
  const u64 wave_base = tile_base + (u64)wave * (SORT_TILE / 4);
  // phase 1: per-wave histograms
  for (int d = lane; d < 256; d += 64) wave_cnt[wave][d] = 0;
  __syncthreads();
  u64 k[SORT_ITEMS]; u32 id[SORT_ITEMS];
#pragma unroll
  for (int j = 0; j < SORT_ITEMS; ++j) {
    const u64 i = wave_base + (u64)j * 64 + lane;
    const bool ok = i < n;
    k[j] = ok ? keys_in[i] : ~0ull;
    id[j] = (HAS_IDX && ok) ? idx_in[i] : 0u;
    if (ok) atomicAdd(&wave_cnt[wave][(k[j] >> shift) & 0xFF], 1u);
  }
  __syncthreads();
  // exclusive scan of the tile's digit totals (thread t = digit t) -> tile-local run starts
  const u32 tot = wave_cnt[0][t] + wave_cnt[1][t] + wave_cnt[2][t] + wave_cnt[3][t];
  scanbuf[t] = tot;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const u32 v = t >= o ? scanbuf[t - o] : 0u;
    __syncthreads();
    scanbuf[t] += v;
    __syncthreads();
  }
  {
    const u32 excl = scanbuf[t] - tot;
    goff[t] = offsets[(u64)t * n_tiles + blockIdx.x] - excl;
    u32 run = excl;
    for (int w = 0; w < 4; ++w) { const u32 c = wave_cnt[w][t]; wave_cnt[w][t] = run; run += c; }
  }
  const u32 valid = scanbuf[255];
  __syncthreads();
  // phase 2: rank and place into LDS, 64 consecutive elements per step
  const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = 0; j < SORT_ITEMS; ++j) {
    const u64 i = wave_base + (u64)j * 64 + lane;
    const bool ok = i < n;
    const u32 d = (u32)(k[j] >> shift) & 0xFF;
    u64 peers = __ballot(ok);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const u64 bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    if (ok) {
      const u32 rank = (u32)__popcll(peers & lt);
      const u32 pos = wave_cnt[wave][d] + rank;
      lk[pos] = k[j];
      if (HAS_IDX) li[pos] = id[j];
    }
    // one lane per digit group advances the running counter (after every lane has read it)
    __builtin_amdgcn_wave_barrier();
    if (ok && (peers & lt) == 0) wave_cnt[wave][d] += (u32)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // phase 3: coalesced write-out of the digit-ordered tile
#pragma unroll
  for (int j = 0; j < SORT_ITEMS; ++j) {
    const u32 e = (u32)j * SORT_THREADS + (u32)t;
    if (e < valid) {
      const u64 key = lk[e];
      const u32 pos = goff[(u32)(key >> shift) & 0xFF] + e;
      keys_out[pos] = key;
      if (HAS_IDX) idx_out[pos] = li[e];
    }
  }
}

// ---- one-sweep LSD passes ------------------------------------------------------------------------------------------
// (a) ONE read of the key column produces the transformed keys AND the global histograms of all eight digits
//     (persistent workgroups, LDS histograms, one flush each) -- instead of one histogram kernel + a 3-launch scan per
//     pass; (b) every scatter pass finds its tile's global offsets itself, by decoupled look-back over per-tile status
//     words, so a pass is ONE kernel that reads and writes the (key, row id) pairs once.
// Status word of (tile, digit): state (bits 63-62: 1 = this tile's count, 2 = inclusive prefix up to and including this
// tile) | epoch (bits 61-32: status words of earlier passes / sorts carry another epoch and read as "not ready") | count
// (bits 31-0).  One naturally aligned 8-byte word written by one relaxed agent-scope store and polled with relaxed
// agent-scope loads: data and flag travel together, no fence (MI355X_MICROARCH.md, persistent kernels: handoff-1to1).
// Tiles are numbered by a ticket taken at workgroup start, so every predecessor of a running tile is itself running or
// done, whatever the dispatch order: the look-back cannot deadlock.
__global__ __launch_bounds__(256) void ssgpu_sort_load_keys_hist_kernel(u64* __restrict__ keys, const u32* __restrict__ idx, const void* __restrict__ col,
                                                 const u8* __restrict__ nulls, u32 width, int kind, int descending,
                                                 int null_pass, u64 n, unsigned long long* __restrict__ bits, u32* __restrict__ hist8,
                                                 u64* __restrict__ compact_out) {
  // compact_out (only with idx == NULL): also the one-word form (high half << 32 | row) of every key ("wide keys" below)
  __shared__ u32 h[8][256];
  __shared__ u64 red[2][4];
  const int t = threadIdx.x;
  for (int d = 0; d < 8; ++d) h[d][t] = 0;
  __syncthreads();
  u64 vor = 0, vand = ~0ull;
  for (u64 base = (u64)blockIdx.x * (256u * LOAD_KEYS_PER_THREAD); base < n; base += (u64)gridDim.x * (256u * LOAD_KEYS_PER_THREAD)) {
#pragma unroll
    for (int j = 0; j < LOAD_KEYS_PER_THREAD; ++j) {
      const u64 i = base + (u64)j * 256u + t;
      if (i >= n) break;
      const u64 row = idx ? idx[i] : i;
      u64 k;
      if (null_pass) {
        const bool isnull = nulls && nulls[row];
        k = descending ? (isnull ? 1ull : 0ull) : (isnull ? 0ull : 1ull);
      } else {
        k = order_key(col, width, kind, row);
        if (descending) k = ~k;
        if (nulls && nulls[row]) k = 0;
      }
      keys[i] = k;
      if (compact_out) compact_out[i] = (k & 0xFFFFFFFF00000000ull) | i;
      vor |= k; vand &= k;
#pragma unroll
      for (int d = 0; d < 8; ++d) atomicAdd(&h[d][(k >> (8 * d)) & 0xFF], 1u);
    }
  }
  for (int o = 32; o; o >>= 1) { vor |= __shfl_xor(vor, o); vand &= __shfl_xor(vand, o); }
  const int lane = t & 63, wave = t >> 6;
  if (lane == 0) { red[0][wave] = vor; red[1][wave] = vand; }
  __syncthreads();
  if (t == 0) {
    atomicOr(&bits[0], red[0][0] | red[0][1] | red[0][2] | red[0][3]);
    atomicAnd(&bits[1], red[1][0] & red[1][1] & red[1][2] & red[1][3]);
  }
  for (int d = 0; d < 8; ++d) if (h[d][t]) atomicAdd(&hist8[d * 256 + t], h[d][t]);
}

// exclusive scan of each digit's 256-bin histogram (one workgroup per digit)
__global__ __launch_bounds__(256) void ssgpu_sort_scan_hist8_kernel(const u32* __restrict__ hist8, u32* __restrict__ base8) {
  __shared__ u32 s[256];
  const int t = threadIdx.x, d = blockIdx.x;
  const u32 v = hist8[d * 256 + t];
  s[t] = v;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const u32 x = t >= o ? s[t - o] : 0u;
    __syncthreads();
    s[t] += x;
    __syncthreads();
  }
  base8[d * 256 + t] = s[t] - v;
}

#define ONESWEEP_AGG (1ull << 62)
#define ONESWEEP_PREFIX (2ull << 62)
#ifndef SSGPU_ONESWEEP_LOOKBACK
#define SSGPU_ONESWEEP_LOOKBACK 4      /* measured per pass, 100 M one-word keys: 1: 0.73 ms, 2: 0.70, 4: 0.68, 8: 0.71, 16: 0.74 */
#endif
// Where a pass's time goes (round 4, 100 M one-word keys, parts of the kernel switched off one at a time, same box; 0.79 ms whole):
//   key loads alone 0.19 | + histogram 0.17 | + look-back 0.14 | + ranking 0.13 | + write-out 0.19 ms -- the phases add up (two
//   512-thread workgroups per CU overlap little), and with histogram and ranking fused (below) the pass without its look-back
//   drops from 0.65 to 0.56 ms while the whole pass stays at 0.77: the chain of tiles waiting for their predecessors' counts sets
//   the pace (profiles/r04_sort_onesweep_attribution.json).
template <bool HAS_IDX, int THREADS, int LB = SSGPU_ONESWEEP_LOOKBACK>
__global__ __launch_bounds__(THREADS) void ssgpu_sort_onesweep_kernel(
    const u64* __restrict__ keys_in, const u32* __restrict__ idx_in, u64* __restrict__ keys_out, u32* __restrict__ idx_out,
    u32 shift, u64 n, const u32* __restrict__ digit_base, unsigned long long* __restrict__ status, u32* __restrict__ ticket,
    u64 epoch, u32* __restrict__ stuck) {
  constexpr int NW = THREADS / 64, TILE = THREADS * SORT_ITEMS;
  __shared__ u32 wave_cnt[NW][256];  // running tile-local positions per wave and digit
  __shared__ u32 scanbuf[256];
  __shared__ u32 goff[256];          // global offset of digit d minus its tile-local start
  __shared__ u64 lk[TILE];
  __shared__ u32 li[HAS_IDX ? TILE : 1];
  __shared__ u32 tile_s;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) tile_s = atomicAdd(ticket, 1u);
  for (int d = lane; d < 256; d += 64) wave_cnt[wave][d] = 0;
  __syncthreads();
  const u32 tile = tile_s;
  const u64 tile_base = (u64)tile * TILE;
  const u64 wave_base = tile_base + (u64)wave * (TILE / NW);
  u64 k[SORT_ITEMS]; u32 id[SORT_ITEMS];
#pragma unroll
  for (int j = 0; j < SORT_ITEMS; ++j) {
    const u64 i = wave_base + (u64)j * 64 + lane;
    const bool ok = i < n;
    k[j] = ok ? keys_in[i] : ~0ull;
    id[j] = (HAS_IDX && ok) ? idx_in[i] : 0u;
  }
  // Histogram AND stable rank in one sweep over the wave's keys (round 4; it used to be one LDS atomic per key here and, after the
  // scan, a second sweep with the same eight ballots per key plus a read-modify-write of the running digit position per step --
  // measured with parts switched off: 0.17 + 0.13 of a 0.79 ms pass).  Per step of 64 keys: the lanes that share a digit find each
  // other by ballots, the lowest of them (the leader) adds the group's size to the wave's counter of that digit with ONE returning
  // LDS atomic -- different leaders hit different counters -- and what it gets back is the number of keys with that digit in the
  // wave's EARLIER steps; every lane's rank among the wave's keys of its digit is that number plus the lanes of its group below it.
  // When the sweep ends the counters hold the histogram, and placing a key later is one add.
  const u64 lt = (1ull << lane) - 1ull;
  u32 off[SORT_ITEMS];
#pragma unroll
  for (int j = 0; j < SORT_ITEMS; ++j) {
    const u64 i = wave_base + (u64)j * 64 + lane;
    const bool ok = i < n;
    const u32 d = (u32)(k[j] >> shift) & 0xFF;
    u64 peers = __ballot(ok);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const u64 bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const int leader = ok ? __ffsll((long long)peers) - 1 : lane;
    u32 before = 0;
    if (ok && lane == leader) before = atomicAdd(&wave_cnt[wave][d], (u32)__popcll(peers));
    before = (u32)__shfl((int)before, leader);
    off[j] = before + (u32)__popcll(peers & lt);
  }
  __syncthreads();
  u32 tot = 0, prefix = 0;
  if (t < 256) {
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += wave_cnt[w][t];
    // publish this tile's count of digit t, then look back for the sum of all earlier tiles
    const u64 tag = (epoch & 0x3FFFFFFFull) << 32;
    unsigned long long* const mine = status + (u64)tile * 256 + t;
    __hip_atomic_store(mine, ONESWEEP_AGG | tag | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // LB predecessors per round trip: the walk is a chain of dependent agent-scope loads (the next address depends on
    // whether this word already is a prefix), so its length in round trips, not its loads, is what a tile waits for
    for (u32 j = tile; j > 0;) {
      const u32 nb = j < (u32)LB ? j : (u32)LB;
      u64 v[LB];
#pragma unroll
      for (int b = 0; b < LB; ++b)
        v[b] = (u32)b < nb ? __hip_atomic_load(status + (u64)(j - 1 - b) * 256 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      bool done = false;
#pragma unroll
      for (int b = 0; b < LB; ++b) {
        if ((u32)b < nb && !done) {
          const unsigned long long* const p = status + (u64)(j - 1 - b) * 256 + t;
          u64 x = v[b];
          u32 spins = 0;
          while ((x >> 62) == 0 || (x & (0x3FFFFFFFull << 32)) != tag) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 24)) { atomicExch(stuck, 1u); break; }      // never expected: give up rather than hang the device
            x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          prefix += (u32)x;
          done = (x >> 62) == 2;
        }
      }
      if (done) break;
      j -= nb;
    }
    __hip_atomic_store(mine, ONESWEEP_PREFIX | tag | (u64)(prefix + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    scanbuf[t] = tot;
  }
  // exclusive scan of the tile's digit totals (thread t = digit t) -> tile-local run starts
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const u32 v = (t < 256 && t >= o) ? scanbuf[t - o] : 0u;
    __syncthreads();
    if (t < 256) scanbuf[t] += v;
    __syncthreads();
  }
  if (t < 256) {
    const u32 excl = scanbuf[t] - tot;
    goff[t] = digit_base[t] + prefix - excl;
    u32 run = excl;
    for (int w = 0; w < NW; ++w) { const u32 c = wave_cnt[w][t]; wave_cnt[w][t] = run; run += c; }
  }
  __syncthreads();
  const u32 valid = scanbuf[255];
  // place into LDS in digit order: the wave's first position of the digit (wave_cnt, set above) + the key's rank among the wave's keys of it
#pragma unroll
  for (int j = 0; j < SORT_ITEMS; ++j) {
    const u64 i = wave_base + (u64)j * 64 + lane;
    if (i < n) {
      const u32 pos = wave_cnt[wave][(u32)(k[j] >> shift) & 0xFF] + off[j];
      lk[pos] = k[j];
      if (HAS_IDX) li[pos] = id[j];
    }
  }
  __syncthreads();
  // coalesced write-out of the digit-ordered tile
#pragma unroll
  for (int j = 0; j < SORT_ITEMS; ++j) {
    const u32 e = (u32)j * THREADS + (u32)t;
    if (e < valid) {
      const u64 key = lk[e];
      const u32 pos = goff[(u32)(key >> shift) & 0xFF] + e;
      keys_out[pos] = key;
      if (HAS_IDX) idx_out[pos] = li[e];
    }
  }
}

// ---- wide keys: sort by the high half, then fix up the ties ---------------------------------------------------------
// A 64-bit key whose high 32 bits already separate almost all rows (100 M uniform keys: 1 % of the rows share their
// high half with a neighbour, in runs of 2 or 3) does not need the four low-digit passes: after the stable passes over
// the high digits the rows of one high half are adjacent and in input order, and sorting each such run by the low half
// (stable insertion sort, in place, one thread per run) completes the order.  A run longer than SORT_TIE_RUN_MAX raises a
// flag instead: the host then sorts all digits in LSD order as usual.
#define SORT_TIE_RUN_MAX 64u
// A thread looks at SORT_TIE_ROWS consecutive rows (two 16-byte loads and the two neighbours) and fixes the runs that START
// among them: almost every row is a run of one, so the kernel is a streaming read of the keys.
#define SORT_TIE_ROWS 4
template <bool HAS_IDX>
__device__ __forceinline__ void sort_fix_run(u64* __restrict__ keys, u32* __restrict__ idx, u64 n, u32 hi_shift, u32* __restrict__ too_long, u64 i) {
  const u64 h = keys[i] >> hi_shift;
  u32 len = 2;
  while (i + len < n && len <= SORT_TIE_RUN_MAX && (keys[i + len] >> hi_shift) == h) ++len;
  if (len > SORT_TIE_RUN_MAX) { if (*too_long == 0u) atomicExch(too_long, 1u); return; }   // the host sorts all digits instead
  for (u32 a = 1; a < len; ++a) {                             // stable insertion sort of the run by the whole key
    const u64 k = keys[i + a]; const u32 r = HAS_IDX ? idx[i + a] : 0u;
    u32 b = a;
    while (b > 0 && keys[i + b - 1] > k) { keys[i + b] = keys[i + b - 1]; if (HAS_IDX) idx[i + b] = idx[i + b - 1]; --b; }
    keys[i + b] = k; if (HAS_IDX) idx[i + b] = r;
  }
}
template <bool HAS_IDX>
__global__ __launch_bounds__(256) void ssgpu_sort_fix_ties_kernel(u64* __restrict__ keys, u32* __restrict__ idx, u64 n, u32 hi_shift, u32* __restrict__ too_long) {
  const u64 base = ((u64)blockIdx.x * 256 + threadIdx.x) * SORT_TIE_ROWS;
  if (base >= n) return;
  u64 h[SORT_TIE_ROWS + 2];   // high parts of rows base - 1 .. base + SORT_TIE_ROWS; have[] = the row exists (permuting a run never changes them)
  bool have[SORT_TIE_ROWS + 2];
#pragma unroll
  for (int j = 0; j < SORT_TIE_ROWS + 2; ++j) {
    const u64 r = base + (u64)j - 1;
    have[j] = !(j == 0 && base == 0) && r < n;
    h[j] = have[j] ? keys[r] >> hi_shift : 0ull;
  }
#pragma unroll
  for (int j = 1; j <= SORT_TIE_ROWS; ++j) {
    if (!have[j]) break;
    const bool prev_same = have[j - 1] && h[j - 1] == h[j], next_same = have[j + 1] && h[j + 1] == h[j];
    if (!prev_same && next_same) sort_fix_run<HAS_IDX>(keys, idx, n, hi_shift, too_long, base + (u64)j - 1);   // the first row of a run of two or more
  }
}

// The same shortcut with the row id INSIDE the key word.  While only the high half is being sorted the low half is dead
// weight: (high half << 32 | row id) is one 64-bit word per row instead of a key word plus a row-id word, so the four
// passes run as the keys-only kernel (8 bytes per row in, 8 out, instead of 12 and 12).  The low halves are only needed
// for the tie runs, which fetch them from the untouched key array by row id.
// (ssgpu_sort_load_keys_hist_kernel writes the one-word keys next to the full ones: compact_out.)
__device__ __forceinline__ void sort_fix_run_compact(u64* __restrict__ kc, const u64* __restrict__ keys, u64 n, u32* __restrict__ too_long, u64 i) {
  const u32 h = (u32)(kc[i] >> 32);
  u32 len = 2;
  while (i + len < n && len <= SORT_TIE_RUN_MAX && (u32)(kc[i + len] >> 32) == h) ++len;
  if (len > SORT_TIE_RUN_MAX) { if (*too_long == 0u) atomicExch(too_long, 1u); return; }
  for (u32 a = 1; a < len; ++a) {                          // stable insertion sort of the run by the low half of the key
    const u64 k = kc[i + a];
    const u32 lo = (u32)keys[(u32)k];
    u32 b = a;
    while (b > 0 && (u32)keys[(u32)kc[i + b - 1]] > lo) { kc[i + b] = kc[i + b - 1]; --b; }
    kc[i + b] = k;
  }
}
// A run costs its thread a chain of dependent random reads (the low halves live in the key array, by row id); with one
// run per ~100 rows nearly every wave of a thread-per-4-rows kernel holds one and waits for it (0.6 ms for a 0.8 GB read).
// Here a workgroup streams SORT_TIE_WG_ROWS rows (high halves through LDS, coalesced), lists the runs that START among
// them, and then fixes them with as many lanes as there are runs: one latency chain per 2048 rows instead of per 256.
#define SORT_TIE_WG_ROWS 2048
__global__ __launch_bounds__(256) void ssgpu_sort_fix_ties_compact_kernel(u64* __restrict__ kc, const u64* __restrict__ keys, u64 n, u32* __restrict__ too_long) {
  __shared__ u32 hs[SORT_TIE_WG_ROWS + 2];          // high halves of rows base - 1 .. base + SORT_TIE_WG_ROWS
  __shared__ u32 runs[SORT_TIE_WG_ROWS / 2 + 1];    // a run holds at least two rows
  __shared__ u32 n_runs;
  const u64 base = (u64)blockIdx.x * SORT_TIE_WG_ROWS;
  const u32 t = threadIdx.x;
  if (t == 0) n_runs = 0;
  for (u32 e = t; e < SORT_TIE_WG_ROWS + 2; e += 256) {
    const u64 r = base + e;                          // row r - 1
    hs[e] = (r >= 1 && r - 1 < n) ? (u32)(kc[r - 1] >> 32) : 0u;
  }
  __syncthreads();
  for (u32 e = t; e < SORT_TIE_WG_ROWS; e += 256) {
    const u64 r = base + e;
    if (r + 1 >= n) break;
    const bool prev_same = r > 0 && hs[e] == hs[e + 1], next_same = hs[e + 2] == hs[e + 1];
    if (!prev_same && next_same) runs[atomicAdd(&n_runs, 1u)] = e;       // the first row of a run of two or more
  }
  __syncthreads();
  const u32 nr = n_runs;
  for (u32 q = t; q < nr; q += 256) sort_fix_run_compact(kc, keys, n, too_long, base + runs[q]);
}
__global__ __launch_bounds__(256) void ssgpu_sort_extract_idx_kernel(u32* __restrict__ idx, const u64* __restrict__ kc, u64 n) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i < n) idx[i] = (u32)kc[i];
}

// ---- payload as records ---------------------------------------------------------------------------------------------
// Gathering P payload columns one by one costs P random 8-byte reads per row, each of which moves a whole 64-byte sector
// (13.8 of the 25.7 ms of the 8-column sort).  With three or more gathered columns the rows are first packed into
// fixed-stride records (one coalesced pass) and the sorted order then fetches ONE record per row: every byte of every
// sector read is used.  Both kernels transpose through LDS so that global accesses are 16 bytes per lane, consecutive
// lanes on consecutive addresses.
__global__ __launch_bounds__(256) void ssgpu_sort_pack_kernel(const SortRecParams P) {
  extern __shared__ __attribute__((aligned(16))) char tile[];
  const int t = threadIdx.x;
  const u64 base = (u64)blockIdx.x * 256;
  const u64 row = base + t;
  const u32 S = P.stride;
  if (row < P.n) {
    for (u32 c = 0; c < P.n_fields; ++c) {
      const SortRecField f = P.fields[c];
      char* dst = tile + (u32)t * S + f.off;
      if (f.width == 8) *reinterpret_cast<u64*>(dst) = reinterpret_cast<const u64*>(f.src)[row];
      else if (f.width == 4) *reinterpret_cast<u32*>(dst) = reinterpret_cast<const u32*>(f.src)[row];
      else *reinterpret_cast<u8*>(dst) = f.src ? reinterpret_cast<const u8*>(f.src)[row] : (u8)0;
    }
  }
  __syncthreads();
  const u64 rows_here = P.n - base < 256 ? P.n - base : 256;
  const u32 chunks = (u32)(rows_here * S / 16);
  uint4* out = reinterpret_cast<uint4*>(reinterpret_cast<char*>(P.recs) + base * S);
  for (u32 j = (u32)t; j < chunks; j += 256) out[j] = reinterpret_cast<const uint4*>(tile)[j];
}

// The same gather for records of CPR x 16 bytes, written for what bounds it: every row is one random 16 x CPR-byte read, and
// a random read is a DRAM round trip.  The generic kernel below loops `load -> wait -> LDS store` once per 16-byte chunk (the
// compiler keeps the wait inside the loop: the trip count is a run-time value), i.e. CPR dependent round trips per thread;
// here all CPR loads of a thread are in flight before the first wait.  Measured with tools/microbench/pmc_calib.hip
// (round 4): random 64-byte reads issued this way run at 43 G records/s over a 4 GiB window; the counters say one 64-byte
// request per record (FETCH_SIZE is exact for them -- it is halved only for 128-byte streaming requests).
// The output columns are written with nontemporal stores: nothing reads them again before the kernel ends.
// (Round 4 also tried packing the records in key-BUCKET order -- a stable one-sweep partition of whole records by the key's top
//  digit instead of the row-order pack, so that this gather reads inside a 25 MB bucket (on chip: 230 G records/s) instead of
//  all over 6.4 GB.  Measured, same box: the gather 3.45 -> 2.95 ms (it is at copy speed: 13 GB move either way), the
//  partitioning pack 4.59 ms against 2.25 ms -- 11.4 vs 9.5 ms for the sort.  Removed; profiles/r04_sort_bucketed_experiment.json.)
template <int CPR>
__global__ __launch_bounds__(256) void ssgpu_sort_gather_rec_fast_kernel(const SortRecParams P, const u32* __restrict__ idx, u32 idx_stride) {
  extern __shared__ __attribute__((aligned(16))) char tile[];
  __shared__ u32 rid[256];
  const int t = threadIdx.x;
  const u64 base = (u64)blockIdx.x * 256;
  const u64 rows_here = P.n - base < 256 ? P.n - base : 256;
  rid[t] = (u64)t < rows_here ? idx[(base + t) * idx_stride] : 0u;   // (row 0 exists: the tail's lanes load something harmless)
  __syncthreads();
  uint4 v[CPR];
#pragma unroll
  for (int i = 0; i < CPR; ++i) {
    const u32 j = (u32)t + 256u * (u32)i, r = j / (u32)CPR, ch = j % (u32)CPR;
    v[i] = reinterpret_cast<const uint4*>(P.recs)[(u64)rid[r] * CPR + ch];
  }
#pragma unroll
  for (int i = 0; i < CPR; ++i) reinterpret_cast<uint4*>(tile)[(u32)t + 256u * (u32)i] = v[i];
  __syncthreads();
  if ((u64)t < rows_here) {
    const u64 row = base + t;
    for (u32 c = 0; c < P.n_fields; ++c) {
      const SortRecField f = P.fields[c];
      if (!f.dst) continue;
      const char* src = tile + (u32)t * (16u * CPR) + f.off;
      if (f.width == 8) __builtin_nontemporal_store(*reinterpret_cast<const u64*>(src), reinterpret_cast<u64*>(f.dst) + row);
      else if (f.width == 4) __builtin_nontemporal_store(*reinterpret_cast<const u32*>(src), reinterpret_cast<u32*>(f.dst) + row);
      else reinterpret_cast<u8*>(f.dst)[row] = *reinterpret_cast<const u8*>(src);
    }
  }
}

__global__ __launch_bounds__(256) void ssgpu_sort_gather_rec_kernel(const SortRecParams P, const u32* __restrict__ idx, u32 idx_stride) {
  extern __shared__ __attribute__((aligned(16))) char tile[];
  __shared__ u32 rid[256];
  const int t = threadIdx.x;
  const u64 base = (u64)blockIdx.x * 256;
  const u32 S = P.stride, cpr = S / 16;
  const u64 rows_here = P.n - base < 256 ? P.n - base : 256;
  if ((u64)t < rows_here) rid[t] = idx[(base + t) * idx_stride];   // stride 2: the low words of (high half << 32 | row id) keys
  __syncthreads();
  const u32 chunks = (u32)rows_here * cpr;
  for (u32 j = (u32)t; j < chunks; j += 256) {
    const u32 r = j / cpr, ch = j - r * cpr;
    reinterpret_cast<uint4*>(tile)[j] = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(P.recs) + (u64)rid[r] * S)[ch];
  }
  __syncthreads();
  if ((u64)t < rows_here) {
    const u64 row = base + t;
    for (u32 c = 0; c < P.n_fields; ++c) {
      const SortRecField f = P.fields[c];
      if (!f.dst) continue;
      const char* src = tile + (u32)t * S + f.off;
      if (f.width == 8) reinterpret_cast<u64*>(f.dst)[row] = *reinterpret_cast<const u64*>(src);
      else if (f.width == 4) reinterpret_cast<u32*>(f.dst)[row] = *reinterpret_cast<const u32*>(src);
      else reinterpret_cast<u8*>(f.dst)[row] = *reinterpret_cast<const u8*>(src);
    }
  }
}

// The sorted keys back to column values (integer kinds only: the transform is a bijection there): the key
// column of the result needs no gather -- and nothing else at all when it is the only output column.
__global__ __launch_bounds__(256) void ssgpu_sort_unkey_kernel(void* __restrict__ out, const u64* __restrict__ keys, u32 width, int kind,
                                                               int descending, u64 n) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  u64 k = keys[i];
  if (descending) k = ~k;
  if (kind == 1) k = width == 8 ? (k ^ 0x8000000000000000ull) : (u64)((u32)k ^ 0x80000000u);
  if (width == 8) reinterpret_cast<u64*>(out)[i] = k;
  else if (width == 4) reinterpret_cast<u32*>(out)[i] = (u32)k;
  else reinterpret_cast<u8*>(out)[i] = (u8)k;
}

// out[i] = col[idx[i]] for one column (and its NULL mask)
__global__ void ssgpu_sort_gather_kernel(void* __restrict__ out, u8* __restrict__ out_nulls, const void* __restrict__ col,
                                         const u8* __restrict__ nulls, u32 width, const u32* __restrict__ idx, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 r = idx[i];
  if (r == 0xFFFFFFFFull) {   // "no row" (the rhs side of an unmatched LEFT_OUTER row): a NULL
    if (width == 8) reinterpret_cast<u64*>(out)[i] = 0ull;
    else if (width == 4) reinterpret_cast<u32*>(out)[i] = 0u;
    else reinterpret_cast<u8*>(out)[i] = (u8)0;
    if (out_nulls) out_nulls[i] = (u8)1;
    return;
  }
  if (width == 8) reinterpret_cast<u64*>(out)[i] = reinterpret_cast<const u64*>(col)[r];
  else if (width == 4) reinterpret_cast<u32*>(out)[i] = reinterpret_cast<const u32*>(col)[r];
  else reinterpret_cast<u8*>(out)[i] = reinterpret_cast<const u8*>(col)[r];
  if (out_nulls) out_nulls[i] = nulls ? nulls[r] : (u8)0;
}

// ---- AggregateClusters: boundary flags --------------------------------------------------
struct ClusterKeys { const void* data[16]; const u8* nulls[16]; u32 width[16]; u32 n; };

__device__ __forceinline__ bool cluster_boundary(const ClusterKeys& K, u64 i) {
  if (i == 0) return true;
  for (u32 k = 0; k < K.n; ++k) {
    const bool an = K.nulls[k] && K.nulls[k][i], bn = K.nulls[k] && K.nulls[k][i - 1];
    if (an != bn) return true;
    if (an) continue;   // NULL == NULL (aggregate_clusters.cc:97-122)
    const u32 w = K.width[k];
    if (w == 8) { if (reinterpret_cast<const u64*>(K.data[k])[i] != reinterpret_cast<const u64*>(K.data[k])[i - 1]) return true; }
    else if (w == 4) { if (reinterpret_cast<const u32*>(K.data[k])[i] != reinterpret_cast<const u32*>(K.data[k])[i - 1]) return true; }
    else { if (reinterpret_cast<const u8*>(K.data[k])[i] != reinterpret_cast<const u8*>(K.data[k])[i - 1]) return true; }
  }
  return false;
}

// DISTINCT aggregates: flag[i] = 1 on the first row of every run of equal (keys..., value) in the sorted input
__global__ __launch_bounds__(256) void ssgpu_cluster_flags_kernel(const ClusterKeys K, u64 n, u8* __restrict__ flag) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i < n) flag[i] = cluster_boundary(K, i) ? (u8)1 : (u8)0;
}

// pass 1: per-tile (512 rows) boundary counts
__global__ __launch_bounds__(256) void ssgpu_cluster_count_kernel(const ClusterKeys K, u64 n, u32* __restrict__ tile_counts) {
  __shared__ u32 wsum[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const u64 r0 = (u64)blockIdx.x * 512 + 2 * t;
  const bool f0 = r0 < n && cluster_boundary(K, r0), f1 = (r0 + 1) < n && cluster_boundary(K, r0 + 1);
  const u32 c = (u32)__popcll(__ballot(f0)) + (u32)__popcll(__ballot(f1));
  if (lane == 0) wsum[wave] = c;
  __syncthreads();
  if (t == 0) tile_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// pass 2: seg_id[i] = (#boundaries in [0, i]) - 1; key columns of each new segment are emitted
struct ClusterKeyOut { void* data[16]; u8* nulls[16]; };
__global__ __launch_bounds__(256) void ssgpu_cluster_assign_kernel(const ClusterKeys K, const ClusterKeyOut O, u64 n,
                                                                    const u32* __restrict__ tile_offsets, u32* __restrict__ seg_id) {
  __shared__ u32 wsum[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const u64 r0 = (u64)blockIdx.x * 512 + 2 * t;
  const bool f0 = r0 < n && cluster_boundary(K, r0), f1 = (r0 + 1) < n && cluster_boundary(K, r0 + 1);
  const u64 b0 = __ballot(f0), b1 = __ballot(f1);
  if (lane == 0) wsum[wave] = (u32)__popcll(b0) + (u32)__popcll(b1);
  __syncthreads();
  u32 base = tile_offsets[blockIdx.x];
  for (int w = 0; w < wave; ++w) base += wsum[w];
  const u64 lt = (1ull << lane) - 1ull;
  const u32 before = base + (u32)__popcll(b0 & lt) + (u32)__popcll(b1 & lt);   // boundaries before row r0
  const u32 s0 = before + (f0 ? 1u : 0u) - 1u, s1 = s0 + (f1 ? 1u : 0u);
  const bool ff[2] = {f0, f1}; const u32 ss[2] = {s0, s1};
  for (int j = 0; j < 2; ++j) {
    const u64 r = r0 + j;
    if (r >= n) continue;
    seg_id[r] = ss[j];
    if (!ff[j]) continue;
    for (u32 k = 0; k < K.n; ++k) {
      if (!O.data[k]) continue;   // (segment ids only: Stage::segment_cols)
      const bool isnull = K.nulls[k] && K.nulls[k][r];
      const u32 w = K.width[k];
      if (w == 8) reinterpret_cast<u64*>(O.data[k])[ss[j]] = isnull ? 0ull : reinterpret_cast<const u64*>(K.data[k])[r];
      else if (w == 4) reinterpret_cast<u32*>(O.data[k])[ss[j]] = isnull ? 0u : reinterpret_cast<const u32*>(K.data[k])[r];
      else reinterpret_cast<u8*>(O.data[k])[ss[j]] = isnull ? (u8)0 : reinterpret_cast<const u8*>(K.data[k])[r];
      if (O.nulls[k]) O.nulls[k][ss[j]] = isnull;
    }
  }
}

// ---- SUM of a floating input into an integer result (AddAggregationWithDefinedOutputType) --------------------------------
// aggregation_operators.h:173-185: `*result += val` with an integer result and a floating value adds in the floating type and
// truncates back after EVERY row (the first non-NULL value is assigned: column_aggregator.cc:154-166), so the result depends
// on the row order -- no parallel reduction has it.  The rows arrive in the reference's order within a segment (the input
// order: the stable sort of the sorted shape, or the clusters themselves) and one thread folds each segment, row after row.
// Conversions as everywhere on the path: floating -> integer truncates, out of range / NaN gives INT64_MIN's bits
// (cvttsd2si), the low bytes are stored for the narrower types.
struct SeqSumParams {
  const void* src; const u8* src_nulls; const u32* seg_id;   // seg_id == nullptr: all rows are one segment (ScalarAggregate)
  void* dst; u8* dst_nulls; u64 n; int src_kind, dst_kind;   // kinds: 0 i32, 1 u32, 2 i64, 3 u64, 4 f32, 5 f64
  // the one-segment fold in pieces (so that a host can look at its interrupt flag between them): rows [first, n) of this launch continue
  // from state[0] = the running result's bits, state[1] = "a value was seen" when resume != 0, and leave them there
  u64 first; u64* state; int resume, pad;
};
template <typename F> __device__ __forceinline__ u64 seq_to_bits(F x) {
  const double tr = trunc((double)x);
  return (tr >= -9223372036854775808.0 && tr < 9223372036854775808.0) ? (u64)(i64)tr : 0x8000000000000000ull;
}
template <typename F> __device__ __forceinline__ F seq_from_bits(u64 bits, int dst_kind) {
  switch (dst_kind) {
    case 0: return (F)(int)(u32)bits;
    case 1: return (F)(u32)bits;
    case 2: return (F)(i64)bits;
    default: return (F)bits;
  }
}
__device__ __forceinline__ void seq_store(const SeqSumParams& P, u64 seg, u64 bits, bool any) {
  if (P.dst_kind == 2 || P.dst_kind == 3) reinterpret_cast<u64*>(P.dst)[seg] = any ? bits : 0ull;
  else reinterpret_cast<u32*>(P.dst)[seg] = any ? (u32)bits : 0u;
  if (P.dst_nulls) P.dst_nulls[seg] = any ? (u8)0 : (u8)1;
}
template <typename F>
__device__ __forceinline__ void seq_step(const SeqSumParams& P, F v, u64& bits, bool& any) {
  if (!any) { bits = seq_to_bits<F>(v); any = true; }
  else bits = seq_to_bits<F>(seq_from_bits<F>(bits, P.dst_kind) + v);
  if (P.dst_kind == 0 || P.dst_kind == 1) bits = (u64)(u32)bits;      // (the result lives in its own width between the rows)
}
template <typename F>
__global__ __launch_bounds__(256) void ssgpu_seq_sum_segments_kernel(const SeqSumParams P) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= P.n) return;
  const u32 seg = P.seg_id[i];
  if (i != 0 && P.seg_id[i - 1] == seg) return;        // not a segment's first row
  u64 bits = 0; bool any = false;
  for (u64 r = i; r < P.n && P.seg_id[r] == seg; ++r) {
    if (P.src_nulls && P.src_nulls[r]) continue;
    seq_step<F>(P, reinterpret_cast<const F*>(P.src)[r], bits, any);
  }
  seq_store(P, seg, bits, any);
}
// one segment (ScalarAggregate): ONE wavefront loads 64 rows at a time, coalesced, and lane 0 folds them in row order
template <typename F>
__global__ __launch_bounds__(64) void ssgpu_seq_sum_all_kernel(const SeqSumParams P) {
  const int lane = threadIdx.x;
  u64 bits = 0; bool any = false;
  if (P.resume && P.state) { bits = P.state[0]; any = P.state[1] != 0ull; }
  for (u64 base = P.first; base < P.n; base += 64) {
    const u64 r = base + (u64)lane;
    F v = F(0); bool ok = false;
    if (r < P.n) { ok = !(P.src_nulls && P.src_nulls[r]); if (ok) v = reinterpret_cast<const F*>(P.src)[r]; }
    const u64 okmask = __ballot(ok);
    if (okmask == 0ull) continue;
    for (int l = 0; l < 64; ++l) {
      const F x = __shfl(v, l);
      if ((okmask >> l) & 1ull) seq_step<F>(P, x, bits, any);
    }
  }
  if (lane == 0) { seq_store(P, 0, bits, any); if (P.state) { P.state[0] = bits; P.state[1] = any ? 1ull : 0ull; } }
}
hipError_t ssgpu_launch_seq_sum(const void* src, const uint8_t* src_nulls, int src_kind, const uint32_t* seg_id, uint64_t n,
                                void* dst, uint8_t* dst_nulls, int dst_kind, hipStream_t s, uint64_t first, uint64_t* state, int resume) {
  SeqSumParams P; P.src = src; P.src_nulls = src_nulls; P.seg_id = seg_id; P.dst = dst; P.dst_nulls = dst_nulls; P.n = n; P.src_kind = src_kind; P.dst_kind = dst_kind;
  P.first = first; P.state = reinterpret_cast<u64*>(state); P.resume = resume; P.pad = 0;
  if (seg_id) {
    if (!n) return hipSuccess;
    if (src_kind == 4) hipLaunchKernelGGL(ssgpu_seq_sum_segments_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, P);
    else hipLaunchKernelGGL(ssgpu_seq_sum_segments_kernel<double>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, P);
  } else {
    if (src_kind == 4) hipLaunchKernelGGL(ssgpu_seq_sum_all_kernel<float>, dim3(1), dim3(64), 0, s, P);
    else hipLaunchKernelGGL(ssgpu_seq_sum_all_kernel<double>, dim3(1), dim3(64), 0, s, P);
  }
  return hipGetLastError();
}

// dense extraction of per-segment aggregates (same record layout as the group table)
struct DenseAggOut { void* data; u8* is_null; int s; int out_kind; int has_cnt; int pad; };
struct DenseExtractParams { const u64* acc; const u32* cnt; u32 n_gaggs; u32 n_out; u64 n_rows; DenseAggOut out[VM_MAX_AGG_SLOTS]; };

__device__ __forceinline__ void dense_emit(void* dst, u64 idx, int kind, u64 v0) {
  switch (kind) {
    case EMIT_U64: reinterpret_cast<u64*>(dst)[idx] = v0; break;
    case EMIT_I64KEY: reinterpret_cast<i64*>(dst)[idx] = (i64)(v0 ^ 0x8000000000000000ull); break;
    case EMIT_U32: reinterpret_cast<u32*>(dst)[idx] = (u32)v0; break;
    case EMIT_I32KEY: reinterpret_cast<int*>(dst)[idx] = (int)(i64)(v0 ^ 0x8000000000000000ull); break;
    case EMIT_F64: reinterpret_cast<u64*>(dst)[idx] = v0; break;
    case EMIT_F32: { double d = __longlong_as_double((i64)v0); reinterpret_cast<float*>(dst)[idx] = (float)d; } break;
    case EMIT_U8: reinterpret_cast<u8*>(dst)[idx] = (u8)(v0 != 0); break;
    case EMIT_FKEY_F64: { u64 b = (v0 & 0x8000000000000000ull) ? (v0 & 0x7FFFFFFFFFFFFFFFull) : ~v0; reinterpret_cast<u64*>(dst)[idx] = b; } break;
    case EMIT_FKEY_F32: { u64 b = (v0 & 0x8000000000000000ull) ? (v0 & 0x7FFFFFFFFFFFFFFFull) : ~v0;
                          reinterpret_cast<float*>(dst)[idx] = (float)__longlong_as_double((i64)b); } break;
    default: break;
  }
}
__global__ void ssgpu_dense_extract_kernel(const DenseExtractParams P) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_rows) return;
  for (u32 q = 0; q < P.n_out; ++q) {
    const DenseAggOut o = P.out[q];
    if (o.out_kind == EMIT_DD_F64 || o.out_kind == EMIT_DDRES_F64) {   // compensated DOUBLE sum: (sum, compensation) in consecutive words
      const double hi = __longlong_as_double((i64)P.acc[i * P.n_gaggs + o.s]), lo = __longlong_as_double((i64)P.acc[i * P.n_gaggs + o.s + 1]);
      const double sm = hi + lo, bp = sm - hi;
      reinterpret_cast<double*>(o.data)[i] = o.out_kind == EMIT_DD_F64 ? (lo == 0.0 ? hi : sm) : (lo == 0.0 ? 0.0 : (hi - (sm - bp)) + (lo - bp));   // the sum / its exact residual (TwoSum)
    } else dense_emit(o.data, i, o.out_kind, P.acc[i * P.n_gaggs + o.s]);
    if (o.is_null) o.is_null[i] = o.has_cnt ? (P.cnt[i * P.n_gaggs + o.s] == 0) : 0;
  }
}

// ---- View-file loader: scatter the pieces of a staged slab to their columns (one workgroup per piece) ----
__global__ __launch_bounds__(256) void ssgpu_unpack_kernel(const char* __restrict__ slab, const UnpackPiece* __restrict__ pieces) {
  const UnpackPiece pc = pieces[blockIdx.x];
  const char* src = slab + pc.src_off;
  char* dst = reinterpret_cast<char*>(pc.dst);
  const u64 n = pc.bytes;
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0) {
    const u64 n16 = n / 16;
    for (u64 i = threadIdx.x; i < n16; i += 256) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (u64 i = n16 * 16 + threadIdx.x; i < n; i += 256) dst[i] = src[i];
  } else if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 7u) == 0) {   // the usual case: 8-byte chunk headers
    const u64 n8 = n / 8;
    for (u64 i = threadIdx.x; i < n8; i += 256) reinterpret_cast<u64*>(dst)[i] = reinterpret_cast<const u64*>(src)[i];
    for (u64 i = n8 * 8 + threadIdx.x; i < n; i += 256) dst[i] = src[i];
  } else {
    for (u64 i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
  }
}
hipError_t ssgpu_launch_unpack(const char* slab, const UnpackPiece* pieces, unsigned int n_pieces, hipStream_t s) {
  if (n_pieces) hipLaunchKernelGGL(ssgpu_unpack_kernel, dim3(n_pieces), dim3(256), 0, s, slab, pieces);
  return hipGetLastError();
}

// ---- launchers ------------------------------------------------------------------------------
static inline int blocks_for(uint64_t n, int per) { return (int)((n + per - 1) / per); }

hipError_t ssgpu_launch_sort_iota(uint32_t* idx, uint64_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(ssgpu_sort_iota_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, idx, (u64)n);
  return hipGetLastError();
}
hipError_t ssgpu_launch_sort_load_keys(uint64_t* keys, const uint32_t* idx, const void* col, const uint8_t* nulls, uint32_t width,
                                       int kind, int descending, int null_pass, uint64_t n, unsigned long long* bits, hipStream_t s) {
  if (n) hipLaunchKernelGGL(ssgpu_sort_load_keys_kernel, dim3(blocks_for(n, 256 * LOAD_KEYS_PER_THREAD)), dim3(256), 0, s, (u64*)keys, idx, col, nulls, width,
                            kind, descending, null_pass, (u64)n, bits);
  return hipGetLastError();
}
uint32_t ssgpu_sort_tiles(uint64_t n) { return (uint32_t)((n + SORT_TILE - 1) / SORT_TILE); }
hipError_t ssgpu_launch_sort_hist(const uint64_t* keys, uint32_t shift, uint64_t n, uint32_t* hist, hipStream_t s) {
  const uint32_t nt = ssgpu_sort_tiles(n);
  if (nt) hipLaunchKernelGGL(ssgpu_sort_hist_kernel, dim3(nt), dim3(SORT_THREADS), 0, s, (const u64*)keys, shift, (u64)n, nt, hist);
  return hipGetLastError();
}
hipError_t ssgpu_launch_sort_scatter(const uint64_t* keys_in, const uint32_t* idx_in, uint64_t* keys_out, uint32_t* idx_out,
                                     uint32_t shift, uint64_t n, const uint32_t* offsets, hipStream_t s) {
  const uint32_t nt = ssgpu_sort_tiles(n);
  if (nt && idx_in) hipLaunchKernelGGL(ssgpu_sort_scatter_kernel<true>, dim3(nt), dim3(SORT_THREADS), 0, s, (const u64*)keys_in, idx_in, (u64*)keys_out,
                                       idx_out, shift, (u64)n, nt, offsets);
  else if (nt) hipLaunchKernelGGL(ssgpu_sort_scatter_kernel<false>, dim3(nt), dim3(SORT_THREADS), 0, s, (const u64*)keys_in, idx_in, (u64*)keys_out,
                                  idx_out, shift, (u64)n, nt, offsets);
  return hipGetLastError();
}
hipError_t ssgpu_launch_sort_load_keys_hist(uint64_t* keys, const uint32_t* idx, const void* col, const uint8_t* nulls, uint32_t width, int kind,
                                            int descending, int null_pass, uint64_t n, unsigned long long* bits, uint32_t* hist8, uint32_t* base8, hipStream_t s,
                                            uint64_t* compact_out) {
  if (!n) return hipSuccess;
  const int grid = (int)std::min<uint64_t>(blocks_for(n, 256 * LOAD_KEYS_PER_THREAD), 2048);
  hipLaunchKernelGGL(ssgpu_sort_load_keys_hist_kernel, dim3(grid), dim3(256), 0, s, (u64*)keys, idx, col, nulls, width, kind, descending, null_pass, (u64)n, bits, hist8, idx ? nullptr : (u64*)compact_out);
  hipLaunchKernelGGL(ssgpu_sort_scan_hist8_kernel, dim3(8), dim3(256), 0, s, (const u32*)hist8, base8);
  return hipGetLastError();
}
#ifndef SSGPU_ONESWEEP_THREADS
#define SSGPU_ONESWEEP_THREADS 512      /* tile = 16 keys per thread: 8192 keys, a digit's run of a tile averages 256 bytes */
#endif
uint32_t ssgpu_onesweep_tiles(uint64_t n) { return (uint32_t)((n + (uint64_t)SSGPU_ONESWEEP_THREADS * SORT_ITEMS - 1) / ((uint64_t)SSGPU_ONESWEEP_THREADS * SORT_ITEMS)); }
hipError_t ssgpu_launch_sort_onesweep(const uint64_t* keys_in, const uint32_t* idx_in, uint64_t* keys_out, uint32_t* idx_out, uint32_t shift, uint64_t n,
                                      const uint32_t* digit_base, unsigned long long* status, uint32_t* ticket, uint64_t epoch, uint32_t* stuck, hipStream_t s) {
  const uint32_t nt = ssgpu_onesweep_tiles(n);
  if (nt && idx_in) hipLaunchKernelGGL((ssgpu_sort_onesweep_kernel<true, SSGPU_ONESWEEP_THREADS>), dim3(nt), dim3(SSGPU_ONESWEEP_THREADS), 0, s, (const u64*)keys_in, idx_in, (u64*)keys_out, idx_out,
                                       shift, (u64)n, digit_base, status, ticket, (u64)epoch, stuck);
  else if (nt) hipLaunchKernelGGL((ssgpu_sort_onesweep_kernel<false, SSGPU_ONESWEEP_THREADS>), dim3(nt), dim3(SSGPU_ONESWEEP_THREADS), 0, s, (const u64*)keys_in, idx_in, (u64*)keys_out, idx_out,
                                  shift, (u64)n, digit_base, status, ticket, (u64)epoch, stuck);   // (1024-thread tiles for the keys-only form measured the same)
  return hipGetLastError();
}
hipError_t ssgpu_launch_sort_fix_ties(uint64_t* keys, uint32_t* idx, uint64_t n, uint32_t hi_shift, uint32_t* too_long, hipStream_t s) {
  if (n && idx) hipLaunchKernelGGL(ssgpu_sort_fix_ties_kernel<true>, dim3(blocks_for(n, 256 * SORT_TIE_ROWS)), dim3(256), 0, s, (u64*)keys, idx, (u64)n, hi_shift, too_long);
  else if (n) hipLaunchKernelGGL(ssgpu_sort_fix_ties_kernel<false>, dim3(blocks_for(n, 256 * SORT_TIE_ROWS)), dim3(256), 0, s, (u64*)keys, idx, (u64)n, hi_shift, too_long);
  return hipGetLastError();
}
hipError_t ssgpu_launch_sort_pack(const SortRecParams& P, hipStream_t s) {
  if (P.n) hipLaunchKernelGGL(ssgpu_sort_pack_kernel, dim3(blocks_for(P.n, 256)), dim3(256), 256 * P.stride, s, P);
  return hipGetLastError();
}
hipError_t ssgpu_launch_sort_gather_rec(const SortRecParams& P, const uint32_t* idx, uint32_t idx_stride, hipStream_t s) {
  if (!P.n) return hipGetLastError();
  const dim3 g(blocks_for(P.n, 256)), b(256);
  switch (P.stride % 16u == 0 ? P.stride / 16u : 0u) {   // records of up to 128 bytes: every chunk load of a thread in flight at once
    case 1: hipLaunchKernelGGL(ssgpu_sort_gather_rec_fast_kernel<1>, g, b, 256 * P.stride, s, P, idx, idx_stride); break;
    case 2: hipLaunchKernelGGL(ssgpu_sort_gather_rec_fast_kernel<2>, g, b, 256 * P.stride, s, P, idx, idx_stride); break;
    case 3: hipLaunchKernelGGL(ssgpu_sort_gather_rec_fast_kernel<3>, g, b, 256 * P.stride, s, P, idx, idx_stride); break;
    case 4: hipLaunchKernelGGL(ssgpu_sort_gather_rec_fast_kernel<4>, g, b, 256 * P.stride, s, P, idx, idx_stride); break;
    case 5: hipLaunchKernelGGL(ssgpu_sort_gather_rec_fast_kernel<5>, g, b, 256 * P.stride, s, P, idx, idx_stride); break;
    case 6: hipLaunchKernelGGL(ssgpu_sort_gather_rec_fast_kernel<6>, g, b, 256 * P.stride, s, P, idx, idx_stride); break;
    case 7: hipLaunchKernelGGL(ssgpu_sort_gather_rec_fast_kernel<7>, g, b, 256 * P.stride, s, P, idx, idx_stride); break;
    case 8: hipLaunchKernelGGL(ssgpu_sort_gather_rec_fast_kernel<8>, g, b, 256 * P.stride, s, P, idx, idx_stride); break;
    default: hipLaunchKernelGGL(ssgpu_sort_gather_rec_kernel, g, b, 256 * P.stride, s, P, idx, idx_stride); break;
  }
  return hipGetLastError();
}
hipError_t ssgpu_launch_sort_fix_ties_compact(uint64_t* kc, const uint64_t* keys, uint64_t n, uint32_t* too_long, hipStream_t s) {
  if (n) hipLaunchKernelGGL(ssgpu_sort_fix_ties_compact_kernel, dim3(blocks_for(n, SORT_TIE_WG_ROWS)), dim3(256), 0, s, (u64*)kc, (const u64*)keys, (u64)n, too_long);
  return hipGetLastError();
}
hipError_t ssgpu_launch_sort_extract_idx(uint32_t* idx, const uint64_t* kc, uint64_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(ssgpu_sort_extract_idx_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, idx, (const u64*)kc, (u64)n);
  return hipGetLastError();
}
hipError_t ssgpu_launch_sort_unkey(void* out, const uint64_t* keys, uint32_t width, int kind, int descending, uint64_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(ssgpu_sort_unkey_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, out, (const u64*)keys, width, kind, descending, (u64)n);
  return hipGetLastError();
}
hipError_t ssgpu_launch_sort_gather(void* out, uint8_t* out_nulls, const void* col, const uint8_t* nulls, uint32_t width,
                                    const uint32_t* idx, uint64_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(ssgpu_sort_gather_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, out, out_nulls, col, nulls, width, idx, (u64)n);
  return hipGetLastError();
}
hipError_t ssgpu_launch_cluster_count(const void* const* data, const uint8_t* const* nulls, const uint32_t* width, uint32_t nkeys,
                                      uint64_t n, uint32_t* tile_counts, hipStream_t s) {
  ClusterKeys K; K.n = nkeys;
  for (uint32_t k = 0; k < 16; ++k) { K.data[k] = k < nkeys ? data[k] : nullptr; K.nulls[k] = k < nkeys ? nulls[k] : nullptr; K.width[k] = k < nkeys ? width[k] : 0; }
  const int nt = blocks_for(n, 512);
  if (nt) hipLaunchKernelGGL(ssgpu_cluster_count_kernel, dim3(nt), dim3(256), 0, s, K, (u64)n, tile_counts);
  return hipGetLastError();
}
hipError_t ssgpu_launch_cluster_flags(const void* const* data, const uint8_t* const* nulls, const uint32_t* width, uint32_t nkeys,
                                      uint64_t n, uint8_t* flag, hipStream_t s) {
  ClusterKeys K; K.n = nkeys;
  for (uint32_t k = 0; k < 16; ++k) { K.data[k] = k < nkeys ? data[k] : nullptr; K.nulls[k] = k < nkeys ? nulls[k] : nullptr; K.width[k] = k < nkeys ? width[k] : 0; }
  if (n) hipLaunchKernelGGL(ssgpu_cluster_flags_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, K, (u64)n, flag);
  return hipGetLastError();
}
hipError_t ssgpu_launch_cluster_assign(const void* const* data, const uint8_t* const* nulls, const uint32_t* width, uint32_t nkeys,
                                       void* const* out_data, uint8_t* const* out_nulls, uint64_t n, const uint32_t* tile_offsets,
                                       uint32_t* seg_id, hipStream_t s) {
  ClusterKeys K; K.n = nkeys; ClusterKeyOut O;
  for (uint32_t k = 0; k < 16; ++k) {
    K.data[k] = k < nkeys ? data[k] : nullptr; K.nulls[k] = k < nkeys ? nulls[k] : nullptr; K.width[k] = k < nkeys ? width[k] : 0;
    O.data[k] = k < nkeys ? out_data[k] : nullptr; O.nulls[k] = k < nkeys ? out_nulls[k] : nullptr;
  }
  const int nt = blocks_for(n, 512);
  if (nt) hipLaunchKernelGGL(ssgpu_cluster_assign_kernel, dim3(nt), dim3(256), 0, s, K, O, (u64)n, tile_offsets, seg_id);
  return hipGetLastError();
}
// ---- first-seen ranks of key clusters (DISTINCT aggregates under max_unique_keys_in_result; runtime.cpp: rank_columns) ---------------
// The rows arrive sorted by their group keys (stable: input order inside a cluster) with their input row id as a column.  A key's
// place in the reference's RowHashSet is its FIRST-SEEN order (row_hash_set.cc:458-517): the rank of its cluster's first row id
// among all clusters' first row ids.  (1) the first row id of every cluster, (2) [host: sort the clusters by it] rank of every
// cluster, (3) per row: its result row = min(rank, limit) and whether its own group keeps a row of its own (rank <= limit).
__global__ void ssgpu_seg_first_kernel(const u32* __restrict__ seg_id, const u64* __restrict__ rowid, u64 n, u32* __restrict__ first) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (i == 0 || seg_id[i] != seg_id[i - 1]) first[seg_id[i]] = (u32)rowid[i];
}
__global__ void ssgpu_rank_scatter_kernel(const u32* __restrict__ sorted, u64 n_seg, u32* __restrict__ rank) {
  const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_seg) rank[sorted[j]] = (u32)j;
}
__global__ void ssgpu_rank_rows_kernel(const u32* __restrict__ seg_id, const u32* __restrict__ rank, u64 n, u32 limit, u32* __restrict__ out_rank, u8* __restrict__ out_own) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 r = rank[seg_id[i]];
  out_rank[i] = r < limit ? r : limit;
  out_own[i] = r <= limit ? (u8)1 : (u8)0;
}
hipError_t ssgpu_launch_seg_first(const uint32_t* seg_id, const uint64_t* rowid, uint64_t n, uint32_t* first, hipStream_t s) {
  if (n) hipLaunchKernelGGL(ssgpu_seg_first_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, seg_id, (const u64*)rowid, (u64)n, first);
  return hipGetLastError();
}
hipError_t ssgpu_launch_rank_scatter(const uint32_t* sorted, uint64_t n_seg, uint32_t* rank, hipStream_t s) {
  if (n_seg) hipLaunchKernelGGL(ssgpu_rank_scatter_kernel, dim3(blocks_for(n_seg, 256)), dim3(256), 0, s, sorted, (u64)n_seg, rank);
  return hipGetLastError();
}
hipError_t ssgpu_launch_rank_rows(const uint32_t* seg_id, const uint32_t* rank, uint64_t n, uint32_t limit, uint32_t* out_rank, uint8_t* out_own, hipStream_t s) {
  if (n) hipLaunchKernelGGL(ssgpu_rank_rows_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, seg_id, rank, (u64)n, limit, out_rank, (u8*)out_own);
  return hipGetLastError();
}
hipError_t ssgpu_launch_dense_extract(const uint64_t* acc, const uint32_t* cnt, uint32_t n_gaggs, uint64_t n_rows,
                                      const GroupAggOut* outs, uint32_t n_out, hipStream_t s) {
  DenseExtractParams P; memset(&P, 0, sizeof(P));
  P.acc = (const u64*)acc; P.cnt = cnt; P.n_gaggs = n_gaggs; P.n_out = n_out; P.n_rows = n_rows;
  for (uint32_t q = 0; q < n_out; ++q) {
    P.out[q].data = outs[q].data; P.out[q].is_null = outs[q].is_null; P.out[q].s = outs[q].s;
    P.out[q].out_kind = outs[q].out_kind; P.out[q].has_cnt = outs[q].has_cnt;
  }
  if (n_rows) hipLaunchKernelGGL(ssgpu_dense_extract_kernel, dim3(blocks_for(n_rows, 256)), dim3(256), 0, s, P);
  return hipGetLastError();
}
