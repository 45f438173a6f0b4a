// exchange_kernels.hip -- result images for the one-collective multi-GPU exchange (see ssgpu.h,
// "result images").  The reference has no distributed code; the shape it documents for a sharded
// GroupAggregate is aggregate-per-shard -> shuffle -> final aggregate (cursor/core/aggregate.h:236-242).
// Here a shard's partial table travels as ONE contiguous device buffer whose header carries the row
// count, so no device value is read on the host between the shard run and the merge run.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "launch.h"

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned char u8;

// bytes [0, n) of src -> dst, 16 bytes per lane when both sides allow it
__device__ __forceinline__ void copy_bytes(char* __restrict__ dst, const char* __restrict__ src, u64 n, u64 first, u64 stride) {
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0) {
    const u64 n16 = n / 16;
    for (u64 i = first; i < n16; i += stride) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (u64 i = n16 * 16 + first; i < n; i += stride) dst[i] = src[i];
  } else {
    for (u64 i = first; i < n; i += stride) dst[i] = src[i];
  }
}

// grid = (blocks per piece, pieces): piece p of the result -> its slot of the image; block (0, 0) writes the header
__global__ __launch_bounds__(256) void ssgpu_pack_image_kernel(const ImagePackParams P) {
  const u64 rows_have = P.rows_dev ? *P.rows_dev : P.rows_host;
  const u64 rows = rows_have < P.capacity ? rows_have : P.capacity;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    u64* h = reinterpret_cast<u64*>(P.image);
    u64 retry = 0;
    for (u32 f = 0; f < P.n_retry; ++f) retry |= (u64)*P.retry_flags[f];
    h[0] = rows; h[1] = P.capacity; h[2] = (rows_have > P.capacity || retry) ? 1ull : 0ull; h[3] = rows_have;
    u64 err = 0;
    for (u32 f = 0; f < P.n_flags; ++f) err |= (u64)(*P.error_flags[f] & 0xFFu);   // evaluation errors only: the NaN-in-MIN/MAX bit is no error (across shards NaNs are skipped, ssgpu.h)
    h[4] = err; h[5] = 0; h[6] = 0; h[7] = 0;
  }
  const ImagePiece pc = P.pieces[blockIdx.y];
  char* dst = reinterpret_cast<char*>(P.image) + pc.image_off;
  const u64 n = rows * pc.width;
  const u64 first = (u64)blockIdx.x * 256 + threadIdx.x, stride = (u64)gridDim.x * 256;
  if (pc.src) copy_bytes(dst, reinterpret_cast<const char*>(pc.src), n, first, stride);
  else for (u64 i = first; i < n; i += stride) dst[i] = 0;     // nullable attribute without a NULL vector: no NULLs
}

// grid = (blocks, pieces + 1, images): piece p of image r -> rows [r * capacity, ...) of the unpacked column;
// the extra piece is the validity column; block (0, 0, 0) folds the headers into the trailer
__global__ __launch_bounds__(256) void ssgpu_unpack_images_kernel(const ImageUnpackParams P) {
  const u32 r = blockIdx.z;
  const char* image = reinterpret_cast<const char*>(P.images) + (u64)r * P.image_bytes;
  const u64* h = reinterpret_cast<const u64*>(image);
  const u64 rows = h[0] < P.capacity ? h[0] : P.capacity;
  const u64 first = (u64)blockIdx.x * 256 + threadIdx.x, stride = (u64)gridDim.x * 256;
  if (blockIdx.y == P.n_pieces) {
    u8* valid = reinterpret_cast<u8*>(P.unpacked) + P.valid_off + (u64)r * P.capacity;
    for (u64 i = first; i < P.capacity; i += stride) valid[i] = i < rows ? (u8)1 : (u8)0;
    if (r == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
      u64 mx = 0, sum = 0, over = 0, err = 0, failed = 0;
      for (u32 q = 0; q < P.n_images; ++q) {
        const u64* hq = reinterpret_cast<const u64*>(reinterpret_cast<const char*>(P.images) + (u64)q * P.image_bytes);
        mx = hq[3] > mx ? hq[3] : mx; sum += hq[0]; over |= hq[2]; err |= hq[4];
        failed = hq[5] > failed ? hq[5] : failed;     // a rank whose shard run FAILED (quota, interrupt ...) still sends an (empty) image: its return code
      }
      u64* t = reinterpret_cast<u64*>(reinterpret_cast<char*>(P.unpacked) + P.trailer_off);
      t[0] = mx; t[1] = sum; t[2] = over; t[3] = (err & 0xFFull) | (failed << 8);
    }
    return;
  }
  const ImagePiece pc = P.pieces[blockIdx.y];
  char* dst = reinterpret_cast<char*>(P.unpacked) + pc.unpacked_off + (u64)r * P.capacity * pc.width;
  copy_bytes(dst, image + pc.image_off, rows * pc.width, first, stride);
}

// ---- key-range exchange ------------------------------------------------------------------------------------------------
// The all-gather form sends every partial table to every rank, which then merges ALL of them (world x redundant work, and
// (world - 1) x the bytes on every link).  Here a shard's partial table is split by key instead: row -> image
// hash(group keys) mod n_dest, one image per destination rank, ONE all-to-all of equally sized images, and every rank
// merges only the groups it owns (1 / world of the key space).  The hash is a function of the key BYTES (NULL keys hash
// as a flag, their value bytes ignored), so every rank routes a key to the same owner.  Rows of one source keep no
// particular order inside an image (a group occurs once per source table: the merge does not depend on it).
// Phase 1 (this kernel): destination and position of every row (one returning atomic per wave and destination on the n_dest counters);
// phase 2 (ssgpu_route_copy_kernel): the cells, column by column, coalesced reads.
__device__ __forceinline__ u64 route_mix(u64 h, u64 x) { h ^= x; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29; return h * 0xBF58476D1CE4E5B9ull; }
__global__ __launch_bounds__(256) void ssgpu_route_rows_kernel(const ImagePackParams P, const ImageRoutePieces R, u32* __restrict__ dest_pos) {
  const u64 rows = P.rows_dev ? *P.rows_dev : P.rows_host;
  const u32 lane = threadIdx.x & 63u;
  const u64 lanes_below = (1ull << lane) - 1ull;
  for (u64 base = (u64)blockIdx.x * 256; base < rows; base += (u64)gridDim.x * 256) {
    const u64 r = base + threadIdx.x;
    const bool active = r < rows;
    u32 d = 0;
    if (active) {
      u64 h = 0x243F6A8885A308D3ull;
      for (u32 k = 0; k < R.n_keys; ++k) {
        const ImagePiece pc = P.pieces[R.key_piece[k]];
        const bool is_null = R.key_null_piece[k] >= 0 && P.pieces[R.key_null_piece[k]].src && reinterpret_cast<const u8*>(P.pieces[R.key_null_piece[k]].src)[r] != 0;
        u64 v = 0;
        if (!is_null) {
          const char* c = reinterpret_cast<const char*>(pc.src) + r * pc.width;
          v = pc.width == 8 ? *reinterpret_cast<const u64*>(c) : pc.width == 4 ? (u64)*reinterpret_cast<const u32*>(c) : (u64)*reinterpret_cast<const u8*>(c);
        }
        h = route_mix(h, v + (is_null ? 0x51ull : 0ull)) + k;
      }
      d = (u32)((h >> 32) % R.n_dest);
    }
    // one atomic per (wave, destination present in it), not one per row: with few destinations every row of the table hits
    // the same handful of counters (measured on one rank, 1e5 rows on ONE counter: 1.2 ms of a 3.7 ms step)
    u32 pos = 0;
    u64 todo = __ballot(active);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const u32 k = (u32)__shfl((int)d, leader);
      const u64 same = __ballot(active && d == k);
      u32 first = 0;
      if ((int)lane == leader) first = atomicAdd(&R.counters[k], (u32)__popcll(same));
      first = (u32)__shfl((int)first, leader);
      if (active && d == k) pos = first + (u32)__popcll(same & lanes_below);
      todo &= ~same;
    }
    if (active) { dest_pos[2 * r] = d; dest_pos[2 * r + 1] = pos; }
  }
}
// grid = (blocks, pieces): piece p of row r -> image dest[r], row pos[r]; block (0, 0) writes the n_dest headers
__global__ __launch_bounds__(256) void ssgpu_route_copy_kernel(const ImagePackParams P, const ImageRoutePieces R, const u32* __restrict__ dest_pos) {
  const u64 rows = P.rows_dev ? *P.rows_dev : P.rows_host;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < R.n_dest) {
    const u32 d = threadIdx.x;
    u64* h = reinterpret_cast<u64*>(reinterpret_cast<char*>(P.image) + (u64)d * R.image_bytes);
    const u64 have = R.counters[d];
    u64 retry = 0, err = 0;
    for (u32 f = 0; f < P.n_retry; ++f) retry |= (u64)*P.retry_flags[f];
    for (u32 f = 0; f < P.n_flags; ++f) err |= (u64)(*P.error_flags[f] & 0xFFu);   // (as in ssgpu_pack_image_kernel)
    h[0] = have < P.capacity ? have : P.capacity; h[1] = P.capacity; h[2] = (have > P.capacity || retry) ? 1ull : 0ull; h[3] = have;
    h[4] = err; h[5] = 0; h[6] = 0; h[7] = 0;
    R.counters[d] = 0u;   // nobody else reads the counters in this launch: left clear for the next routing (no per-step fill launch)
  }
  const ImagePiece pc = P.pieces[blockIdx.y];
  const u32 w = pc.width;
  for (u64 r = (u64)blockIdx.x * 256 + threadIdx.x; r < rows; r += (u64)gridDim.x * 256) {
    const u32 d = dest_pos[2 * r], pos = dest_pos[2 * r + 1];
    if (pos >= P.capacity) continue;                     // this image is full: flagged in its header, the caller regrows and repeats
    char* dst = reinterpret_cast<char*>(P.image) + (u64)d * R.image_bytes + pc.image_off + (u64)pos * w;
    if (!pc.src) { for (u32 b = 0; b < w; ++b) dst[b] = 0; continue; }
    const char* src = reinterpret_cast<const char*>(pc.src) + r * w;
    if (w == 8) *reinterpret_cast<u64*>(dst) = *reinterpret_cast<const u64*>(src);
    else if (w == 4) *reinterpret_cast<u32*>(dst) = *reinterpret_cast<const u32*>(src);
    else *dst = *src;
  }
}
hipError_t ssgpu_launch_route_images(const ImagePackParams& P, const ImageRoutePieces& R, hipStream_t s) {
  // dest_pos scratch lives behind the counters (the caller sized it: 256 + 2 * capacity_in words)
  u32* dest_pos = R.counters + 256;   // (n_dest <= 256: a fixed head, so that the counters a routing leaves clear stay clear whatever the next n_dest)
  const u64 rows_max = P.rows_dev ? P.rows_host : P.rows_host;   // (rows_host carries the upper bound of the row count when rows_dev is given)
  const unsigned bx = (unsigned)std::min<u64>(std::max<u64>((rows_max + 255) / 256, 1), 512);
  hipLaunchKernelGGL(ssgpu_route_rows_kernel, dim3(bx), dim3(256), 0, s, P, R, dest_pos);
  hipLaunchKernelGGL(ssgpu_route_copy_kernel, dim3(bx, std::max<u32>(P.n_pieces, 1)), dim3(256), 0, s, P, R, (const u32*)dest_pos);
  return hipGetLastError();
}

hipError_t ssgpu_launch_pack_image(const ImagePackParams& P, hipStream_t s) {
  const unsigned bx = (unsigned)std::min<u64>(std::max<u64>((P.capacity * 8 + 4095) / 4096, 1), 256);
  hipLaunchKernelGGL(ssgpu_pack_image_kernel, dim3(bx, std::max<u32>(P.n_pieces, 1)), dim3(256), 0, s, P);
  return hipGetLastError();
}
hipError_t ssgpu_launch_unpack_images(const ImageUnpackParams& P, hipStream_t s) {
  const unsigned bx = (unsigned)std::min<u64>(std::max<u64>((P.capacity * 8 + 4095) / 4096, 1), 128);
  hipLaunchKernelGGL(ssgpu_unpack_images_kernel, dim3(bx, P.n_pieces + 1, std::max<u32>(P.n_images, 1)), dim3(256), 0, s, P);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Dense-slot tables across ranks (DenseKeyMap / DenseFoldParams in launch.h; SURVEY 8(e): partial tables that share a
// deterministic slot function combine element by element).  After ONE all-to-all of equally shaped table chunks a rank
// holds n_chunks images of the slot range it owns; this kernel folds them, word by word with the word's merge function,
// into the table the usual extraction reads.  DOUBLE sums cross as their raw (hi, lo) accumulator pairs and are added in
// double-double, so a group's sum is rounded once, at the extraction -- whatever the number of shards (the reference is
// single-process: cursor/core/aggregate_groups.cc:332-433 keeps one accumulator per group; this is its N-way restatement).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double u2d_x(u64 v) { return __longlong_as_double((long long)v); }
__device__ __forceinline__ u64 d2u_x(double v) { return (u64)__double_as_longlong(v); }
__global__ __launch_bounds__(256) void ssgpu_dense_fold_kernel(const DenseFoldParams P) {
  const u64 n_slots = (u64)P.slots + 1ull, ng = P.n_gaggs, n_words = n_slots * ng;
  const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) {
      u32 flags = 0, err = 0, failed = 0;
      for (u32 c = 0; c < P.n_chunks; ++c) {
        const u32* h = reinterpret_cast<const u32*>(static_cast<const char*>(P.chunks) + (u64)c * P.chunk_bytes);
        flags |= h[0] & 0xFFu; err |= h[1];
        failed = (h[0] >> 8) > failed ? (h[0] >> 8) : failed;   // bits 8..: the return code of a rank whose shard run failed (bit 3 set)
      }
      P.flags_out[0] = flags | (failed << 8); P.flags_out[1] = err;   // (what THIS fold's headers carried)
    }
    for (int q = 0; q < 2; ++q) for (u32 w = threadIdx.x; w < P.n_clear[q]; w += 256u) P.clear[q][w] = 0u;
  }
  if (i >= n_words) return;
  const u64 slot = i / ng; const u32 w = (u32)(i - slot * ng);
  const u64 keys_off = SSGPU_DENSE_HEADER, acc_off = keys_off + n_slots * 8ull, cnt_off = acc_off + n_words * 8ull;
  const u32 op = P.merge_op[w];
  if (w == 0) {
    u64 key = ~0ull;   // VM_KEY_EMPTY
    for (u32 c = 0; c < P.n_chunks; ++c) {
      const u64 k = reinterpret_cast<const u64*>(static_cast<const char*>(P.chunks) + (u64)c * P.chunk_bytes + keys_off)[slot];
      if (k != ~0ull) key = k;
    }
    P.keys[slot] = key;
  }
  if (P.any_cnt) {
    u32 n = 0;
    for (u32 c = 0; c < P.n_chunks; ++c) n += reinterpret_cast<const u32*>(static_cast<const char*>(P.chunks) + (u64)c * P.chunk_bytes + cnt_off)[i];
    P.cnt[i] = n;
  }
  if (op == 3u /* VM_MERGE_ADD_F64 */ && w > 0 && P.merge_op[w - 1] == 4u /* VM_MERGE_ADD_F64_HI */) return;   // the low word of a pair: written with its high word
  const u64* a0 = reinterpret_cast<const u64*>(static_cast<const char*>(P.chunks) + acc_off);
  u64 v = a0[i];
  if (op == 4u) {
    double hi = u2d_x(v), lo = u2d_x(a0[i + 1]);
    for (u32 c = 1; c < P.n_chunks; ++c) {
      const u64* a = reinterpret_cast<const u64*>(static_cast<const char*>(P.chunks) + (u64)c * P.chunk_bytes + acc_off);
      const double bh = u2d_x(a[i]), bl = u2d_x(a[i + 1]);
      const double s = hi + bh, bb = s - hi;
      double e = (hi - (s - bb)) + (bh - bb);     // TwoSum: s + e == hi + bh exactly
      e += lo + bl;
      const double h2 = s + e;
      lo = e - (h2 - s); hi = h2;
    }
    P.acc[i] = d2u_x(hi); P.acc[i + 1] = d2u_x(lo);
    return;
  }
  for (u32 c = 1; c < P.n_chunks; ++c) {
    const u64 b = reinterpret_cast<const u64*>(static_cast<const char*>(P.chunks) + (u64)c * P.chunk_bytes + acc_off)[i];
    if (op == 0u) v += b;
    else if (op == 1u) v = b < v ? b : v;
    else if (op == 2u) v = b > v ? b : v;
    else v = d2u_x(u2d_x(v) + u2d_x(b));
  }
  P.acc[i] = v;
}
hipError_t ssgpu_launch_dense_fold(const DenseFoldParams& P, hipStream_t stream) {
  const u64 n_words = ((u64)P.slots + 1ull) * P.n_gaggs;
  const unsigned int blocks = (unsigned int)std::max<u64>((n_words + 255) / 256, 1);
  hipLaunchKernelGGL(ssgpu_dense_fold_kernel, dim3(blocks), dim3(256), 0, stream, P);
  return hipGetLastError();
}
// the header of every chunk of a freshly filled table buffer: the run's overflow / domain-miss flags and its evaluation-error word
__global__ void ssgpu_dense_headers_kernel(char* chunks, u32 n_chunks, u64 chunk_bytes, const u32* __restrict__ overflow4, const u32* __restrict__ error_flag, u32 forced) {
  const u32 flags = forced ? forced : (overflow4[0] ? 1u : 0u) | (overflow4[1] ? 2u : 0u) | (overflow4[3] ? 4u : 0u);
  const u32 err = !forced && error_flag ? *error_flag : 0u;
  for (u32 c = threadIdx.x; c < n_chunks; c += blockDim.x) {
    u32* h = reinterpret_cast<u32*>(chunks + (u64)c * chunk_bytes);
    h[0] = flags; h[1] = err;
  }
}
hipError_t ssgpu_launch_dense_headers(void* chunks, unsigned int n_chunks, unsigned long long chunk_bytes, const unsigned int* overflow4, const unsigned int* error_flag, hipStream_t stream, unsigned int forced_flags) {
  hipLaunchKernelGGL(ssgpu_dense_headers_kernel, dim3(1), dim3(64), 0, stream, static_cast<char*>(chunks), n_chunks, (u64)chunk_bytes, overflow4, error_flag, forced_flags);
  return hipGetLastError();
}
