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
    for (u32 f = 0; f < P.n_flags; ++f) err |= (u64)*P.error_flags[f];
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
      u64 mx = 0, sum = 0, over = 0, err = 0;
      for (u32 q = 0; q < P.n_images; ++q) {
        const u64* hq = reinterpret_cast<const u64*>(reinterpret_cast<const char*>(P.images) + (u64)q * P.image_bytes);
        mx = hq[3] > mx ? hq[3] : mx; sum += hq[0]; over |= hq[2]; err |= hq[4];
      }
      u64* t = reinterpret_cast<u64*>(reinterpret_cast<char*>(P.unpacked) + P.trailer_off);
      t[0] = mx; t[1] = sum; t[2] = over; t[3] = err;
    }
    return;
  }
  const ImagePiece pc = P.pieces[blockIdx.y];
  char* dst = reinterpret_cast<char*>(P.unpacked) + pc.unpacked_off + (u64)r * P.capacity * pc.width;
  copy_bytes(dst, image + pc.image_off, rows * pc.width, first, stride);
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
