// pipeline_kernels.hip -- hand-written gfx950 kernels for the fused
// Scan -> Compute/Project -> Filter -> {ScalarAggregate | GroupAggregate |
// materialise} pipeline.  See vm.h for the execution model.
//
// Reference loops restated here (paths relative to the reference tree):
//   K1/K2 VectorBinaryPrimitive / VectorUnaryPrimitive
//         supersonic/expression/vector/vector_primitives.h:99-105,393-412
//   K3    null propagation  expression/core/projecting_bound_expressions.cc:66-82,
//         vector_logic.h:30-50, elementary_bound_expressions.cc:343-404
//   K4    FilterCursor::PrepareInputRowIds  cursor/core/filter.cc:170-199
//   K5    DataCopier<...SELECTION>  base/infrastructure/copy_column.cc:199-217
//   K6    ColumnAggregatorImpl::UpdateAggregation  cursor/core/column_aggregator.cc:108-124
//         + AggregationOperator  base/infrastructure/aggregation_operators.h:173-228
//   K7    RowHashSetImpl::InsertUnique  cursor/infrastructure/row_hash_set.cc:458-517
//
// Compile with -ffp-contract=off: the reference's IEEE + - * / are correctly
// rounded single operations; FMA contraction would change results.
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif
#include "vm.h"
#include "launch.h"
#ifdef SSGPU_RTC_NINSTR
#include "rtc_prog.h"   // static constexpr unsigned int kRtcProg[SSGPU_RTC_NINSTR + 1][8]: this plan's program (rtc.cpp)
#endif

typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int u32;
typedef int i32;
typedef unsigned char u8;

#ifdef SSGPU_PC_PROFILE   // per-instruction cycle profile: development builds only (make PCPROF=1)
#define PC_PROF(...) __VA_ARGS__
#else
#define PC_PROF(...)
#endif
// The kernel's LDS is all dynamic and starts at LDS address 0 (no static __shared__ in this
// kernel), so VM register offsets ARE LDS addresses: `smem + off` is a plain integer-to-LDS-
// pointer cast, with no per-use "add the symbol's base" scalar instruction.
struct LdsBase {
  __device__ __forceinline__ char* operator+(u32 off) const {
    return (char*)(__attribute__((address_space(3))) char*)off;
  }
};
static constexpr LdsBase smem{};

template <typename T> struct Vec2 { typedef T type __attribute__((ext_vector_type(2))); };

template <typename T>
__device__ __forceinline__ typename Vec2<T>::type lds_load2(u32 off, int p) {
  return *reinterpret_cast<const typename Vec2<T>::type*>(smem + off + (u32)p * (2u * sizeof(T)));
}
template <typename T>
__device__ __forceinline__ void lds_store2(u32 off, int p, T x, T y) {
  typename Vec2<T>::type v; v.x = x; v.y = y;
  *reinterpret_cast<typename Vec2<T>::type*>(smem + off + (u32)p * (2u * sizeof(T))) = v;
}
template <typename T> __device__ __forceinline__ T imm_as(u64 imm) {
  T v; __builtin_memcpy(&v, &imm, sizeof(T)); return v;
}
// Operand fetch: register pair from LDS, or the broadcast immediate.
// Immediates live in an LDS constant pool (one 16-byte replicated entry per instruction,
// written once per kernel): an immediate operand is the same pair load with stride 0, so
// handlers are branch-free and their K loads issue back to back.
template <typename T>
__device__ __forceinline__ typename Vec2<T>::type fetch2(u32 off, u32 mask, int p) {
  // (pair offset) & mask: a row register has mask ~0, the immediate mask 0 (stride-0 read)
  return *reinterpret_cast<const typename Vec2<T>::type*>(smem + off + (((u32)p * (u32)(2u * sizeof(T))) & mask));
}

// ---------------------------------------------------------------------------
// wave-level reductions: DPP inside each 16-lane row, v_readlane across rows.
// Result is wave-uniform.  EXEC must be full (handlers run in uniform flow).
// ---------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ u32 dpp32(u32 v) {
  return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL> __device__ __forceinline__ u64 dpp64(u64 v) {
  u32 lo = dpp32<CTRL>((u32)v), hi = dpp32<CTRL>((u32)(v >> 32));
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 readlane64(u64 v, int lane) {
  u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, lane);
  u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), lane);
  return ((u64)hi << 32) | lo;
}
#define DPP_QUAD_XOR1 0xB1  /* quad_perm:[1,0,3,2] */
#define DPP_QUAD_XOR2 0x4E  /* quad_perm:[2,3,0,1] */
#define DPP_ROW_ROR4 0x124
#define DPP_ROW_ROR8 0x128

template <typename Op> __device__ __forceinline__ u64 wave_reduce_u64(u64 v, Op op) {
  v = op(v, dpp64<DPP_QUAD_XOR1>(v));
  v = op(v, dpp64<DPP_QUAD_XOR2>(v));
  v = op(v, dpp64<DPP_ROW_ROR4>(v));
  v = op(v, dpp64<DPP_ROW_ROR8>(v));
  u64 r0 = readlane64(v, 0), r1 = readlane64(v, 16), r2 = readlane64(v, 32), r3 = readlane64(v, 48);
  return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ double u2d(u64 v) { return __longlong_as_double((i64)v); }
__device__ __forceinline__ u64 d2u(double v) { return (u64)__double_as_longlong(v); }

// double-double (hi, lo) helpers; every operation is a plain IEEE add/sub so
// the error terms are exact (Knuth TwoSum).  No FMA, no reassociation.
struct DD { double hi, lo; };
__device__ __forceinline__ DD dd_add_d(DD a, double v) {
  double t = a.hi + v;
  double bp = t - a.hi;
  double err = (a.hi - (t - bp)) + (v - bp);
  DD r; r.hi = t; r.lo = a.lo + err; return r;
}
__device__ __forceinline__ DD dd_add(DD a, DD b) {
  DD r = dd_add_d(a, b.hi);
  r.lo = r.lo + b.lo;
  return r;
}
template <int CTRL> __device__ __forceinline__ DD dd_dpp(DD v) {
  DD r; r.hi = u2d(dpp64<CTRL>(d2u(v.hi))); r.lo = u2d(dpp64<CTRL>(d2u(v.lo))); return r;
}
__device__ __forceinline__ DD dd_readlane(DD v, int lane) {
  DD r; r.hi = u2d(readlane64(d2u(v.hi), lane)); r.lo = u2d(readlane64(d2u(v.lo), lane)); return r;
}
__device__ __forceinline__ DD wave_reduce_dd(DD v) {
  v = dd_add(v, dd_dpp<DPP_QUAD_XOR1>(v));
  v = dd_add(v, dd_dpp<DPP_QUAD_XOR2>(v));
  v = dd_add(v, dd_dpp<DPP_ROW_ROR4>(v));
  v = dd_add(v, dd_dpp<DPP_ROW_ROR8>(v));
  DD r0 = dd_readlane(v, 0), r1 = dd_readlane(v, 16), r2 = dd_readlane(v, 32), r3 = dd_readlane(v, 48);
  return dd_add(dd_add(r0, r1), dd_add(r2, r3));
}

// order-preserving maps so that every MIN/MAX runs on u64 keys
__device__ __forceinline__ u64 key_i64(i64 v) { return (u64)v ^ 0x8000000000000000ull; }
__device__ __forceinline__ i64 unkey_i64(u64 k) { return (i64)(k ^ 0x8000000000000000ull); }

// compensated atomic add: acc[0] += v with the add's exact rounding error accumulated in acc[1]
__device__ __forceinline__ void dd_atomic_add(double* acc, double v) {
  const double old = unsafeAtomicAdd(acc, v);
  const double t = old + v;
  const double bp = t - old;
  const double err = (old - (t - bp)) + (v - bp);
  if (err != 0.0) unsafeAtomicAdd(acc + 1, err);
}

// ---------------------------------------------------------------------------
// row validity for sinks: row exists, passes the selection, value not NULL.
// ---------------------------------------------------------------------------
struct Valid2 { bool x, y; };
// Absent selection / NULL masks read constant all-ones / all-zeros LDS arrays of one tile
// (2 x tile_rows bytes at const_off) exactly like real masks, so a sink's three LDS reads
// (selection, NULL mask, value) need no special case and issue back to back; both rows'
// mask bytes come in ONE 16-bit read each.
__device__ __forceinline__ Valid2 valid_pair_c(int p, u32 tile_valid, u32 null_off, u32 sel_off, u32 const_off, u32 tile_rows) {
  const u32 r0 = 2u * (u32)p;
  const u32 so = sel_off == VM_NONE ? const_off : sel_off;
  const u32 no = null_off == VM_NONE ? const_off + tile_rows : null_off;
  const u32 s = *reinterpret_cast<const unsigned short*>(smem + so + r0);
  const u32 z = *reinterpret_cast<const unsigned short*>(smem + no + r0);
  Valid2 v;
  v.x = (r0 < tile_valid) && (s & 0xFFu) && !(z & 0xFFu);
  v.y = ((r0 + 1u) < tile_valid) && (s >> 8) && !(z >> 8);
  return v;
}
#define valid_pair_(p, tile_valid, null_off, sel_off) valid_pair_c(p, tile_valid, null_off, sel_off, P.const_lds_off, (u32)(VM_TILE_UNIT * K))

__device__ __forceinline__ VmAccRec* acc_rec(const VmParams& P, u32 slot, int wave) {
  return reinterpret_cast<VmAccRec*>(smem + P.acc_lds_off + slot * VM_ACC_STRIDE + wave * 32);
}

// ---------------------------------------------------------------------------
// group table helpers (open addressing, linear probing, 64-bit packed keys)
// ---------------------------------------------------------------------------
__device__ __forceinline__ u32 hash64(u64 k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return (u32)k;
}
__device__ __forceinline__ u32 group_insert(const VmGroupTable& G, u64 key) {
  const u32 mask = G.capacity_mask;
  if (key == VM_KEY_EMPTY) {           // the one key equal to the sentinel owns slot `capacity`
    __hip_atomic_store(&G.keys[mask + 1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // != EMPTY: present
    return mask + 1;
  }
  u32 slot = hash64(key) & mask;
  for (u32 probe = 0; probe <= mask; ++probe) {
    u64 cur = __hip_atomic_load(&G.keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == key) return slot;
    if (cur == VM_KEY_EMPTY) {
      u64 old = atomicCAS(&G.keys[slot], VM_KEY_EMPTY, key);
      if (old == VM_KEY_EMPTY || old == key) return slot;
    }
    slot = (slot + 1) & mask;
    if (probe >= 4096) break;          // table (nearly) full: host regrows and reruns
  }
  atomicExch(G.overflow, 1u);
  return 0xFFFFFFFFu;
}

// Workgroup-private table in LDS: the same open addressing with a short probe limit.  Returns
// VM_SLOT_LOCAL | index, or 0xFFFFFFFF when the key has no place here (table full around its
// home slot, or the EMPTY-valued key): that row then goes straight to the global table.
// The table is split into local_sub independent sub-tables (local_sub_capacity entries each,
// home slot = mulhi(hash, capacity)) and a lane uses sub-table (lane % local_sub): with few
// groups, rows of one group in different lanes then hit different LDS words instead of
// serialising on one.
__device__ __forceinline__ u32 hash_local(u64 key) {   // two 32-bit multiplies (the 64-bit mixer costs ten)
  u32 h = (u32)key ^ ((u32)(key >> 32) * 0x9E3779B1u);
  h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13;
  return h;
}
// hash partition of a key: decorrelated from the in-table home slot (both use the HIGH bits of
// their hash through mulhi)
// Segment (partition p, scatter workgroup wg) of the record buffer: partition-major, so that phase 2 reads one
// contiguous region per partition.  (Workgroup-major -- the NP segments a scatter workgroup appends to adjacent, a few
// MB -- made the scatter 6 % faster at 512 partitions but phase 2 read 768 regions 11.5 MB apart per partition and the
// 1024-partition case ran 28 ms instead of 5.6 ms; A/B on one box.)
#define SEG_INDEX(p, wg, NP, G) ((p) * (G) + (wg))
__device__ __forceinline__ u32 part_of(u64 key, u32 n_parts) {
  u32 h = hash_local(key) * 0x2C1B3C6Du; h ^= h >> 16;
  return __umulhi(h * 0x297A2D39u, n_parts);
}
__device__ __forceinline__ u32 group_probe_local(u64* keys, u32 cap, u32 tag, u64 key, u32 i, u64 cur) {
  // `cur` = keys[i] already read by the caller (the common case is a hit on the home slot)
  for (int probe = 0; probe < 16; ++probe) {
    if (cur == key) return tag + i;
    if (cur == VM_KEY_EMPTY) {
      const u64 old = atomicCAS(&keys[i], VM_KEY_EMPTY, key);
      if (old == VM_KEY_EMPTY || old == key) return tag + i;
    }
    i = i + 1u == cap ? 0u : i + 1u;
    cur = __hip_atomic_load(&keys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  return 0xFFFFFFFFu;
}

// ---------------------------------------------------------------------------
// input staging: global -> registers -> LDS.
// Every compute lane owns the row pairs p = k*256 + t of a tile, so one "unit" (staged
// column s, sub-tile k) is 2 rows = 16 / 8 / 2 bytes per lane: a fully coalesced 1 KiB
// (512 B, 128 B) wave load.  The units of tile i+1 are loaded into a register file
// (VM_PF_UNITS x 16 B per lane) right after tile i has been committed to LDS, and stay in
// flight while the program runs over tile i: HBM requests are outstanding continuously
// without spending LDS on a second input buffer (in-flight bytes live in VGPRs, 8 KiB per
// wave), and no lane ever reads another lane's rows, so staging needs no barrier at all.
// Loads are non-temporal: every input byte is streamed exactly once.
// ---------------------------------------------------------------------------
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));

// Only FULL tiles go through the register file; the (single) partial tile at the end of the
// input is staged in place by stage_partial_tile().
// staged column s: element width and LDS offset of its register -- compile-time constants in a specialised build
#if defined(SSGPU_RTC_NINSTR) && !defined(SSGPU_RTC_DYNAMIC_STAGING)
#define STAGED_WIDTH(P, s) kRtcStagedWidth[s]
#define STAGED_LDS_OFF(P, s) kRtcStagedOff[s]
#define STAGED_COUNT(P) SSGPU_RTC_NSTAGED
#else
#define STAGED_WIDTH(P, s) (P).staged[s].width
#define STAGED_LDS_OFF(P, s) (P).staged[s].lds_off
#define STAGED_COUNT(P) (P).n_staged
#endif
template <int K>
__device__ __forceinline__ void load_unit(const VmParams& P, int u, i64 tile_base, int t, u32x4& v) {
  const int s = u / K, k = u % K;
  const char* const src0 = reinterpret_cast<const char*>(P.staged[s].src);
  const u32 w = STAGED_WIDTH(P, s);
  if (src0 == nullptr) return;  // nullable attribute whose View carries no is_null vector: stays 0
  const u32 p = (u32)(k * VM_COMPUTE_THREADS + t);
  if (w == 8) {
    v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src0 + (tile_base << 3) + (size_t)(p * 16u)));
  } else if (w == 4) {
    const u32x2 x = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(src0 + (tile_base << 2) + (size_t)(p * 8u)));
    v[0] = x[0]; v[1] = x[1];
  } else if ((reinterpret_cast<uintptr_t>(src0) & 1) == 0) {
    v[0] = (u32)__builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(src0 + tile_base + (size_t)(p * 2u)));
  } else {
    const u8* b = reinterpret_cast<const u8*>(src0 + tile_base + (size_t)(p * 2u));
    v[0] = (u32)b[0] | ((u32)b[1] << 8);
  }
}

template <int K>
__device__ __forceinline__ void commit_unit(const VmParams& P, int u, u32x4 v, int t) {
  const int s = u / K, k = u % K;
  const u32 w = STAGED_WIDTH(P, s);
  const u32 p = (u32)(k * VM_COMPUTE_THREADS + t);
  char* const dst = smem + STAGED_LDS_OFF(P, s);
  if (w == 8) *reinterpret_cast<u32x4*>(dst + p * 16u) = v;
  else if (w == 4) { u32x2 x = {v[0], v[1]}; *reinterpret_cast<u32x2*>(dst + p * 8u) = x; }
  else *reinterpret_cast<unsigned short*>(dst + p * 2u) = (unsigned short)v[0];
}

// In-place staging (global -> LDS, latency exposed): the partial last tile, and the units of
// wide schemas that do not fit the register file.  Rows past the end of the input read as 0.
template <int K>
__device__ __forceinline__ void stage_units_direct(const VmParams& P, int u_begin, int u_end, i64 tile_base, u32 tile_valid, int t) {
  for (int u = u_begin; u < u_end; ++u) {
    const int s = u / K, k = u % K;
    const char* const src0 = reinterpret_cast<const char*>(P.staged[s].src);
    const u32 w = STAGED_WIDTH(P, s);
    const u32 p = (u32)(k * VM_COMPUTE_THREADS + t), r0 = 2u * p;
    char* const dst = smem + STAGED_LDS_OFF(P, s);
    const bool v0 = src0 != nullptr && r0 < tile_valid, v1 = src0 != nullptr && r0 + 1u < tile_valid;
    if (w == 8) {
      const u64* src = reinterpret_cast<const u64*>(src0) + tile_base + r0;
      u64 a = v0 ? src[0] : 0ull, b = v1 ? src[1] : 0ull;
      reinterpret_cast<u64*>(dst)[r0] = a; reinterpret_cast<u64*>(dst)[r0 + 1u] = b;
    } else if (w == 4) {
      const u32* src = reinterpret_cast<const u32*>(src0) + tile_base + r0;
      u32 a = v0 ? src[0] : 0u, b = v1 ? src[1] : 0u;
      reinterpret_cast<u32*>(dst)[r0] = a; reinterpret_cast<u32*>(dst)[r0 + 1u] = b;
    } else {
      const u8* src = reinterpret_cast<const u8*>(src0) + tile_base + r0;
      u8 a = v0 ? src[0] : (u8)0, b = v1 ? src[1] : (u8)0;
      reinterpret_cast<u8*>(dst)[r0] = a; reinterpret_cast<u8*>(dst)[r0 + 1u] = b;
    }
  }
}

// ---------------------------------------------------------------------------
// handler generators
// ---------------------------------------------------------------------------
// Keeps LLVM's speculative-execution / hoisting passes from lifting every case's
// LDS loads above the switch (which costs >250 VGPRs and all the occupancy).
#define CASE_FENCE asm volatile("" ::: "memory")
// Workgroup barrier used INSIDE handlers (cross-wave scratch of the selection sinks).
#define WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define FOR_PAIRS for (int k = 0, p = tp; k < K; ++k, p += VM_COMPUTE_THREADS)

#define BINOP(OPNAME, TA, TB, TD, EXPR)                                        \
  case VM_##OPNAME: { CASE_FENCE;                                              \
    _Pragma("unroll") FOR_PAIRS {                                              \
      auto va = fetch2<TA>(I.a, I.a_mask, p);                            \
      auto vb = fetch2<TB>(I.b, I.b_mask, p);                            \
      TD r0, r1;                                                               \
      { TA a = va.x; TB b = vb.x; r0 = (TD)(EXPR); }                           \
      { TA a = va.y; TB b = vb.y; r1 = (TD)(EXPR); }                           \
      lds_store2<TD>(I.dst, p, r0, r1);                                        \
    }                                                                          \
  } break;

#define UNOP(OPNAME, TA, TD, EXPR)                                             \
  case VM_##OPNAME: { CASE_FENCE;                                              \
    _Pragma("unroll") FOR_PAIRS {                                              \
      auto va = fetch2<TA>(I.a, I.a_mask, p);                            \
      TD r0, r1;                                                               \
      { TA a = va.x; r0 = (TD)(EXPR); }                                        \
      { TA a = va.y; r1 = (TD)(EXPR); }                                        \
      lds_store2<TD>(I.dst, p, r0, r1);                                        \
    }                                                                          \
  } break;

// Scalar aggregates: the first VM_FAST_SLOTS slots keep PER-LANE accumulators in registers
// for the whole kernel (vectors F0/F1/FC indexed by the wave-uniform slot -> v_movrel, no
// switch): per tile an aggregate costs a select-combine per row and no cross-lane traffic;
// the wave reduction happens once, at the end of the kernel.  Further slots fall back to a
// per-tile wave reduction into LDS records.

// The scalar aggregate sinks of vm_body.inc, one text per handler SHAPE (instantiated there per value type).
// AGG_U64: values that combine in a 64-bit unsigned domain -- integers as order-preserving keys, BOOL as 0 / 1.
//   CONV: the row's value `e` (of type T) as u64; IDENT: the identity of COMBINE; COMBINE: (x, y) -> u64.
#define AGG_U64_STEP(ACC, E, M, T, CONV, IDENT, COMBINE)                       \
        { T e = (E); u64 x = ACC, y = (M) ? (u64)(CONV) : (u64)(IDENT); ACC = (COMBINE); }
#define AGG_U64(OPNAME, T, CONV, IDENT, COMBINE)                               \
  case VM_##OPNAME: { CASE_FENCE;                                              \
    if (I.dst < VM_FAST_SLOTS) {                                               \
      u64 acc = F0[I.dst]; u32 ac = FC[I.dst];                                 \
      _Pragma("unroll") FOR_PAIRS {                                            \
        auto vv = lds_load2<T>(I.a, p);                                        \
        Valid2 m = valid_pair_(p, tile_valid, I.b, I.c);                       \
        AGG_U64_STEP(acc, vv.x, m.x, T, CONV, IDENT, COMBINE)                  \
        AGG_U64_STEP(acc, vv.y, m.y, T, CONV, IDENT, COMBINE)                  \
        ac += (u32)m.x + (u32)m.y;                                             \
      }                                                                        \
      F0[I.dst] = acc; FC[I.dst] = ac;                                         \
    } else {                                                                   \
      u64 local = (IDENT); u32 cnt = 0;                                        \
      _Pragma("unroll") FOR_PAIRS {                                            \
        auto vv = lds_load2<T>(I.a, p);                                        \
        Valid2 m = valid_pair_(p, tile_valid, I.b, I.c);                       \
        AGG_U64_STEP(local, vv.x, m.x, T, CONV, IDENT, COMBINE)                \
        AGG_U64_STEP(local, vv.y, m.y, T, CONV, IDENT, COMBINE)                \
        cnt += (u32)__popcll(__ballot(m.x)) + (u32)__popcll(__ballot(m.y));    \
      }                                                                        \
      u64 tot = wave_reduce_u64(local, [](u64 x, u64 y) { return (u64)(COMBINE); }); \
      if (lane == 0 && cnt) {                                                  \
        VmAccRec* A = acc_rec(P, I.dst, wave);                                 \
        u64 x = A->cnt ? A->v0 : (u64)(IDENT), y = tot;                        \
        A->v0 = (COMBINE); A->cnt += cnt;                                      \
      }                                                                        \
    }                                                                          \
  } break;

// floating MIN / MAX in the double domain.  BETTER: "y replaces x".  A NaN never replaces anything (every comparison with
// it is false); that one was met is remembered per lane for the whole kernel (vm_nan_seen: a lane mask in two SGPRs) and
// reported once, at the end of the kernel -- no branch and no atomic per tile.
#define AGG_FMM_STEP(ACC, E, M, BETTER)                                        \
        { double x = ACC, y = (double)(E); vm_nan_seen |= (M) && (y != y); if ((M) && (BETTER)) ACC = y; }
#define AGG_FMINMAX(OPNAME, T, IDENT, BETTER)                                  \
  case VM_##OPNAME: { CASE_FENCE;                                              \
    if (I.dst < VM_FAST_SLOTS) {                                               \
      double acc = u2d(F0[I.dst]); u32 ac = FC[I.dst];                         \
      _Pragma("unroll") FOR_PAIRS {                                            \
        auto vv = lds_load2<T>(I.a, p);                                        \
        Valid2 m = valid_pair_(p, tile_valid, I.b, I.c);                       \
        AGG_FMM_STEP(acc, vv.x, m.x, BETTER)                                   \
        AGG_FMM_STEP(acc, vv.y, m.y, BETTER)                                   \
        ac += (u32)m.x + (u32)m.y;                                             \
      }                                                                        \
      F0[I.dst] = d2u(acc); FC[I.dst] = ac;                                    \
    } else {                                                                   \
      double local = (IDENT); u32 cnt = 0;                                     \
      _Pragma("unroll") FOR_PAIRS {                                            \
        auto vv = lds_load2<T>(I.a, p);                                        \
        Valid2 m = valid_pair_(p, tile_valid, I.b, I.c);                       \
        AGG_FMM_STEP(local, vv.x, m.x, BETTER)                                 \
        AGG_FMM_STEP(local, vv.y, m.y, BETTER)                                 \
        cnt += (u32)__popcll(__ballot(m.x)) + (u32)__popcll(__ballot(m.y));    \
      }                                                                        \
      u64 tot = wave_reduce_u64(d2u(local), [](u64 xa, u64 ya) { double x = u2d(xa), y = u2d(ya); return (BETTER) ? ya : xa; }); \
      if (lane == 0 && cnt) {                                                  \
        VmAccRec* A = acc_rec(P, I.dst, wave);                                 \
        double x = A->cnt ? u2d(A->v0) : (double)(IDENT), y = u2d(tot);        \
        A->v0 = d2u((BETTER) ? y : x); A->cnt += cnt;                          \
      }                                                                        \
    }                                                                          \
  } break;

// FIRST / LAST: (value, global row id) pairs; BETTER: "row id y replaces row id x"; IDENT: the row id that never wins.
#define AGG_FL_STEP(ROW, VAL, R, E, M, BETTER)                                 \
        { u64 x = ROW, y = (R); if ((M) && (BETTER)) { ROW = y; VAL = (u64)(E); } }
#define AGG_FIRSTLAST(OPNAME, T, IDENT, BETTER)                                \
  case VM_##OPNAME: { CASE_FENCE;                                              \
    if (I.dst < VM_FAST_SLOTS) {                                               \
      u64 av = F0[I.dst], ar = F1[I.dst]; u32 ac = FC[I.dst];                  \
      _Pragma("unroll") FOR_PAIRS {                                            \
        auto vv = lds_load2<T>(I.a, p);                                        \
        Valid2 m = valid_pair_(p, tile_valid, I.b, I.c);                       \
        u64 r0 = (u64)(P.row_id_base + tile_base + 2 * (i64)p), r1 = r0 + 1;  \
        if (I.d != VM_NONE) { auto oo = lds_load2<u64>(I.d, p); r0 = oo.x; r1 = oo.y; } /* the order is a column's (rows re-ordered on the way) */ \
        AGG_FL_STEP(ar, av, r0, vv.x, m.x, BETTER)                             \
        AGG_FL_STEP(ar, av, r1, vv.y, m.y, BETTER)                             \
        ac += (u32)m.x + (u32)m.y;                                             \
      }                                                                        \
      F0[I.dst] = av; F1[I.dst] = ar; FC[I.dst] = ac;                          \
    } else {                                                                   \
      u64 brow = (IDENT), bval = 0; u32 cnt = 0;                               \
      _Pragma("unroll") FOR_PAIRS {                                            \
        auto vv = lds_load2<T>(I.a, p);                                        \
        Valid2 m = valid_pair_(p, tile_valid, I.b, I.c);                       \
        u64 r0 = (u64)(P.row_id_base + tile_base + 2 * (i64)p), r1 = r0 + 1;  \
        if (I.d != VM_NONE) { auto oo = lds_load2<u64>(I.d, p); r0 = oo.x; r1 = oo.y; } \
        AGG_FL_STEP(brow, bval, r0, vv.x, m.x, BETTER)                         \
        AGG_FL_STEP(brow, bval, r1, vv.y, m.y, BETTER)                         \
        cnt += (u32)__popcll(__ballot(m.x)) + (u32)__popcll(__ballot(m.y));    \
      }                                                                        \
      u64 trow = wave_reduce_u64(brow, [](u64 x, u64 y) { return (BETTER) ? y : x; }); \
      u64 owner = __ballot(brow == trow);                                      \
      if (cnt) {                                                               \
        int src = __ffsll((long long)owner) - 1;                               \
        u64 tval = readlane64(bval, src);                                      \
        if (lane == 0) {                                                       \
          VmAccRec* A = acc_rec(P, I.dst, wave);                               \
          u64 x = A->cnt ? A->v1 : (u64)(IDENT), y = trow;                     \
          if (BETTER) { A->v1 = trow; A->v0 = tval; }                          \
          A->cnt += cnt;                                                       \
        }                                                                      \
      }                                                                        \
    }                                                                          \
  } break;

// SUM(a OP b) with the binary operator fused into the sink (lower.cpp emits these for register-resident slots only)
#define AGG_FUSED_I64(OPNAME, EXPR)                                            \
  case VM_##OPNAME: { CASE_FENCE;                                              \
    if (I.dst < VM_FAST_SLOTS) {                                               \
      u64 acc = F0[I.dst]; u32 ac = FC[I.dst];                                 \
      _Pragma("unroll") FOR_PAIRS {                                            \
        auto va = fetch2<u64>(I.a, I.a_mask, p);                               \
        auto vd = fetch2<u64>(I.d, I.b_mask, p);                               \
        Valid2 m = valid_pair_(p, tile_valid, I.b, I.c);                       \
        { u64 a = va.x, b = vd.x; acc += m.x ? (EXPR) : 0ull; }                \
        { u64 a = va.y, b = vd.y; acc += m.y ? (EXPR) : 0ull; }                \
        ac += (u32)m.x + (u32)m.y;                                             \
      }                                                                        \
      F0[I.dst] = acc; FC[I.dst] = ac;                                         \
    }                                                                          \
  } break;
#define AGG_FUSED_F64(OPNAME, EXPR)                                            \
  case VM_##OPNAME: { CASE_FENCE;                                              \
    if (I.dst < VM_FAST_SLOTS) {                                               \
      DD local; local.hi = u2d(F0[I.dst]); local.lo = u2d(F1[I.dst]); u32 ac = FC[I.dst]; \
      _Pragma("unroll") FOR_PAIRS {                                            \
        auto va = fetch2<double>(I.a, I.a_mask, p);                            \
        auto vd = fetch2<double>(I.d, I.b_mask, p);                            \
        Valid2 m = valid_pair_(p, tile_valid, I.b, I.c);                       \
        { double a = va.x, b = vd.x; if (m.x) local = dd_add_d(local, (EXPR)); } \
        { double a = va.y, b = vd.y; if (m.y) local = dd_add_d(local, (EXPR)); } \
        ac += (u32)m.x + (u32)m.y;                                             \
      }                                                                        \
      F0[I.dst] = d2u(local.hi); F1[I.dst] = d2u(local.lo); FC[I.dst] = ac;    \
    }                                                                          \
  } break;

#define STORE_OP(OPNAME, T)                                                    \
  case VM_##OPNAME: { CASE_FENCE;                                              \
    T* out = reinterpret_cast<T*>(P.outputs[I.dst].dst);                       \
    _Pragma("unroll") FOR_PAIRS {                                              \
      i64 r0 = tile_base + 2 * (i64)p;                                         \
      auto vv = fetch2<T>(I.a, I.a_mask, p);                             \
      if (r0 + 1 < vm_n_rows) {                                                \
        *reinterpret_cast<typename Vec2<T>::type*>(out + r0) = vv;             \
      } else if (r0 < vm_n_rows) {                                             \
        out[r0] = vv.x;                                                        \
      }                                                                        \
    }                                                                          \
  } break;

#ifdef SS_STOREC_NT   /* development A/B through SSGPU_RTC_FLAGS: compacted survivors written with nontemporal stores */
#define STOREC_ST(ptr, v) __builtin_nontemporal_store((v), (ptr))
#else
#define STOREC_ST(ptr, v) (*(ptr) = (v))
#endif
#define STOREC_OP(OPNAME, T)                                                   \
  case VM_##OPNAME: { CASE_FENCE;                                              \
    T* out = reinterpret_cast<T*>(P.outputs[I.dst].dst);                       \
    _Pragma("unroll") FOR_PAIRS {                                              \
      Valid2 m = valid_pair_(p, tile_valid, VM_NONE, I.c);          \
      auto vv = fetch2<T>(I.a, I.a_mask, p);                             \
      auto rk = lds_load2<u32>(I.b, p);                                        \
      if (m.x) STOREC_ST(out + rk.x, vv.x);                                    \
      if (m.y) STOREC_ST(out + rk.y, vv.y);                                    \
    }                                                                          \
  } break;

// survivors of the tile -> output rows [base, base + cnt): lane g writes output row g, wave stores start on
// 64-element boundaries of the OUTPUT column (full lines except at the tile's two ends)
#define STOREG_OP(OPNAME, T)                                                   \
  case VM_##OPNAME: { CASE_FENCE;                                              \
    T* out = reinterpret_cast<T*>(P.outputs[I.dst].dst);                       \
    const u32* sc = reinterpret_cast<const u32*>(smem + P.scratch_lds_off);    \
    const u32 base = sc[40], end = base + sc[41];                              \
    const u32* inv = reinterpret_cast<const u32*>(smem + I.b);                 \
    for (u32 g = (base & ~63u) + (u32)tp; g < end; g += VM_COMPUTE_THREADS) {  \
      if (g >= base) {                                                         \
        const u32 src = inv[g - base];                                         \
        out[g] = *reinterpret_cast<const T*>(smem + I.a + ((src * (u32)sizeof(T)) & I.a_mask)); \
      }                                                                        \
    }                                                                          \
  } break;

#define KEY_APPEND_OP(OPNAME, T, UT)                                           \
  case VM_##OPNAME: { CASE_FENCE;                                              \
    const u32 shift = (u32)(I.imm & 0xFF), bits = (u32)((I.imm >> 8) & 0xFF);  \
    const u32 nullbit = (u32)((I.imm >> 16) & 0xFF);                           \
    const u64 vmask = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);            \
    _Pragma("unroll") FOR_PAIRS {                                              \
      auto kk = lds_load2<u64>(I.dst, p);                                      \
      auto vv = lds_load2<T>(I.a, p);                                          \
      u64 a0 = ((u64)(UT)vv.x) & vmask, a1 = ((u64)(UT)vv.y) & vmask;          \
      if (I.b != VM_NONE) {                                                    \
        auto z = lds_load2<u8>(I.b, p);                                        \
        if (z.x) a0 = 1ull << (nullbit - shift);                               \
        if (z.y) a1 = 1ull << (nullbit - shift);                               \
      }                                                                        \
      lds_store2<u64>(I.dst, p, kk.x | (a0 << shift), kk.y | (a1 << shift));   \
    }                                                                          \
  } break;

// group aggregate: atomics on acc[slot * n_gaggs + s] of the workgroup's LDS table (slot ids
// tagged VM_SLOT_LOCAL) or of the global table
#define GAGG_PROLOGUE(LOADT)                                                   \
      Valid2 m = valid_pair_(p, tile_valid, I.b, VM_NONE);          \
      auto sl = lds_load2<u32>(I.c, p);                                        \
      m.x = m.x && sl.x != 0xFFFFFFFFu; m.y = m.y && sl.y != 0xFFFFFFFFu;      \
      auto vv = lds_load2<LOADT>(I.a, p);

#define GAGG_APPLY(LOADT, SL, E, ATOM)                                         \
      if ((SL) & VM_SLOT_LOCAL) {                                              \
        const u32 li = ((SL) & ~VM_SLOT_LOCAL) * P.group.local_stride + s;     \
        u64* A = reinterpret_cast<u64*>(smem + P.group.local_acc_off) + li; LOADT e = (E); ATOM; \
        if (has_cnt) atomicAdd(reinterpret_cast<u32*>(smem + P.group.local_cnt_off) + li, 1u); \
      } else {                                                                 \
        u64* A = &P.group.acc[(u64)(SL) * ng + s]; LOADT e = (E); ATOM;        \
        if (has_cnt) atomicAdd(&P.group.cnt[(u64)(SL) * ng + s], 1u);          \
      }

#define GAGG_ATOMIC(OPNAME, LOADT, ATOM)                                       \
  case VM_##OPNAME: { CASE_FENCE;                                              \
    const u32 ng = (u32)(I.imm >> 32) & 0x7FFFFFFFu, s = (u32)I.imm;           \
    const bool has_cnt = (I.imm >> 63) != 0;                                   \
    _Pragma("unroll") FOR_PAIRS {                                              \
      GAGG_PROLOGUE(LOADT)                                                     \
      if (m.x) { GAGG_APPLY(LOADT, sl.x, vv.x, ATOM) }                         \
      if (m.y) { GAGG_APPLY(LOADT, sl.y, vv.y, ATOM) }                         \
    }                                                                          \
  } break;

// Decoupled look-back of the single-pass compaction (SEL_RANK_LB): publishes tile `tile`'s survivor count, adds up the
// counts of the earlier tiles down to the first one that already knows its inclusive prefix, publishes this tile's own
// inclusive prefix and returns the exclusive one.  Called by wave 0 of the workgroup, all 64 lanes (lane j looks at the
// j-th tile back).  Kept out of line: inlined into the interpreter its spin loop changes the code the compiler makes
// for every OTHER program (the 8-column headline ran 1.58 ms instead of 1.07 ms with this loop inlined, same box).
__device__ __noinline__ u32 lookback_rows_before(unsigned long long* status, int tile, u32 run, u64 tag, unsigned int* ctrl, unsigned int* error_flag, int lane) {
  unsigned long long* const mine = status + tile;
  if (lane == 0) __hip_atomic_store(mine, (1ull << 62) | tag | (u64)run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  u32 excl = 0, spins = 0;
  for (int j0 = tile - 1; j0 >= 0;) {
    const int j = j0 - lane;
    u64 v = (2ull << 62) | tag;                      // before tile 0: an inclusive prefix of zero rows
    if (j >= 0) v = __hip_atomic_load(status + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ready = (v >> 62) != 0 && (v & (0x3FFFFFFFull << 32)) == tag;
    const u64 pending = __ballot(!ready), prefixes = __ballot(ready && (v >> 62) == 2);
    // usable: every lane up to the first inclusive prefix (or all 64) has published
    const u32 stop = prefixes ? (u32)__builtin_ctzll(prefixes) : 63u;
    const u64 need = stop >= 63u ? ~0ull : ((2ull << stop) - 1ull);
    if (pending & need) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 22)) { if (lane == 0) { atomicExch(ctrl + 3, 1u); atomicExch(error_flag, 3u); } break; }   // never expected: give up rather than hang
      continue;
    }
    u32 part = (u32)lane <= stop ? (u32)v : 0u;
    for (int o = 32; o > 0; o >>= 1) part += (u32)__shfl_xor((int)part, o);
    excl += part;
    if (prefixes) break;
    j0 -= 64;
  }
  if (lane == 0) __hip_atomic_store(mine, (2ull << 62) | tag | (u64)(excl + run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return excl;
}

// ---------------------------------------------------------------------------
// the pipeline kernel: persistent workgroups stride over tiles
// ---------------------------------------------------------------------------
// One persistent 4-wave workgroup strides over the tiles.  Per tile: commit the prefetched
// units to the LDS input registers, issue the loads of the next tile, run the program.
// libm functions of the MATH kernel variant (device libm: within a few ULP of the host's)
__device__ __forceinline__ double vm_math1(u32 fn, double a) {
  switch (fn) {
    case VM_MATH_EXP: return exp(a);     case VM_MATH_LN: return log(a);       case VM_MATH_LOG10: return log10(a);
    case VM_MATH_LOG2: return log2(a);   case VM_MATH_SIN: return sin(a);      case VM_MATH_COS: return cos(a);
    case VM_MATH_TAN: return tan(a);     case VM_MATH_ASIN: return asin(a);    case VM_MATH_ACOS: return acos(a);
    case VM_MATH_ATAN: return atan(a);   case VM_MATH_SINH: return sinh(a);    case VM_MATH_COSH: return cosh(a);
    case VM_MATH_TANH: return tanh(a);   case VM_MATH_ASINH: return asinh(a);  case VM_MATH_ACOSH: return acosh(a);
    default: return atanh(a);
  }
}

#ifndef SSGPU_RTC_PART   // (the runtime compilation of the partition-aggregation kernel, below, leaves the pipeline kernel out)
// MATH = true adds the libm handlers (MATH1_F64 / MATH2_F64).  They live in their own instantiation so that
// the kernel every other program runs is not touched by their code size and register demand.
template <int K, bool MATH>
__global__ __launch_bounds__(VM_WG_THREADS, VM_WAVES_PER_EU) void ssgpu_pipeline_kernel(const VmParams P) {
#ifdef SSGPU_RTC_STATIC_LDS
  // A module-loaded kernel cannot ask for more than 64 KiB of dynamic LDS: the specialised build of a launch that needs
  // more declares its LDS statically (rtc.cpp).  It is the kernel's only LDS object, so it starts at LDS address 0 and
  // every access below still goes through plain integer LDS addresses; the asm keeps the otherwise unreferenced array.
  __shared__ __attribute__((aligned(16))) char rtc_static_lds[SSGPU_RTC_STATIC_LDS];
  asm volatile("" ::"v"((u32)(size_t)(__attribute__((address_space(3))) char*)rtc_static_lds));
#endif
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int tile_rows = VM_TILE_UNIT * K;
  // the input's row count: a kernel argument, or -- for a stage that runs straight behind the stage that produces its input,
  // without the host reading the count in between -- a device word written by that stage (the argument is then an upper bound)
  i64 vm_n_rows = P.n_rows; int vm_n_tiles = P.n_tiles;
  if (P.n_rows_dev) {
    const u64 have = *P.n_rows_dev;
    vm_n_rows = have < (u64)P.n_rows ? (i64)have : P.n_rows;
    vm_n_tiles = (int)((vm_n_rows + tile_rows - 1) / tile_rows);
  }
  // Which tile a workgroup takes first; round `it` of the tile loop takes tile first + it * grid.  Workgroups are dealt to the 8 XCDs
  // round-robin (block b runs on XCD b % 8), so with first = b the 8 tiles around any row position belong to 8 different L2s.  A
  // stage that WRITES compacted rows (the materialising Filter's store pass) then has both halves of the output line two
  // neighbouring tiles share in two L2s, and each writes a partial line.  VM_FLAG_XCD_CHUNKS hands every XCD a contiguous eighth
  // of each round's tiles: neighbours share an L2 (and the line is merged there) except at the 7 chunk borders.
  const int vm_first_tile = ((P.flags & VM_FLAG_XCD_CHUNKS) && (gridDim.x & 7u) == 0u) ? (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const int n_my_tiles = vm_n_tiles > vm_first_tile ? (vm_n_tiles - vm_first_tile + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int n_units = STAGED_COUNT(P) * K;

  // zero this workgroup's LDS aggregate records (slow slots; fast slots are written once)
  for (u32 o = (u32)t * 8u; o < (u32)P.n_slots * VM_ACC_STRIDE; o += VM_COMPUTE_THREADS * 8u)
    *reinterpret_cast<u64*>(smem + P.acc_lds_off + o) = 0ull;
  // per-lane register accumulators of the first VM_FAST_SLOTS aggregate slots
  typedef u64 u64xS __attribute__((ext_vector_type(VM_FAST_SLOTS)));
  typedef u32 u32xS __attribute__((ext_vector_type(VM_FAST_SLOTS)));
  u64xS F0, F1; u32xS FC;
#pragma unroll
  for (int s = 0; s < VM_FAST_SLOTS; ++s) { F0[s] = P.slot_init0[s]; F1[s] = P.slot_init1[s]; FC[s] = 0; }
  bool vm_nan_seen = false;   // this lane met a NaN in a floating MIN / MAX (AGG_FMINMAX): reported once, after the tile loop
  // constant LDS arrays: tile_rows x 0x01 (absent selection) then tile_rows x 0x00 (absent NULL mask)
  for (u32 o = (u32)t * 8u; o < 2u * (u32)tile_rows; o += VM_COMPUTE_THREADS * 8u)
    *reinterpret_cast<u64*>(smem + P.const_lds_off + o) = o < (u32)tile_rows ? 0x0101010101010101ull : 0ull;

  typedef u32 u32x8 __attribute__((ext_vector_type(8)));
  typedef const __attribute__((address_space(4))) u32x8* ProgPtr;
  const ProgPtr prog0 = (ProgPtr)(P.prog);

  // immediates -> LDS constant pool (entry pc: the value replicated to 16 bytes by width)
  for (int pc = t; pc < P.n_instr; pc += VM_COMPUTE_THREADS) {
    const VmInstr J = P.prog[pc];
    const u32 w = J.imm_width;
    if (w) {
      u64 lo = J.imm;
      if (w == 4) lo = (lo & 0xFFFFFFFFull) * 0x100000001ull;
      else if (w == 1) lo = (lo & 0xFFull) * 0x0101010101010101ull;
      u64* e = reinterpret_cast<u64*>(smem + P.imm_pool_lds_off + (u32)pc * 16u);
      e[0] = lo; e[1] = lo;
    }
  }

  if (P.group.local_capacity) {  // workgroup-private group table: empty keys, identity accumulators
    const VmGroupTable& G = P.group;
    for (u32 e = (u32)t; e < G.local_capacity; e += VM_COMPUTE_THREADS)
      reinterpret_cast<u64*>(smem + G.local_keys_off)[e] = VM_KEY_EMPTY;
    for (u32 i = (u32)t; i < G.local_capacity * G.local_stride; i += VM_COMPUTE_THREADS) {
      reinterpret_cast<u64*>(smem + G.local_acc_off)[i] = G.acc_init[(i % G.local_stride) % G.n_gaggs];
      if (G.local_cnt_off != VM_NONE) reinterpret_cast<u32*>(smem + G.local_cnt_off)[i] = 0u;
    }
    if (t < 2) reinterpret_cast<u32*>(smem + P.scratch_lds_off + 128u)[t] = 0u;
  }
  if (P.part_n) {  // partition pass: rows this workgroup has written to each of its (partition, workgroup) segments
    u32* h = reinterpret_cast<u32*>(smem + P.part_lds_off);
    for (u32 i = (u32)t; i < P.part_n; i += VM_COMPUTE_THREADS) h[i] = 0u;
  }
  PC_PROF(if (P.debug_pc) for (int i = t; i <= P.n_instr; i += VM_COMPUTE_THREADS) reinterpret_cast<u64*>(smem + P.debug_pc_lds_off)[i] = 0ull;)
  __syncthreads();  // constant pool, accumulator records and group table visible to all waves

  // register prefetch file: the first VM_PF_UNITS units of the NEXT (full) tile
  u32x4 pf[VM_PF_UNITS];
#pragma unroll
  for (int u = 0; u < VM_PF_UNITS; ++u) pf[u] = u32x4{0u, 0u, 0u, 0u};
  const int n_pf = n_units < VM_PF_UNITS ? n_units : VM_PF_UNITS;
  {
    const i64 tb = (i64)vm_first_tile * tile_rows;
    if (n_my_tiles > 0 && tb + tile_rows <= vm_n_rows) {
#pragma unroll
      for (int u = 0; u < VM_PF_UNITS; ++u)
        if (u < n_pf) load_unit<K>(P, u, tb, t, pf[u]);
    }
  }

  u64 dbg_wait = 0;
  const u64 dbg_t0 = P.debug ? __builtin_amdgcn_s_memtime() : 0;
  const ProgPtr prog = prog0;
  for (int it = 0; it < n_my_tiles; ++it) {
    const int tile = vm_first_tile + it * (int)gridDim.x;
    const i64 tile_base = (i64)tile * tile_rows;
    const u32 tile_valid = (u32)((vm_n_rows - tile_base) < (i64)tile_rows ? (vm_n_rows - tile_base) : (i64)tile_rows);
    const u64 tw0 = (P.debug || P.debug_pc) ? __builtin_amdgcn_s_memtime() : 0;
    // A wave that is about to commit its tile and put the next tile's loads in flight is issued ahead of
    // waves in the middle of their program: the sooner the loads leave, the more of HBM's latency they hide
    // (1.5-3 % on the 8-column headline, A/B on the same box; a priority above 1 gains nothing more).
    __builtin_amdgcn_s_setprio(VM_STAGE_PRIO);
    if (tile_valid == (u32)tile_rows) {
      // commit the prefetched units (this is where a wave waits for HBM) ...
      int tc = t;
      asm volatile("" : "+v"(tc));
#pragma unroll
      for (int u = 0; u < VM_PF_UNITS; ++u)
        if (u < n_pf) commit_unit<K>(P, u, pf[u], tc);
      if (P.debug) dbg_wait += __builtin_amdgcn_s_memtime() - tw0;
      // ... units beyond the register file are fetched in place (wide schemas still stream
      // their first VM_PF_UNITS units ahead)
      if (n_units > VM_PF_UNITS) stage_units_direct<K>(P, VM_PF_UNITS, n_units, tile_base, tile_valid, t);
    } else {
      stage_units_direct<K>(P, 0, n_units, tile_base, tile_valid, t);
    }
    // ... and put the next tile's loads in flight before the program runs over this one
    {
      const i64 tb = tile_base + (i64)gridDim.x * tile_rows;
      if (it + 1 < n_my_tiles && tb + tile_rows <= vm_n_rows) {
        // launder the thread id: otherwise the per-unit lane offsets are hoisted out of the
        // tile loop as 64-bit VGPR pairs and spilled (scratch reloads between the loads)
        int tl = t;
        asm volatile("" : "+v"(tl));
#pragma unroll
        for (int u = 0; u < VM_PF_UNITS; ++u)
          if (u < n_pf) load_unit<K>(P, u, tb, tl, pf[u]);
      }
    }

    // The program is immutable for the launch: fetch it through the constant address
    // space so every instruction is ONE scalar s_load_dwordx8, and fetch the next
    // instruction while the current one executes.
    __builtin_amdgcn_s_setprio(0);
#ifdef SSGPU_RTC_NINSTR
    // Per-plan specialisation (runtime compilation, rtc.cpp): the program is a compile-time constant, the loop is
    // unrolled and every instruction's opcode switch, operand offsets and masks fold away -- what is left is the
    // straight-line sequence of this plan's handlers.
#define VM_PC_LOOP _Pragma("unroll") for (int pc = 0; pc < SSGPU_RTC_NINSTR; ++pc)
#define VM_FETCH_NEXT(pc) u32x8{kRtcProg[pc][0], kRtcProg[pc][1], kRtcProg[pc][2], kRtcProg[pc][3], kRtcProg[pc][4], kRtcProg[pc][5], kRtcProg[pc][6], kRtcProg[pc][7]}
    (void)prog;
    u32x8 raw_next = VM_FETCH_NEXT(0);
#else
#define VM_PC_LOOP for (int pc = 0; pc < P.n_instr; ++pc)
#define VM_FETCH_NEXT(pc) prog[pc]
    u32x8 raw_next = prog[0];
#endif
    PC_PROF(u64 dbg_pc_last = 0;
    if (P.debug_pc) { dbg_pc_last = __builtin_amdgcn_s_memtime(); if (t == 0) reinterpret_cast<u64*>(smem + P.debug_pc_lds_off)[P.n_instr] += dbg_pc_last - tw0; })
#ifdef SSGPU_RTC_NINSTR
#include "rtc_steps.h"   // { constexpr int pc = k; #include "vm_body.inc" } for k = 0 .. SSGPU_RTC_NINSTR - 1 (rtc.cpp)
#else
    VM_PC_LOOP {
#include "vm_body.inc"
    }
#endif
    PC_PROF(if (P.debug_pc && P.n_instr > 0 && t == 0) reinterpret_cast<u64*>(smem + P.debug_pc_lds_off)[P.n_instr - 1] += __builtin_amdgcn_s_memtime() - dbg_pc_last;)
  }

  if (vm_nan_seen && P.error_flag) atomicOr(P.error_flag, SSGPU_FLAG_NAN_IN_MINMAX);
  if (P.part_n && P.tile_counts) {  // partition pass: publish the fill of this workgroup's segments
    __syncthreads();
    const u32* h = reinterpret_cast<const u32*>(smem + P.part_lds_off);
    for (u32 i = (u32)t; i < P.part_n; i += VM_COMPUTE_THREADS) P.tile_counts[(u64)i * gridDim.x + blockIdx.x] = h[i] < P.part_seg_cap ? h[i] : P.part_seg_cap;
  }
  if (P.group.local_capacity) {
    // merge the workgroup's table into the global one: one atomic per (group, aggregate)
    // per workgroup instead of one per row
    const VmGroupTable& G = P.group;
    __syncthreads();
    const u32 ng = G.n_gaggs;
    u32 occupied = 0;
    for (u32 e = (u32)t; e < G.local_capacity; e += VM_COMPUTE_THREADS) {
      const u64 key = reinterpret_cast<const u64*>(smem + G.local_keys_off)[e];
      if (key == VM_KEY_EMPTY) continue;
      ++occupied;
      const u32 gs = group_insert(G, key);
      if (gs == 0xFFFFFFFFu) continue;  // global table full: the host regrows and reruns
      for (u32 s = 0; s < ng; ++s) {
        const u64 v = reinterpret_cast<const u64*>(smem + G.local_acc_off)[e * G.local_stride + s];
        u64* A = &G.acc[(u64)gs * ng + s];
        const u32 op = G.merge_op[s];
        if (op == VM_MERGE_ADD_U64) { if (v) atomicAdd(A, v); }
        else if (op == VM_MERGE_MIN_U64) atomicMin(A, v);
        else if (op == VM_MERGE_MAX_U64) atomicMax(A, v);
        else if (op == VM_MERGE_ADD_F64_HI) dd_atomic_add(reinterpret_cast<double*>(A), u2d(v));
        else unsafeAtomicAdd(reinterpret_cast<double*>(A), u2d(v));
        if (G.local_cnt_off != VM_NONE) {
          const u32 c = reinterpret_cast<const u32*>(smem + G.local_cnt_off)[e * G.local_stride + s];
          if (c) atomicAdd(&G.cnt[(u64)gs * ng + s], c);
        }
      }
    }
    u32* st = reinterpret_cast<u32*>(smem + P.scratch_lds_off + 128u);
    if (occupied) atomicAdd(&st[1], occupied);
    __syncthreads();
    if (t == 0) { if (st[0]) atomicAdd(&G.stats[0], st[0]); atomicMax(&G.stats[1], st[1]); }
  }

  PC_PROF(if (P.debug_pc && t == 0) for (int i = 0; i <= P.n_instr; ++i) atomicAdd(&P.debug_pc[i], reinterpret_cast<const u64*>(smem + P.debug_pc_lds_off)[i]);)
  if (P.debug && t == 0) {
    P.debug[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime() - dbg_t0;
    P.debug[blockIdx.x * 4 + 1] = dbg_wait;
    P.debug[blockIdx.x * 4 + 2] = (u64)n_my_tiles;
  }
  // fold the per-lane register accumulators of the fast slots: one wave reduction per slot
  // for the whole kernel, written to the wave's LDS record
#pragma unroll
  for (int s = 0; s < VM_FAST_SLOTS; ++s) {
    if (s < P.n_slots) {
      const int kind = P.slot_kind[s];
      const u64 cnt = wave_reduce_u64((u64)FC[s], [](u64 x, u64 y) { return x + y; });
      u64 v0 = 0, v1 = 0;
      switch (kind) {
        case SLOT_COUNT: v0 = cnt; break;
        case SLOT_SUM_INT: v0 = wave_reduce_u64(F0[s], [](u64 x, u64 y) { return x + y; }); break;
        case SLOT_MIN_U64: v0 = wave_reduce_u64(F0[s], [](u64 x, u64 y) { return x < y ? x : y; }); break;
        case SLOT_MAX_U64: v0 = wave_reduce_u64(F0[s], [](u64 x, u64 y) { return x > y ? x : y; }); break;
        case SLOT_MIN_F64: v0 = wave_reduce_u64(F0[s], [](u64 x, u64 y) { return u2d(y) < u2d(x) ? y : x; }); break;
        case SLOT_MAX_F64: v0 = wave_reduce_u64(F0[s], [](u64 x, u64 y) { return u2d(x) < u2d(y) ? y : x; }); break;
        case SLOT_SUM_DD: {
          DD d; d.hi = u2d(F0[s]); d.lo = u2d(F1[s]);
          d = wave_reduce_dd(d); v0 = d2u(d.hi); v1 = d2u(d.lo);
        } break;
        case SLOT_FIRST: case SLOT_LAST: {
          // lanes without a contribution hold the identity row, which never wins
          const bool first = kind == SLOT_FIRST;
          const u64 trow = first ? wave_reduce_u64(F1[s], [](u64 x, u64 y) { return x < y ? x : y; })
                                 : wave_reduce_u64(FC[s] ? F1[s] : 0ull, [](u64 x, u64 y) { return x > y ? x : y; });
          const u64 owner = __ballot(FC[s] != 0 && F1[s] == trow);
          const int src = owner ? __ffsll((long long)owner) - 1 : 0;
          v0 = readlane64(F0[s], src); v1 = trow;
        } break;
        default: break;
      }
      if (lane == 0) { VmAccRec* A = acc_rec(P, (u32)s, wave); A->v0 = v0; A->v1 = v1; A->cnt = cnt; A->pad = 0; }
    }
  }

  __syncthreads();
  // publish this workgroup's partial aggregates: the four wave records in wave order; the
  // combine rule is applied by the finish kernel (fixed shape -> reproducible)
  for (int s = t; s < P.n_slots; s += VM_COMPUTE_THREADS)
    for (int w = 0; w < VM_WAVES; ++w)
      P.wg_partials[((size_t)blockIdx.x * P.n_slots + s) * VM_WAVES + w] = *acc_rec(P, (u32)s, w);
}
#endif  // SSGPU_RTC_PART

#ifndef __HIPCC_RTC__   // runtime compilation (rtc.cpp) specialises the pipeline kernel and, further down, ssgpu_part_agg_kernel
// ---------------------------------------------------------------------------
// finish kernel: combine [grid][n_slots][4 waves] partial records per slot, in
// (workgroup, wave) order, by the slot's rule.  One thread per slot.
// ---------------------------------------------------------------------------

__device__ __forceinline__ void combine_rec(int kind, VmAccRec& acc, const VmAccRec& r) {
  if (r.cnt == 0) return;
  switch (kind) {
    case SLOT_COUNT:
    case SLOT_SUM_INT: acc.v0 += r.v0; break;
    case SLOT_SUM_DD: {
      DD a; a.hi = u2d(acc.v0); a.lo = u2d(acc.v1);
      DD b; b.hi = u2d(r.v0); b.lo = u2d(r.v1);
      a = dd_add(a, b); acc.v0 = d2u(a.hi); acc.v1 = d2u(a.lo);
    } break;
    case SLOT_MIN_U64: acc.v0 = acc.cnt ? (r.v0 < acc.v0 ? r.v0 : acc.v0) : r.v0; break;
    case SLOT_MAX_U64: acc.v0 = acc.cnt ? (r.v0 > acc.v0 ? r.v0 : acc.v0) : r.v0; break;
    case SLOT_MIN_F64: acc.v0 = acc.cnt ? (u2d(r.v0) < u2d(acc.v0) ? r.v0 : acc.v0) : r.v0; break;
    case SLOT_MAX_F64: acc.v0 = acc.cnt ? (u2d(acc.v0) < u2d(r.v0) ? r.v0 : acc.v0) : r.v0; break;
    case SLOT_FIRST: if (!acc.cnt || r.v1 < acc.v1) { acc.v0 = r.v0; acc.v1 = r.v1; } break;
    case SLOT_LAST: if (!acc.cnt || r.v1 >= acc.v1) { acc.v0 = r.v0; acc.v1 = r.v1; } break;
  }
  acc.cnt += r.cnt;
}

// One workgroup per slot: thread t folds records t, t+256, ... in order, then a fixed
// LDS tree folds the 256 thread results.  The shape is fixed, so results are reproducible
// run to run (and exact whenever the partial sums are exact).
// Reducible-state <-> slot records for the multi-GPU exchange.  The state is
// eight u64 arrays of n_slots elements: [sum_i64 | cnt | dd_hi | dd_lo | min_u64 |
// max_u64 | min_f64 | max_f64], each combined across ranks by one element-wise
// all-reduce (sum / sum / sum / sum / min / max / min / max).
__device__ __forceinline__ void slot_to_state(const VmAccRec r, int kind, int s, int n_slots, u64* __restrict__ state) {
  u64* sum_i = state; u64* cnt = state + n_slots; u64* hi = state + 2 * n_slots; u64* lo = state + 3 * n_slots;
  u64* mn = state + 4 * n_slots; u64* mx = state + 5 * n_slots;
  u64* mnf = state + 6 * n_slots; u64* mxf = state + 7 * n_slots;
  const bool fl = kind == SLOT_FIRST || kind == SLOT_LAST;   // value in the sum array, row id in the min / max array
  sum_i[s] = (kind == SLOT_SUM_INT || kind == SLOT_COUNT || fl) ? r.v0 : 0;
  cnt[s] = r.cnt;
  hi[s] = kind == SLOT_SUM_DD ? r.v0 : d2u(0.0);
  lo[s] = kind == SLOT_SUM_DD ? r.v1 : d2u(0.0);
  // signed-order view so that an int64 all-reduce min/max is order-correct for u64 keys
  mn[s] = (kind == SLOT_MIN_U64 && r.cnt) ? (r.v0 ^ 0x8000000000000000ull) : (kind == SLOT_FIRST && r.cnt) ? (r.v1 ^ 0x8000000000000000ull) : 0x7FFFFFFFFFFFFFFFull;
  mx[s] = (kind == SLOT_MAX_U64 && r.cnt) ? (r.v0 ^ 0x8000000000000000ull) : (kind == SLOT_LAST && r.cnt) ? (r.v1 ^ 0x8000000000000000ull) : 0x8000000000000000ull;
  mnf[s] = (kind == SLOT_MIN_F64 && r.cnt) ? r.v0 : d2u(__builtin_inf());
  mxf[s] = (kind == SLOT_MAX_F64 && r.cnt) ? r.v0 : d2u(-__builtin_inf());
}
__device__ __forceinline__ VmAccRec state_to_slot(const u64* __restrict__ state, int kind, int s, int n_slots) {
  VmAccRec r; r.v0 = 0; r.v1 = 0; r.pad = 0;
  r.cnt = state[n_slots + s];
  switch (kind) {
    case SLOT_COUNT: case SLOT_SUM_INT: r.v0 = state[s]; break;
    case SLOT_SUM_DD: {
      // the ranks' hi and lo parts were summed independently; renormalise
      DD a; a.hi = u2d(state[2 * n_slots + s]); a.lo = 0.0;
      a = dd_add_d(a, u2d(state[3 * n_slots + s]));
      r.v0 = d2u(a.hi); r.v1 = d2u(a.lo);
    } break;
    case SLOT_MIN_U64: r.v0 = state[4 * n_slots + s] ^ 0x8000000000000000ull; break;
    case SLOT_MAX_U64: r.v0 = state[5 * n_slots + s] ^ 0x8000000000000000ull; break;
    case SLOT_MIN_F64: r.v0 = state[6 * n_slots + s]; break;
    case SLOT_MAX_F64: r.v0 = state[7 * n_slots + s]; break;
    case SLOT_FIRST: r.v0 = state[s]; r.v1 = state[4 * n_slots + s] ^ 0x8000000000000000ull; break;
    case SLOT_LAST: r.v0 = state[s]; r.v1 = state[5 * n_slots + s] ^ 0x8000000000000000ull; break;
    default: break;
  }
  return r;
}

// `state` (may be NULL): a partial run (multi-GPU) leaves the slot's reducible state next to its record -- one launch less.
__device__ __forceinline__ void emit_scalar_one(const EmitDesc d, const VmAccRec r);
// descs != NULL: the workgroup of slot s also emits the result columns that read it (one launch less per ScalarAggregate run: the
// emit kernel used to follow as a launch of its own)
__global__ __launch_bounds__(256) void ssgpu_finish_slots_kernel(const VmAccRec* __restrict__ partials, int n_slots,
                                                                 int n_parts, const int* __restrict__ slot_kind,
                                                                 VmAccRec* __restrict__ out, u64* __restrict__ state,
                                                                 const EmitDesc* __restrict__ descs, int n_out) {
  __shared__ VmAccRec tree[256];
  const int s = blockIdx.x, t = threadIdx.x;
  const int kind = slot_kind[s];
  VmAccRec acc; acc.v0 = 0; acc.v1 = 0; acc.cnt = 0; acc.pad = 0;
  if (kind == SLOT_SUM_DD) { acc.v0 = d2u(-0.0); acc.v1 = d2u(0.0); }
  for (int i = t; i < n_parts; i += 256) {
    const VmAccRec r = partials[((size_t)(i / VM_WAVES) * n_slots + s) * VM_WAVES + (i % VM_WAVES)];
    combine_rec(kind, acc, r);
  }
  tree[t] = acc;
  __syncthreads();
  for (int stride = 128; stride > 0; stride >>= 1) {
    if (t < stride) { VmAccRec a = tree[t]; combine_rec(kind, a, tree[t + stride]); tree[t] = a; }
    __syncthreads();
  }
  if (t == 0) { out[s] = tree[0]; if (state) slot_to_state(tree[0], kind, s, n_slots, state); }
  if (descs)
    for (int i = t; i < n_out; i += 256) { const EmitDesc d = descs[i]; if (d.slot == s) emit_scalar_one(d, tree[0]); }
}

__global__ void ssgpu_slots_to_state_kernel(const VmAccRec* __restrict__ recs, int n_slots,
                                            const int* __restrict__ slot_kind, u64* __restrict__ state) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  slot_to_state(recs[s], slot_kind[s], s, n_slots, state);
}
__global__ void ssgpu_state_to_slots_kernel(const u64* __restrict__ state, int n_slots,
                                            const int* __restrict__ slot_kind, VmAccRec* __restrict__ recs) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  recs[s] = state_to_slot(state, slot_kind[s], s, n_slots);
}

// Emit one aggregate result column element from its final slot record.
// out_kind encodes the conversion from the accumulator domain to the column type.

__device__ __forceinline__ void emit_value(void* dst, size_t idx, int out_kind, u64 v0, u64 v1) {
  switch (out_kind) {
    case EMIT_U64: reinterpret_cast<u64*>(dst)[idx] = v0; break;
    case EMIT_I64KEY: reinterpret_cast<i64*>(dst)[idx] = unkey_i64(v0); break;
    case EMIT_U32: reinterpret_cast<u32*>(dst)[idx] = (u32)v0; break;
    case EMIT_I32KEY: reinterpret_cast<i32*>(dst)[idx] = (i32)unkey_i64(v0); break;
    case EMIT_F64: reinterpret_cast<double*>(dst)[idx] = u2d(v0); break;
    case EMIT_F32: reinterpret_cast<float*>(dst)[idx] = (float)u2d(v0); break;
    case EMIT_DD_F64: { double hi = u2d(v0), lo = u2d(v1);
      reinterpret_cast<double*>(dst)[idx] = (lo == 0.0) ? hi : hi + lo; } break;
    case EMIT_DDRES_F64: { const double hi = u2d(v0), lo = u2d(v1), sm = hi + lo, bp = sm - hi;   // TwoSum: exact
      reinterpret_cast<double*>(dst)[idx] = (lo == 0.0) ? 0.0 : (hi - (sm - bp)) + (lo - bp); } break;
    case EMIT_DD_F32: { double hi = u2d(v0), lo = u2d(v1);
      reinterpret_cast<float*>(dst)[idx] = (float)((lo == 0.0) ? hi : hi + lo); } break;
    case EMIT_U8: reinterpret_cast<u8*>(dst)[idx] = (u8)(v0 != 0); break;
    default: break;
  }
}


__device__ __forceinline__ void emit_scalar_one(const EmitDesc d, const VmAccRec r) {
  if (d.out_kind == EMIT_CNT_U64 || d.out_kind == EMIT_CNT_U32)   // COUNT(*) sharing another slot's row count
    emit_value(d.data, 0, d.out_kind == EMIT_CNT_U64 ? EMIT_U64 : EMIT_U32, r.cnt, 0);
  else
    emit_value(d.data, 0, d.out_kind, r.v0, r.v1);
  if (d.is_null) d.is_null[0] = r.cnt == 0;
}
__global__ void ssgpu_emit_scalar_kernel(const VmAccRec* __restrict__ recs, const EmitDesc* __restrict__ descs,
                                         int n_out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const EmitDesc d = descs[i];
  emit_scalar_one(d, recs[d.slot]);
}

// ---------------------------------------------------------------------------
// exclusive scan of per-tile counts (single workgroup; n is #tiles, small)
// out[i] = sum_{j<i} in[j]; total written to *total.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ssgpu_scan_counts_kernel(const u32* __restrict__ in, u32* __restrict__ out,
                                                                  int n, u64* __restrict__ total) {
  __shared__ u64 wave_sums[16];
  __shared__ u64 carry_s;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024 * 8) {
    // each thread owns 8 consecutive elements
    u32 v[8]; u64 sum = 0;
    int i0 = base + t * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = (i0 + j < n) ? in[i0 + j] : 0u; sum += v[j]; }
    // inclusive scan of `sum` across the wave by shuffles
    u64 inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { u64 o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    u64 wpre = 0;
    for (int w = 0; w < wave; ++w) wpre += wave_sums[w];
    u64 run = carry_s + wpre + (inc - sum);
#pragma unroll
    for (int j = 0; j < 8; ++j) { if (i0 + j < n) out[i0 + j] = (u32)run; run += v[j]; }
    __syncthreads();
    if (t == 1023) carry_s = run;
    __syncthreads();
  }
  if (t == 0) *total = carry_s;
}

// ---------------------------------------------------------------------------
// group table extraction: occupied slots -> dense result rows (slot order made
// deterministic by a count / scan / scatter over 512-slot tiles).
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool group_slot_occupied(const GroupExtractParams& P, u32 slot) {
  if (slot > P.capacity + P.extra_slots) return false;
  // the special slot `capacity` owns the key whose value equals the EMPTY sentinel
  return P.keys[slot] != VM_KEY_EMPTY;  // the special slot holds 0 once its key was seen
}

__global__ __launch_bounds__(256) void ssgpu_group_count_kernel(const GroupExtractParams P, u32* __restrict__ tile_counts) {
  __shared__ u32 wsum[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  u32 s0 = blockIdx.x * 512u + t * 2u;
  bool o0 = group_slot_occupied(P, s0);
  bool o1 = group_slot_occupied(P, s0 + 1);
  u32 c = (u32)__popcll(__ballot(o0)) + (u32)__popcll(__ballot(o1));
  if (lane == 0) wsum[wave] = c;
  __syncthreads();
  if (t == 0) tile_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// One occupied slot of the table -> result row `row` (keys unpacked, aggregates emitted)
__device__ __forceinline__ void group_extract_slot(const GroupExtractParams& P, u32 slot, u32 row);
// count + scan + extract in ONE launch: tiles of 512 slots take a ticket, publish their occupied-slot count and look back
// over the earlier tickets for their first result row (decoupled look-back, the status words stamped with the run's epoch
// so that the buffer is never cleared).  Same row order as the three-kernel form (slot order).  ctrl = [rows u64][ticket u32]
// [gave-up u32], zero at launch.  A GroupAggregate run used to end with three dependent small launches; at the row counts
// of a sharded step (1e5 groups: 513 tiles) they cost more than the work.
__global__ __launch_bounds__(256) void ssgpu_group_extract_lb_kernel(const GroupExtractParams P, unsigned long long* __restrict__ status, u64 tag,
                                                                     unsigned int* __restrict__ ctrl, unsigned int* __restrict__ error_flag) {
  __shared__ u32 wsum[4];
  __shared__ u32 s_tile, s_base;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) s_tile = atomicAdd(ctrl + 2, 1u);
  __syncthreads();
  const u32 tile = s_tile;
  const u32 s0 = tile * 512u + (u32)t * 2u;
  const bool o0 = group_slot_occupied(P, s0), o1 = group_slot_occupied(P, s0 + 1);
  const u64 b0 = __ballot(o0), b1 = __ballot(o1);
  if (lane == 0) wsum[wave] = (u32)__popcll(b0) + (u32)__popcll(b1);
  __syncthreads();
  if (wave == 0) {
    const u32 run = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const u32 excl = lookback_rows_before(status, (int)tile, run, tag, ctrl, error_flag, lane);
    if (lane == 0) {
      s_base = excl;
      if (tile == gridDim.x - 1u) *reinterpret_cast<u64*>(ctrl) = (u64)excl + run;
    }
  }
  __syncthreads();
  u32 base = s_base;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  const u64 lt = (1ull << lane) - 1ull;
  const u32 r0 = base + (u32)__popcll(b0 & lt) + (u32)__popcll(b1 & lt);
  if (o0) group_extract_slot(P, s0, r0);
  if (o1) group_extract_slot(P, s0 + 1, r0 + (o0 ? 1u : 0u));
}

__global__ __launch_bounds__(256) void ssgpu_group_extract_kernel(const GroupExtractParams P) {
  __shared__ u32 wsum[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  u32 s0 = blockIdx.x * 512u + t * 2u;
  bool o0 = group_slot_occupied(P, s0);
  bool o1 = group_slot_occupied(P, s0 + 1);
  u64 b0 = __ballot(o0), b1 = __ballot(o1);
  if (lane == 0) wsum[wave] = (u32)__popcll(b0) + (u32)__popcll(b1);
  __syncthreads();
  u32 base = P.tile_offsets[blockIdx.x];
  for (int w = 0; w < wave; ++w) base += wsum[w];
  const u64 lt = (1ull << lane) - 1ull;
  u32 r0 = base + (u32)__popcll(b0 & lt) + (u32)__popcll(b1 & lt);
  if (o0) group_extract_slot(P, s0, r0);
  if (o1) group_extract_slot(P, s0 + 1, r0 + (o0 ? 1u : 0u));
}
__device__ __forceinline__ void group_extract_slot(const GroupExtractParams& P, u32 slot, u32 row) {
  {
    const u64 key = slot == P.capacity ? VM_KEY_EMPTY : P.keys[slot];
    for (u32 q = 0; q < P.n_keys; ++q) {
      const GroupKeyOut ko = P.keys_out[q];
      u64 field = key >> ko.shift;
      bool isnull = ko.nullbit != 0xFF && ((key >> ko.nullbit) & 1ull);
      u64 vmask = ko.bits >= 64 ? ~0ull : ((1ull << ko.bits) - 1ull);
      u64 v = isnull ? 0ull : (field & vmask);
      if (ko.width == 8) reinterpret_cast<u64*>(ko.data)[row] = v;
      else if (ko.width == 4) reinterpret_cast<u32*>(ko.data)[row] = (u32)v;
      else reinterpret_cast<u8*>(ko.data)[row] = (u8)v;
      if (ko.is_null) ko.is_null[row] = isnull;
    }
    for (u32 q = 0; q < P.n_aggs_out; ++q) {
      const GroupAggOut ao = P.aggs_out[q];
      u64 v0 = P.acc[(u64)slot * P.n_gaggs + ao.s];
      u32 c = ao.has_cnt ? P.cnt[(u64)slot * P.n_gaggs + ao.s] : 1u;
      if (ao.out_kind == EMIT_FKEY_F64) {  // ordered-double key back to double bits
        v0 = (v0 & 0x8000000000000000ull) ? (v0 & 0x7FFFFFFFFFFFFFFFull) : ~v0;
        emit_value(ao.data, row, EMIT_F64, v0, 0);
      } else if (ao.out_kind == EMIT_FKEY_F32) {
        v0 = (v0 & 0x8000000000000000ull) ? (v0 & 0x7FFFFFFFFFFFFFFFull) : ~v0;
        emit_value(ao.data, row, EMIT_F32, v0, 0);
      } else {
        emit_value(ao.data, row, ao.out_kind, v0, (ao.out_kind == EMIT_DD_F64 || ao.out_kind == EMIT_DDRES_F64) ? P.acc[(u64)slot * P.n_gaggs + ao.s + 1] : 0ull);
      }
      if (ao.is_null) ao.is_null[row] = c == 0;
    }
  }
}

#endif  // __HIPCC_RTC__

#if !defined(__HIPCC_RTC__) || defined(SSGPU_RTC_PART)
// ---------------------------------------------------------------------------
// partitioned GroupAggregate, phase 2 (see PartAggParams in launch.h): one workgroup per hash partition
// reads the partition's records (n_segs segments, one per workgroup of the scatter pass), aggregates them
// in an LDS table and dumps the table into its own slot range of the global table -- no global atomic.
// Everything the inner loop touches is LDS through address-space-3 pointers (ds_* instructions, not flat):
// measured on this part, ds_add_f64 runs at 3 lanes / clock / CU, ds_min / max / add_u64 at 5.5, 32-bit
// adds at 7 (tools/microbench/lds_atomics.hip) -- a 12-aggregate row costs about 3 clocks of atomics.
// DOUBLE sums are compensated (the returning atomic gives the exact rounding error of every add).
// The one key whose packed value equals the EMPTY sentinel lives in the reserved table entry C.
// ---------------------------------------------------------------------------
// Specialised by runtime compilation (rtc.cpp: ssgpu_rtc_specialize_part_agg, -DSSGPU_RTC_PART): the aggregates'
// descriptors, the record and accumulator widths and the LDS size are constants of rtc_part.h and the loop over the
// aggregates is unrolled, so every field offset, opcode switch and record-word select below folds to the one instruction it
// stands for -- the generic kernel spends ~50 vector instructions per aggregate and wave on that decoding.
#ifdef SSGPU_RTC_PART
#include "rtc_part.h"   // kPartNAggs, kPartDesc[], kPartW, kPartNg, kPartAnyCnt, kPartLdsBytes
#define PART_NG(P) kPartNg
#define PART_W(P) kPartW
#define PART_ANY_CNT(P) (kPartAnyCnt != 0u)
#define PART_NAGGS(P) kPartNAggs
#define PART_AGG_LOOP _Pragma("unroll") for (u32 s = 0; s < kPartNAggs; ++s)
#define PART_DESC(s) kPartDesc[s]
#define PART_DENSE(P) (kPartDense != 0u)
#define PART_SPLIT(P) (kPartSplit != 0u)
#else
#define PART_NG(P) (P).n_gaggs
#define PART_W(P) (P).rec_words
#define PART_ANY_CNT(P) ((P).any_cnt != 0u)
#define PART_NAGGS(P) (P).n_aggs
#define PART_AGG_LOOP for (u32 s = 0; s < n_aggs; ++s)
#define PART_DESC(s) readlane64(mydesc, (int)(s))
#define PART_DENSE(P) ((P).dense.on != 0u)
#define PART_SPLIT(P) ((P).split != 0u)
#endif
#ifndef FKEY   /* (vm_body.inc defines the same for the pipeline kernel's GAGG handlers) */
#define FKEY(d) ({ u64 b_ = d2u((double)(d)); (b_ & 0x8000000000000000ull) ? ~b_ : (b_ | 0x8000000000000000ull); })
#endif
/* PART_ROWS: records per lane per step -- a constant of part_agg_body (2; the resident form, at one workgroup per CU, takes RESIDENT_ROWS) */
#ifndef SSGPU_RESIDENT_ROWS
#define SSGPU_RESIDENT_ROWS 2   /* (4 and 8 measured the same: the kernel is bound by instruction issue, ~350 VALU instructions per row) */
#endif
#ifndef SSGPU_PART_NT
#define SSGPU_PART_NT 0         /* 1: partition records are read with non-temporal loads (read once; A/B through SSGPU_RTC_FLAGS) */
#endif
#if SSGPU_PART_NT
#define PART_LD(p) __builtin_nontemporal_load(p)
#else
#define PART_LD(p) (*(p))
#endif
#ifndef SSGPU_PART_ROWS
#define SSGPU_PART_ROWS 2       /* records per lane per step of the record form (4: measured the same, profiles/r06_part_rows.txt) */
#endif
#define LDS_AS __attribute__((address_space(3)))
template <int MAXW> struct RecVec { typedef u64 type __attribute__((ext_vector_type(MAXW))); };
template <int MAXW> __device__ __forceinline__ u64 rec_word(const typename RecVec<MAXW>::type& r, u32 w) {
  u64 v = r[0];
#pragma unroll
  for (int k = 1; k < MAXW; ++k) v = (w == (u32)k) ? r[k] : v;   // w is wave-uniform: scalar conditions, no register indexing
  return v;
}
__device__ __forceinline__ u64 rec_field(u64 word, u32 off, u32 width) {
  const u64 vmask = width >= 8 ? ~0ull : ((1ull << (width * 8u)) - 1ull);
  return (word >> ((off & 7u) * 8u)) & vmask;
}
__device__ __forceinline__ void lds_dd_add(LDS_AS double* acc, double v) {
  const double old = __hip_atomic_fetch_add(acc, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const double t = old + v;
  const double bp = t - old;
  const double err = (old - (t - bp)) + (v - bp);
  if (err != 0.0) __hip_atomic_fetch_add(acc + 1, err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#define LDS_MIN(A, x) __hip_atomic_fetch_min((A), (u64)(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define LDS_MAX(A, x) __hip_atomic_fetch_max((A), (u64)(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define LDS_ADD(A, x) __hip_atomic_fetch_add((A), (u64)(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define PART_APPLY(OPNAME, EXPR)                                                      \
  case VM_##OPNAME: {                                                                 \
    _Pragma("unroll") for (int j = 0; j < PART_ROWS; ++j) if (ok[j]) { LDS_AS u64* A = lacc + li[j] + word; const u64 raw = val[j]; EXPR; } \
  } break;
#define PART_APPLY_FLOAT(OPNAME, EDECL, LDSOP, NEUTRAL)                               \
  case VM_##OPNAME: {                                                                 \
    _Pragma("unroll") for (int j = 0; j < PART_ROWS; ++j) {                           \
      const u64 raw = val[j]; EDECL;                                                  \
      const bool isnan = !(e == e);                                                    \
      const u64 k = isnan ? (u64)(NEUTRAL) : FKEY(e);                                  \
      nan_acc |= (isnan && ok[j]) ? 1u : 0u;                                           \
      if (ok[j]) { LDS_AS u64* A = lacc + li[j] + word; LDSOP(A, k); }                \
    }                                                                                 \
  } break;
// exclusive scan of one value per thread over the workgroup (1024 threads); *total gets the sum
__device__ __forceinline__ u32 part_block_scan(u32 v, LDS_AS u32* wsum, u32 t, u32* total) {
  const u32 lane = t & 63u, wave = t >> 6;
  u32 inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d); if (lane >= (u32)d) inc += o; }
  if (lane == 63u) wsum[wave] = inc;
  __syncthreads();
  u32 pre = 0, tot = 0;
#pragma unroll
  for (u32 w = 0; w < SSGPU_PART_THREADS / 64; ++w) { const u32 x = wsum[w]; if (w < wave) pre += x; tot += x; }
  if (total) *total = tot;
  __syncthreads();
  return pre + inc - v;
}

// One predicate of a plain stage on one row (the same function as group_scatter_kernel.hip's pscat_pred).
__device__ __forceinline__ bool plain_pred(const void* data, const u8* nulls, u64 c, u32 kind, u32 cmp, bool col_on_left, u64 row) {
  const bool is_null = nulls ? nulls[row] != 0 : false;    // a NULL predicate drops the row (filter.cc:170-199)
  bool lt, gt, eq;   // column < constant, column > constant, column == constant
  switch (kind) {
    case 0: { const i32 v = reinterpret_cast<const i32*>(data)[row], k = (i32)(u32)c; lt = v < k; gt = v > k; eq = v == k; } break;
    case 1: { const u32 v = reinterpret_cast<const u32*>(data)[row], k = (u32)c; lt = v < k; gt = v > k; eq = v == k; } break;
    case 2: { const i64 v = reinterpret_cast<const i64*>(data)[row], k = (i64)c; lt = v < k; gt = v > k; eq = v == k; } break;
    case 3: { const u64 v = reinterpret_cast<const u64*>(data)[row], k = c; lt = v < k; gt = v > k; eq = v == k; } break;
    case 4: { const float v = reinterpret_cast<const float*>(data)[row], k = __uint_as_float((u32)c); lt = v < k; gt = v > k; eq = v == k; } break;
    case 6: { const u8 v = reinterpret_cast<const u8*>(data)[row], k = (u8)c; lt = v < k; gt = v > k; eq = v == k; } break;   // a BOOL column as the predicate: (column != FALSE)
    default: { const double v = reinterpret_cast<const double*>(data)[row], k = u2d(c); lt = v < k; gt = v > k; eq = v == k; } break;
  }
  bool r;
  switch (cmp) {
    case 0: r = col_on_left ? lt : gt; break;
    case 1: r = col_on_left ? (lt || eq) : (gt || eq); break;
    case 2: r = eq; break;
    default: r = !eq; break;
  }
  return r && !is_null;
}
struct NoRowSource {};
#ifdef SSGPU_RTC_PART_PLAIN   /* rtc_part.h of a resident-form build: the row source's descriptors are constants, too */
#define RS_NKEYS kRsNKeys
#define RS_KEY_WIDTH(k) kRsKeyWidth[k]
#define RS_KEY_SHIFT(k) kRsKeyShift[k]
#define RS_KEY_BITS(k) kRsKeyBits[k]
#define RS_KEY_NULLBIT(k) kRsKeyNullbit[k]
#define RS_NFIELDS kRsNFields
#define RS_FIELD_WIDTH(f) kRsFieldWidth[f]
#define RS_FIELD_OFF(f) kRsFieldOff[f]
#define RS_NPREDS kRsNPreds
#define RS_PRED_KIND(q) kRsPredKind[q]
#define RS_PRED_CMP(q) kRsPredCmp[q]
#define RS_PRED_COL_LEFT(q) (kRsPredColLeft[q] != 0u)
#define RS_UNROLL _Pragma("unroll")
#else
#define RS_NKEYS S.n_keys
#define RS_KEY_WIDTH(k) S.keys[k].width
#define RS_KEY_SHIFT(k) S.keys[k].shift
#define RS_KEY_BITS(k) S.keys[k].bits
#define RS_KEY_NULLBIT(k) S.keys[k].nullbit
#define RS_NFIELDS S.n_fields
#define RS_FIELD_WIDTH(f) S.fields[f].width
#define RS_FIELD_OFF(f) S.fields[f].off
#define RS_NPREDS S.n_preds
#define RS_PRED_KIND(q) S.preds[q].kind
#define RS_PRED_CMP(q) S.preds[q].cmp
#define RS_PRED_COL_LEFT(q) (S.preds[q].col_on_left != 0u)
#define RS_UNROLL
#endif
__device__ __forceinline__ bool plain_pred_value(u64 raw, bool is_null, u64 c, u32 kind, u32 cmp, bool col_on_left) {
  bool lt, gt, eq;
  switch (kind) {
    case 0: { const i32 v = (i32)(u32)raw, k = (i32)(u32)c; lt = v < k; gt = v > k; eq = v == k; } break;
    case 1: { const u32 v = (u32)raw, k = (u32)c; lt = v < k; gt = v > k; eq = v == k; } break;
    case 2: { const i64 v = (i64)raw, k = (i64)c; lt = v < k; gt = v > k; eq = v == k; } break;
    case 3: { const u64 v = raw, k = c; lt = v < k; gt = v > k; eq = v == k; } break;
    case 4: { const float v = __uint_as_float((u32)raw), k = __uint_as_float((u32)c); lt = v < k; gt = v > k; eq = v == k; } break;
    case 6: { const u8 v = (u8)raw, k = (u8)c; lt = v < k; gt = v > k; eq = v == k; } break;
    default: { const double v = u2d(raw), k = u2d(c); lt = v < k; gt = v > k; eq = v == k; } break;
  }
  bool r;
  switch (cmp) {
    case 0: r = col_on_left ? lt : gt; break;
    case 1: r = col_on_left ? (lt || eq) : (gt || eq); break;
    case 2: r = eq; break;
    default: r = !eq; break;
  }
  return r && !is_null;
}
__device__ __forceinline__ u64 load_by_width(const void* p, u32 width, u64 row) {
  return width == 8u ? reinterpret_cast<const u64*>(p)[row] : width == 4u ? (u64)reinterpret_cast<const u32*>(p)[row] : (u64)reinterpret_cast<const u8*>(p)[row];
}
#ifdef SSGPU_RTC_PART_PLAIN
// The specialised resident kernel is software-pipelined: the column loads of step i + 1 are issued (into these registers:
// every bound below is a constant of the build) before step i's LDS work starts and are first looked at one iteration
// later.  At one 1024-thread workgroup per CU there are too few waves for the loads of some to hide behind the LDS
// work of others: without this the two phases simply add up (measured: 0.9 ms of loads + 0.9 ms of LDS work = 1.8 ms).
template <int ROWS> struct ResidentRaw {
  u64 k[ROWS][kRsNKeys ? kRsNKeys : 1]; u32 kn[ROWS][kRsNKeys ? kRsNKeys : 1];
  u64 f[ROWS][kRsNFields ? kRsNFields : 1];
  u64 p[ROWS][kRsNPreds ? kRsNPreds : 1]; u32 pn[ROWS][kRsNPreds ? kRsNPreds : 1];
  bool in[ROWS];
};
template <int ROWS>
__device__ __forceinline__ void resident_issue(const PlainScatterParams& S, u64 base, u64 n_rows, u32 t, ResidentRaw<ROWS>& R) {
#pragma unroll
  for (int j = 0; j < ROWS; ++j) {
    const u64 row = base + (u64)j * SSGPU_PART_THREADS + t;
    R.in[j] = row < n_rows;
    const u64 rowc = R.in[j] ? row : 0ull;       // unconditional loads on a row that exists
#pragma unroll
    for (u32 q = 0; q < kRsNPreds; ++q) {
      R.p[j][q] = load_by_width(S.preds[q].data, kRsPredKind[q] == 6u ? 1u : (kRsPredKind[q] == 0u || kRsPredKind[q] == 1u || kRsPredKind[q] == 4u) ? 4u : 8u, rowc);
      R.pn[j][q] = S.preds[q].nulls ? (u32)S.preds[q].nulls[rowc] : 0u;
    }
#pragma unroll
    for (u32 k = 0; k < kRsNKeys; ++k) {
      R.k[j][k] = load_by_width(S.keys[k].data, kRsKeyWidth[k], rowc);
      R.kn[j][k] = S.keys[k].nulls ? (u32)S.keys[k].nulls[rowc] : 0u;
    }
#pragma unroll
    for (u32 f = 0; f < kRsNFields; ++f) R.f[j][f] = S.fields[f].src ? load_by_width(S.fields[f].src, kRsFieldWidth[f], rowc) : 0ull;
  }
}
#endif
// PLAIN (the LDS-RESIDENT form of a plain stage, ssgpu_group_resident_kernel): there are no records in memory at all --
// every workgroup holds a table of ALL groups (the slab form's table and end-of-kernel merge) and assembles its rows'
// records in registers straight from the input columns, exactly as the plain scatter would have laid them out (key word,
// then the field list of PlainScatterParams), so the aggregates' descriptors are the same.  40 bytes per row cross HBM
// once instead of three times.
template <int MAXW, bool PLAIN, typename SRC>
__device__ __forceinline__ void part_agg_body(const PartAggParams& P, const SRC& S) {
  typedef typename RecVec<MAXW>::type Rec;
  constexpr int PART_ROWS = PLAIN ? SSGPU_RESIDENT_ROWS : SSGPU_PART_ROWS;
  const u32 t = threadIdx.x, part = blockIdx.x;
  const u32 C = P.local_capacity, ng = PART_NG(P), W = PART_W(P);
  const bool any_cnt = PART_ANY_CNT(P);
  // slab mode: this workgroup's segments are a run of the single partition's segments
  const u32 seg0 = P.slab_segs ? part * P.slab_segs : 0u;
  const u32 G = PLAIN ? 0u : P.slab_segs ? (seg0 < P.n_segs ? (P.n_segs - seg0 < P.slab_segs ? P.n_segs - seg0 : P.slab_segs) : 0u) : P.n_segs;
  // an entry's accumulator words are `st` words apart, st odd: the lanes of a wave hit word k of 64 different entries,
  // and with an even stride (16 words = 128 B) those addresses fall into two LDS banks -- a 32-way conflict on every
  // atomic (measured: 5x slower)
  const u32 st = ng | 1u;
  // LDS carve-up (C + 1 table entries: entry C belongs to the EMPTY-valued key)
#ifdef SSGPU_RTC_PART
  // a kernel loaded from a module cannot opt into more than 64 KiB of DYNAMIC LDS; its size is known here: static
  __shared__ __attribute__((aligned(16))) char part_lds[kPartLdsBytes];
  LDS_AS u64* const lkeys = (LDS_AS u64*)(LDS_AS char*)part_lds;
#else
  LDS_AS u64* const lkeys = (LDS_AS u64*)0u;
#endif
  LDS_AS u64* const lacc = lkeys + (C + 1u);
  LDS_AS u32* const lcnt = (LDS_AS u32*)(lacc + (size_t)(C + 1u) * st);
  LDS_AS u32* const segoff = lcnt + (any_cnt ? (C + 1u) * st : 0u);   // [G + 1] first record of every segment in the flat order
  LDS_AS u32* const wsum = segoff + G + 1u;                              // [16] scan scratch
  for (u32 e = t; e <= C; e += SSGPU_PART_THREADS) lkeys[e] = VM_KEY_EMPTY;
  for (u32 i = t; i < (C + 1u) * st; i += SSGPU_PART_THREADS) { lacc[i] = P.T.acc_init[(i % st) % ng]; if (any_cnt) lcnt[i] = 0u; }
  if constexpr (PLAIN) {
    if (P.hot_only) {   // heavy hitters: the table holds the hot keys and nothing else.  One thread seeds it, so every workgroup's copy is identical
      if (t < 2u) wsum[t] = 0u;   // which entries a row of THIS run reached (a plan keeps its hot keys across runs; other data may not hold them)
      __syncthreads();
      if (t == 0)
        for (u32 h = 0; h < S.n_hot; ++h) {
          const u64 key = S.hot_keys[h];
          u32 i = __umulhi(hash_local(key), C);
          while (lkeys[i] != VM_KEY_EMPTY && lkeys[i] != key) i = i + 1u == C ? 0u : i + 1u;
          lkeys[i] = key;
        }
    }
  }
  if constexpr (!PLAIN) {
    if (P.accumulate && PART_DENSE(P) && !P.slab_segs) {
      // the table an earlier launch dumped for this partition (same layout as the dump at the end of this function), read back: the
      // rows of this launch are one more range of the same input
      __syncthreads();
      const u32 chunk = part / P.dense.chunk_parts, pin = part - chunk * P.dense.chunk_parts;
      const u64* const keys_c = reinterpret_cast<const u64*>(reinterpret_cast<const char*>(P.T.keys) + (u64)chunk * P.dense.chunk_stride);
      const u64* const acc_c = reinterpret_cast<const u64*>(reinterpret_cast<const char*>(P.T.acc) + (u64)chunk * P.dense.chunk_stride);
      const u32* const cnt_c = reinterpret_cast<const u32*>(reinterpret_cast<const char*>(P.T.cnt) + (u64)chunk * P.dense.chunk_stride);
      const u64 sp = (u64)P.dense.chunk_slots;
      const bool special_used = keys_c[sp] != VM_KEY_EMPTY;
      for (u32 e0 = 0; e0 < C; e0 += SSGPU_PART_THREADS) {
        const u32 e = e0 + t, ec = e < C ? e : 0u;
        const u64 packed = ssgpu_dense_key_of(P.dense, ec * P.dense.n_parts + part);   // (uniform loop over the keys: outside the divergent part)
        if (e < C) {
          const bool from_special = packed == VM_KEY_EMPTY && special_used;
          if (from_special) {
            lkeys[e] = 0ull;
            for (u32 s = 0; s < ng; ++s) { lacc[(size_t)e * st + s] = acc_c[sp * ng + s]; if (any_cnt) lcnt[(size_t)e * st + s] = cnt_c[sp * ng + s]; }
          } else if (keys_c[(u64)pin * C + e] != VM_KEY_EMPTY) lkeys[e] = 0ull;
        }
      }
      __syncthreads();
      for (u32 i = t; i < C * ng; i += SSGPU_PART_THREADS) {
        const u32 e = i / ng, l = e * st + i % ng;
        if (lkeys[e] != VM_KEY_EMPTY && !(special_used && ssgpu_dense_key_of(P.dense, e * P.dense.n_parts + part) == VM_KEY_EMPTY)) {
          lacc[l] = acc_c[(u64)pin * C * ng + i];
          if (any_cnt) lcnt[l] = cnt_c[(u64)pin * C * ng + i];
        }
      }
    }
  }
  u32 total = 0;
  if constexpr (!PLAIN) {
    u32 n = t < G ? P.counts[P.slab_segs ? (u64)(seg0 + t) : (u64)part * G + t] : 0u;    // G <= 1024 (the host caps the scatter grid)
    n = n < P.seg_cap ? n : P.seg_cap;   // (a segment that ran full: the plain scatter's counter keeps counting; the host reruns with larger segments)
    const u32 ex = part_block_scan(n, wsum, t, &total);
    if (t < G) segoff[t] = ex;
    if (t == 0) segoff[G] = total;
  }
  __syncthreads();
  // split records (dense partitions): W - 1 payload words per record in `recs`, the table entry (word 0) in a 16-bit array of its own
  const bool split = !PLAIN && PART_SPLIT(P);
  const u32 RW = split ? W - 1u : W;
  const u64* const recs = PLAIN ? nullptr : P.recs + (P.slab_segs ? (u64)seg0 : (u64)SEG_INDEX(part, 0u, P.n_parts, G)) * P.seg_cap * RW;
  const unsigned short* const rentry = split ? P.recs_entry + (u64)SEG_INDEX(part, 0u, P.n_parts, G) * P.seg_cap : nullptr;
  const u64 seg_step = PLAIN ? 0ull : P.slab_segs ? (u64)P.seg_cap : (u64)(SEG_INDEX(part, 1u, P.n_parts, G) - SEG_INDEX(part, 0u, P.n_parts, G)) * P.seg_cap;   // records between this partition's consecutive segments
  const u32 seg_cap = P.seg_cap, n_aggs = PART_NAGGS(P);
  const u64 mydesc = (t & 63u) < n_aggs ? P.desc[t & 63u] : 0ull;   // lane s of every wave holds aggregate s's descriptor
  (void)mydesc; (void)n_aggs; (void)seg_cap;
  u32 seg = 0;
  // (fetching step i + 1's records while step i is aggregated -- one record per lane per step, two sets in registers to stay
  // under 64 VGPRs -- was tried: 2.95 ms instead of 1.85 ms for config #3; two records per lane and no prefetch it stays)
  // (records: [0, total) of this workgroup's segments.  PLAIN: the rows of the input, tiles dealt round-robin to the workgroups)
  u64 row_first = 0, row_limit = total, row_stride = SSGPU_PART_THREADS * PART_ROWS;
  if constexpr (PLAIN) { row_first = (u64)part * (SSGPU_PART_THREADS * PART_ROWS); row_limit = S.n_rows; row_stride = (u64)gridDim.x * (SSGPU_PART_THREADS * PART_ROWS); }
  u64 trip_limit = row_limit;
  u32 nan_acc = 0;                 // this lane met a NaN in a floating MIN / MAX
  u64 touched = 0;                 // hot_only: the seeded entries this lane's rows reached (local_capacity = SSGPU_HOT_SLOTS = 64)
  bool miss = false;               // dense slots: a row whose key lies outside the ranges
#ifdef SSGPU_RTC_PART_PLAIN
  // Trip k issues the loads of tile k and aggregates tile k - 1 (one more trip than tiles; the first aggregates nothing).
  // There is deliberately no load ahead of the loop: loads pending on entry made the compiler wait, in every trip, for
  // the first of the loads just issued (its wait-count bookkeeping merges the entry edge with the back edge) -- the
  // pipeline then runs but hides nothing.
  ResidentRaw<PART_ROWS> raw = {};
  if constexpr (PLAIN) trip_limit = row_limit + row_stride;
#endif
  // The record form, specialised builds with narrow records (two sets of a record's words fit the 64-VGPR budget): the same pipeline --
  // trip k issues the loads of records [k] and aggregates records [k - 1].  The kernel is bound by these loads (4 GB at 5 TB/s for
  // config #3 with every wave alternating between loading and LDS atomics); see the note above about no load ahead of the loop.
#if defined(SSGPU_RTC_PART) && !defined(SSGPU_PART_NO_PREFETCH)
  constexpr bool kPrefetch = !PLAIN && kPartW <= 6u && kPartPrefetch != 0u;
#else
  constexpr bool kPrefetch = false;
#endif
  Rec pre[PART_ROWS]; bool pre_live[PART_ROWS];
#pragma unroll
  for (int j = 0; j < PART_ROWS; ++j) { pre_live[j] = false; _Pragma("unroll") for (int w = 0; w < MAXW; ++w) pre[j][w] = 0ull; }
  if constexpr (kPrefetch) trip_limit = row_limit + row_stride;
  for (u64 base = row_first; base < trip_limit; base += row_stride) {
    Rec rec[PART_ROWS]; bool live[PART_ROWS]; u32 li[PART_ROWS];
    if constexpr (PLAIN) {
#ifdef SSGPU_RTC_PART_PLAIN
#pragma unroll
      for (int j = 0; j < PART_ROWS; ++j) {         // the rows whose loads were issued one step ago
        live[j] = raw.in[j];
#pragma unroll
        for (u32 q = 0; q < kRsNPreds; ++q)
          live[j] = live[j] & plain_pred_value(raw.p[j][q], raw.pn[j][q] != 0u, S.preds[q].bits, kRsPredKind[q], kRsPredCmp[q], kRsPredColLeft[q] != 0u);
        u64 key = 0ull;
#pragma unroll
        for (u32 k = 0; k < kRsNKeys; ++k) {
          u64 a = raw.k[j][k] & (kRsKeyBits[k] >= 64u ? ~0ull : ((1ull << (kRsKeyBits[k] & 63u)) - 1ull));
          if (raw.kn[j][k]) a = 1ull << ((kRsKeyNullbit[k] - kRsKeyShift[k]) & 63u);
          key |= a << kRsKeyShift[k];
        }
#pragma unroll
        for (int w = 1; w < MAXW; ++w) rec[j][w] = 0ull;
        rec[j][0] = key;
#pragma unroll
        for (u32 f = 0; f < kRsNFields; ++f) rec[j][(kRsFieldOff[f] >> 3) < (u32)MAXW ? (kRsFieldOff[f] >> 3) : 0u] |= raw.f[j][f] << ((kRsFieldOff[f] & 7u) * 8u);
      }
      {   // this tile's loads: in flight during the LDS work on the tile before it (past the end: the clamped row 0, discarded)
        const u64 nb = base;
        resident_issue<PART_ROWS>(S, nb < row_limit ? nb : 0ull, nb < row_limit ? row_limit : 0ull, t, raw);
        // the loads stay HERE: left alone, the compiler sinks them to their first use -- the top of the next trip -- and
        // the pipeline is gone (seen in the ISA: 24 loads, then s_waitcnt vmcnt(22) straight away)
        asm volatile("" ::: "memory");
      }
#else
#pragma unroll
      for (int j = 0; j < PART_ROWS; ++j) {
        const u64 row = base + (u64)j * SSGPU_PART_THREADS + t;
        live[j] = row < row_limit;
        const u64 rowc = live[j] ? row : 0ull;   // every load is unconditional (on a row that exists): the descriptor loops stay out of divergent regions
        RS_UNROLL for (u32 q = 0; q < RS_NPREDS; ++q)
          live[j] = live[j] & plain_pred(S.preds[q].data, S.preds[q].nulls, S.preds[q].bits, RS_PRED_KIND(q), RS_PRED_CMP(q), RS_PRED_COL_LEFT(q), rowc);
        u64 key = 0ull;
        RS_UNROLL for (u32 k = 0; k < RS_NKEYS; ++k) {
          const u32 kw = RS_KEY_WIDTH(k), kbits = RS_KEY_BITS(k), kshift = RS_KEY_SHIFT(k);
          u64 a = kw == 8 ? reinterpret_cast<const u64*>(S.keys[k].data)[rowc] : kw == 4 ? (u64)reinterpret_cast<const u32*>(S.keys[k].data)[rowc]
                                                                                         : (u64)reinterpret_cast<const u8*>(S.keys[k].data)[rowc];
          a &= kbits >= 64 ? ~0ull : ((1ull << kbits) - 1ull);
          if (S.keys[k].nulls && S.keys[k].nulls[rowc]) a = 1ull << (RS_KEY_NULLBIT(k) - kshift);
          key |= a << kshift;
        }
#pragma unroll
        for (int w = 1; w < MAXW; ++w) rec[j][w] = 0ull;
        rec[j][0] = key;
        RS_UNROLL for (u32 f = 0; f < RS_NFIELDS; ++f) {
          const void* src = S.fields[f].src;
          const u32 fw = RS_FIELD_WIDTH(f), fo = RS_FIELD_OFF(f);
          if (!src) continue;                      // an absent NULL mask: zeros
          const u64 v = fw == 8u ? reinterpret_cast<const u64*>(src)[rowc] : fw == 4u ? (u64)reinterpret_cast<const u32*>(src)[rowc] : (u64)reinterpret_cast<const u8*>(src)[rowc];
          const u64 placed = v << ((fo & 7u) * 8u);
          const u32 word = fo >> 3;              // (wave-uniform: scalar compares, no register indexing)
#pragma unroll
          for (int w = 1; w < MAXW; ++w) rec[j][w] |= (word == (u32)w) ? placed : 0ull;
        }
      }
#endif
    } else if constexpr (kPrefetch) {
#pragma unroll
    for (int j = 0; j < PART_ROWS; ++j) { live[j] = pre_live[j]; rec[j] = pre[j]; }       // the records whose loads were issued one trip ago
#pragma unroll
    for (int j = 0; j < PART_ROWS; ++j) {
      const u64 i64 = base + (u64)j * SSGPU_PART_THREADS + t;
      pre_live[j] = i64 < (u64)total;
      const u32 i = pre_live[j] ? (u32)i64 : 0u;
      if (pre_live[j]) {
        while (i >= segoff[seg + 1u]) ++seg;
        const u64* rp = recs + ((u64)seg * seg_step + (i - segoff[seg])) * W;
#pragma unroll
        for (int w = 0; w < MAXW; ++w) pre[j][w] = (u32)w < W ? PART_LD(rp + w) : 0ull;
      }
    }
    asm volatile("" ::: "memory");     // (the loads stay here: left alone the compiler sinks them to their first use, the top of the next trip)
    } else {
#pragma unroll
    for (int j = 0; j < PART_ROWS; ++j) {
      const u32 i = (u32)base + (u32)j * SSGPU_PART_THREADS + t;
      live[j] = i < total;
      if (live[j]) {
        while (i >= segoff[seg + 1u]) ++seg;              // empty segments are stepped over
        const u64 ri = (u64)seg * seg_step + (i - segoff[seg]);
        if (split) {
          // (RW even -- a specialised build knows it: the payload is 16-byte aligned and leaves as dwordx4 loads)
          const u64* rp = recs + ri * RW;
          if (!(RW & 1u)) rp = reinterpret_cast<const u64*>(__builtin_assume_aligned(rp, 16));
          rec[j][0] = (u64)__builtin_nontemporal_load(rentry + ri);
#pragma unroll
          for (int w = 1; w < MAXW; ++w) rec[j][w] = (u32)w < W ? __builtin_nontemporal_load(rp + (w - 1)) : 0ull;
        } else {
        const u64* rp = recs + ri * W;
#pragma unroll
        for (int w = 0; w < MAXW; ++w) rec[j][w] = (u32)w < W ? PART_LD(rp + w) : 0ull;
        }
      } else {
#pragma unroll
        for (int w = 0; w < MAXW; ++w) rec[j][w] = 0ull;
      }
    }
    }
    if constexpr (PLAIN) {
      if (PART_DENSE(P)) {   // dense slots: the packed key becomes the group's dense index (outside every divergent region: a uniform loop over the keys)
#pragma unroll
        for (int j = 0; j < PART_ROWS; ++j) {
          u32 idx;
          const bool in = ssgpu_dense_index(P.dense, rec[j][0], &idx);
          miss = miss || (live[j] && !in);
          live[j] = live[j] && in;
          rec[j][0] = (u64)idx;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < PART_ROWS; ++j) {
      li[j] = C * st;
      const u64 key = rec[j][0];
      if (P.debug & 2u) { li[j] = __umulhi(hash_local(key), C) * st; continue; }   // development: no probe
      if (live[j]) {
        if (PART_DENSE(P)) {
          // no probe: the entry IS the index (one table of all slots) or index / partitions; the key word only marks the entry used
          u32 pp;
          const u32 e = (P.slab_segs || split) ? (u32)key : ssgpu_dense_entry(P.dense, (u32)key, &pp);
          li[j] = e * st;
          lkeys[e] = 0ull;
        } else if (PLAIN && P.hot_only) {
          // only the seeded keys have an entry: an EMPTY entry on the probe path means "not a heavy hitter" -- the row belongs to the scatter
          u32 i = __umulhi(hash_local(key), C), found = 0xFFFFFFFFu;
          if (key != VM_KEY_EMPTY)
            for (u32 probe = 0; probe < C; ++probe) {
              const u64 cur = lkeys[i];
              if (cur == key) { found = i; break; }
              if (cur == VM_KEY_EMPTY) break;
              i = i + 1u == C ? 0u : i + 1u;
            }
          if (found == 0xFFFFFFFFu) live[j] = false; else { li[j] = found * st; touched |= 1ull << (found & 63u); }
        } else if (key == VM_KEY_EMPTY) {
          lkeys[C] = 0ull;                               // marks the reserved entry as used
        } else {
          u32 i = __umulhi(hash_local(key), C), found = 0xFFFFFFFFu;
          const int probe_limit = P.slab_segs ? (int)C : 32;   // hash partitions: a longer cluster = the table is too full, the host re-partitions finer
          for (int probe = 0; probe < probe_limit; ++probe) {
            const u64 cur = __hip_atomic_load(lkeys + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (cur == key) { found = i; break; }
            if (cur == VM_KEY_EMPTY) {
              u64 expect = VM_KEY_EMPTY;
              if (__hip_atomic_compare_exchange_strong(lkeys + i, &expect, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) || expect == key) { found = i; break; }
            }
            i = i + 1u == C ? 0u : i + 1u;
          }
          if (found == 0xFFFFFFFFu) { atomicExch(P.T.overflow, 1u); live[j] = false; }   // partition too large: the host re-partitions finer
          else li[j] = found * st;
        }
      }
    }
    u32 last_off = 0xFFFFFFFFu; u64 val[PART_ROWS];
#pragma unroll
    for (int j = 0; j < PART_ROWS; ++j) val[j] = 0;
    if (P.debug & 1u) {   // development: records loaded and probed, nothing aggregated
#pragma unroll
      for (int j = 0; j < PART_ROWS; ++j) if (live[j]) LDS_ADD(lacc + li[j], rec[j][1] ^ rec[j][(MAXW - 1) & 4]);
      continue;
    }
    PART_AGG_LOOP {
      // the aggregate's descriptor comes out of a register (lane s of `mydesc`): a scalar memory load here would
      // share its wait counter with the LDS atomics in flight and drain them once per aggregate
      const u64 d = PART_DESC(s);
      const u32 op = (u32)(d & 0xFFFFu), word = (u32)(d >> 16) & 0xFFu, voff = (u32)(d >> 24) & 0xFFu, vw = (u32)(d >> 32) & 0xFFu,
                noff = (u32)(d >> 40) & 0xFFu;
      const bool has_cnt = ((d >> 48) & 1ull) != 0;
      if (voff != 0xFFu && voff != last_off) {   // MIN / MAX / SUM of one column share the fetched value
        last_off = voff;
#pragma unroll
        for (int j = 0; j < PART_ROWS; ++j) val[j] = rec_field(rec_word<MAXW>(rec[j], voff >> 3), voff, vw);
      }
      bool ok[PART_ROWS];
#pragma unroll
      for (int j = 0; j < PART_ROWS; ++j) {
        ok[j] = live[j];
        if (noff != 0xFFu) ok[j] = ok[j] && rec_field(rec_word<MAXW>(rec[j], noff >> 3), noff, 1) == 0;   // NULL input: skipped
        if (has_cnt && ok[j]) __hip_atomic_fetch_add(lcnt + li[j] + word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      switch (op) {
        PART_APPLY(GAGG_COUNT, LDS_ADD(A, 1ull))
        PART_APPLY(GAGG_SUM_I32, LDS_ADD(A, (u64)(i64)(i32)(u32)raw))
        PART_APPLY(GAGG_SUM_U32, LDS_ADD(A, (u64)(u32)raw))
        PART_APPLY(GAGG_SUM_I64, LDS_ADD(A, raw))
        PART_APPLY(GAGG_SUM_F32, __hip_atomic_fetch_add((LDS_AS double*)A, (double)__uint_as_float((u32)raw), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
        case VM_GAGG_SUM_F64: {
          // compensated: the returning atomic gives the exact rounding error of the add.  All rows' adds are issued before
          // the first returned value is looked at -- one LDS round trip per aggregate instead of one per row
          double old[PART_ROWS];
          _Pragma("unroll") for (int j = 0; j < PART_ROWS; ++j) {
            old[j] = 0.0;
            if (ok[j]) old[j] = __hip_atomic_fetch_add((LDS_AS double*)(lacc + li[j] + word), u2d(val[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          _Pragma("unroll") for (int j = 0; j < PART_ROWS; ++j) {
            const double v = u2d(val[j]), t2 = old[j] + v, bp = t2 - old[j], err = (old[j] - (t2 - bp)) + (v - bp);
            if (ok[j] && err != 0.0) __hip_atomic_fetch_add((LDS_AS double*)(lacc + li[j] + word) + 1, err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        } break;
        PART_APPLY(GAGG_MIN_I32, LDS_MIN(A, key_i64((i64)(i32)(u32)raw)))
        PART_APPLY(GAGG_MIN_U32, LDS_MIN(A, (u64)(u32)raw))
        PART_APPLY(GAGG_MIN_I64, LDS_MIN(A, key_i64((i64)raw)))
        PART_APPLY(GAGG_MIN_U64, LDS_MIN(A, raw))
        PART_APPLY(GAGG_MIN_B8, LDS_MIN(A, (u64)((raw & 0xFFull) != 0)))
        PART_APPLY(GAGG_MAX_I32, LDS_MAX(A, key_i64((i64)(i32)(u32)raw)))
        PART_APPLY(GAGG_MAX_U32, LDS_MAX(A, (u64)(u32)raw))
        PART_APPLY(GAGG_MAX_I64, LDS_MAX(A, key_i64((i64)raw)))
        PART_APPLY(GAGG_MAX_U64, LDS_MAX(A, raw))
        PART_APPLY(GAGG_MAX_B8, LDS_MAX(A, (u64)((raw & 0xFFull) != 0)))
        // floating MIN / MAX: a NaN becomes the operation's neutral key and is remembered in a register (the stage's flag is
        // raised once, after the loop) -- no branch per value; the key of a value is computed outside the `ok` region, so
        // the MIN and the MAX of one column share it
        PART_APPLY_FLOAT(GAGG_MIN_F32, const float e = __uint_as_float((u32)raw), LDS_MIN, ~0ull)
        PART_APPLY_FLOAT(GAGG_MIN_F64, const double e = u2d(raw), LDS_MIN, ~0ull)
        PART_APPLY_FLOAT(GAGG_MAX_F32, const float e = __uint_as_float((u32)raw), LDS_MAX, 0ull)
        PART_APPLY_FLOAT(GAGG_MAX_F64, const double e = u2d(raw), LDS_MAX, 0ull)
        default: break;
      }
    }
  }
  if (nan_acc && P.nan_flag) atomicOr(P.nan_flag, SSGPU_FLAG_NAN_IN_MINMAX);
  if (miss) atomicExch(P.T.overflow + 3, 1u);   // the host widens the ranges and repeats the run
  if constexpr (PLAIN) {
    if (P.hot_only && touched) {
      if ((u32)touched) __hip_atomic_fetch_or(wsum, (u32)touched, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if ((u32)(touched >> 32)) __hip_atomic_fetch_or(wsum + 1, (u32)(touched >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  if (P.slab_segs) {
    // slab mode: every workgroup saw (a slab of) all groups: merge the occupied entries into the global table, one
    // atomic per (group, accumulator word) -- the direct path's end-of-kernel merge
    if (PART_DENSE(P)) {
      // one table of all dense slots: entry e = index e.  Its global slot follows the job-wide layout (partition = index % n_parts,
      // chunk = partition / chunk_parts), so every table of the job is the same array; the packed key is rebuilt from the index
      for (u32 e0 = 0; e0 < C; e0 += SSGPU_PART_THREADS) {
        const u32 e = e0 + t, ec = e < C ? e : 0u;
        const u64 packed = ssgpu_dense_key_of(P.dense, ec);       // (uniform loops: outside the divergent part)
        u32 pp; const u32 q = ssgpu_dense_entry(P.dense, ec, &pp);
        const u32 chunk = pp / P.dense.chunk_parts;
        const bool used = e < C && lkeys[ec] != VM_KEY_EMPTY;
        if (used) {
          const u32 gs = packed == VM_KEY_EMPTY ? P.dense.chunk_slots : (pp - chunk * P.dense.chunk_parts) * P.dense.part_cap + q;
          u64* const keys_c = reinterpret_cast<u64*>(reinterpret_cast<char*>(P.T.keys) + (u64)chunk * P.dense.chunk_stride);
          u64* const acc_c = reinterpret_cast<u64*>(reinterpret_cast<char*>(P.T.acc) + (u64)chunk * P.dense.chunk_stride);
          u32* const cnt_c = reinterpret_cast<u32*>(reinterpret_cast<char*>(P.T.cnt) + (u64)chunk * P.dense.chunk_stride);
          keys_c[gs] = packed == VM_KEY_EMPTY ? 0ull : packed;
          for (u32 s = 0; s < ng; ++s) {
            const u64 v = lacc[(size_t)e * st + s];
            u64* A = &acc_c[(u64)gs * ng + s];
            const u32 op = P.T.merge_op[s];
            if (op == VM_MERGE_ADD_U64) { if (v) atomicAdd(A, v); }
            else if (op == VM_MERGE_MIN_U64) atomicMin(A, v);
            else if (op == VM_MERGE_MAX_U64) atomicMax(A, v);
            else if (op == VM_MERGE_ADD_F64_HI) dd_atomic_add(reinterpret_cast<double*>(A), u2d(v));
            else unsafeAtomicAdd(reinterpret_cast<double*>(A), u2d(v));
            if (P.any_cnt) { const u32 c = lcnt[(size_t)e * st + s]; if (c) atomicAdd(&cnt_c[(u64)gs * ng + s], c); }
          }
        }
      }
      return;
    }
    for (u32 e = t; e <= C; e += SSGPU_PART_THREADS) {
      const u64 key = lkeys[e];
      if (key == VM_KEY_EMPTY) continue;
      u32 gs;
      if (P.hot_only) {   // heavy hitters: entry e = dense slot hot_base + e (every workgroup that saw a row of the key stores the same key).
        // A seeded key that no row of this run carries publishes nothing: the extraction would emit it as a group of zero rows
        if (e == C || !((wsum[(e >> 5) & 1u] >> (e & 31u)) & 1u)) continue;
        gs = P.hot_base + e; P.T.keys[gs] = key;
      }
      else if (e == C) { gs = P.T.capacity_mask + 1u; P.T.keys[gs] = 0ull; }   // the EMPTY-valued key's reserved slot
      else gs = group_insert(P.T, key);
      if (gs == 0xFFFFFFFFu) continue;                                     // global table full: flagged, the host regrows
      for (u32 s = 0; s < ng; ++s) {
        const u64 v = lacc[(size_t)e * st + s];
        u64* A = &P.T.acc[(u64)gs * ng + s];
        const u32 op = P.T.merge_op[s];
        if (op == VM_MERGE_ADD_U64) { if (v) atomicAdd(A, v); }
        else if (op == VM_MERGE_MIN_U64) atomicMin(A, v);
        else if (op == VM_MERGE_MAX_U64) atomicMax(A, v);
        else if (op == VM_MERGE_ADD_F64_HI) dd_atomic_add(reinterpret_cast<double*>(A), u2d(v));
        else unsafeAtomicAdd(reinterpret_cast<double*>(A), u2d(v));
        if (P.any_cnt) { const u32 c = lcnt[(size_t)e * st + s]; if (c) atomicAdd(&P.T.cnt[(u64)gs * ng + s], c); }
      }
    }
    return;
  }
  if (PART_DENSE(P)) {
    // dense slots: entry e of partition `part` is index e * n_parts + part; it lands in slot (part % chunk_parts) * C + e of chunk
    // part / chunk_parts with its packed key rebuilt (an index whose packed key is the EMPTY value goes to the chunk's special slot)
    const u32 chunk = part / P.dense.chunk_parts, pin = part - chunk * P.dense.chunk_parts;
    u64* const keys_c = reinterpret_cast<u64*>(reinterpret_cast<char*>(P.T.keys) + (u64)chunk * P.dense.chunk_stride);
    u64* const acc_c = reinterpret_cast<u64*>(reinterpret_cast<char*>(P.T.acc) + (u64)chunk * P.dense.chunk_stride);
    u32* const cnt_c = reinterpret_cast<u32*>(reinterpret_cast<char*>(P.T.cnt) + (u64)chunk * P.dense.chunk_stride);
    LDS_AS u32* const s_special = wsum + 2;   // (scan scratch, free by now) the entry whose packed key is the EMPTY value, if it was used
    if (t == 0) *s_special = 0xFFFFFFFFu;
    __syncthreads();
    for (u32 e0 = 0; e0 < C; e0 += SSGPU_PART_THREADS) {
      const u32 e = e0 + t, ec = e < C ? e : 0u;
      const u64 packed = ssgpu_dense_key_of(P.dense, ec * P.dense.n_parts + part);
      if (e < C) {
        const bool used = lkeys[e] != VM_KEY_EMPTY;
        const bool special = used && packed == VM_KEY_EMPTY;
        if (special) *s_special = e;
        keys_c[(u64)pin * C + e] = (used && !special) ? packed : VM_KEY_EMPTY;
      }
    }
    for (u32 i = t; i < C * ng; i += SSGPU_PART_THREADS) {
      const u32 l = (i / ng) * st + i % ng;
      acc_c[(u64)pin * C * ng + i] = lacc[l];
      if (P.any_cnt) cnt_c[(u64)pin * C * ng + i] = lcnt[l];
    }
    __syncthreads();
    const u32 se = *s_special;
    if (se != 0xFFFFFFFFu) {
      const u64 sp = (u64)P.dense.chunk_slots;
      if (t == 0) keys_c[sp] = 0ull;
      for (u32 i = t; i < ng; i += SSGPU_PART_THREADS) {
        acc_c[sp * ng + i] = lacc[(size_t)se * st + i];
        if (P.any_cnt) cnt_c[sp * ng + i] = lcnt[(size_t)se * st + i];
      }
    }
    return;
  }
  // dump: local entry e of partition `part` = global slot part * C + e (empty entries stay empty); the reserved
  // entry, if this partition saw the EMPTY-valued key, = the global table's reserved slot
  for (u32 e = t; e < C; e += SSGPU_PART_THREADS) P.T.keys[(u64)part * C + e] = lkeys[e];
  for (u32 i = t; i < C * ng; i += SSGPU_PART_THREADS) {
    const u32 l = (i / ng) * st + i % ng;
    P.T.acc[(u64)part * C * ng + i] = lacc[l];
    if (P.any_cnt) P.T.cnt[(u64)part * C * ng + i] = lcnt[l];
  }
  if (lkeys[C] != VM_KEY_EMPTY) {
    const u64 special = (u64)P.T.capacity_mask + 1ull;
    if (t == 0) P.T.keys[special] = 0ull;
    for (u32 i = t; i < ng; i += SSGPU_PART_THREADS) {
      P.T.acc[special * ng + i] = lacc[(size_t)C * st + i];
      if (P.any_cnt) P.T.cnt[special * ng + i] = lcnt[(size_t)C * st + i];
    }
  }
}

template <int MAXW>
__global__ __launch_bounds__(SSGPU_PART_THREADS, MAXW <= 8 ? 8 : 4) void ssgpu_part_agg_kernel(const PartAggParams P) {
  part_agg_body<MAXW, false, NoRowSource>(P, NoRowSource());
}
template <int MAXW>
__global__ __launch_bounds__(SSGPU_PART_THREADS, 4) void ssgpu_group_resident_kernel(const PartAggParams P, const PlainScatterParams S) {
  part_agg_body<MAXW, true, PlainScatterParams>(P, S);
}

#endif  // !__HIPCC_RTC__ || SSGPU_RTC_PART

#ifndef __HIPCC_RTC__
// ---------------------------------------------------------------------------
// HashJoin index build (see JoinBuildParams in launch.h)
// ---------------------------------------------------------------------------
// The two key words of rhs row i (keys of 65..128 packed bits); false for a NULL key.
__device__ __forceinline__ bool join_wide_key_of_row(const JoinBuildParams& P, u64 i, u64* lo, u64* hi) {
  u64 w[2] = {0ull, 0ull};
  for (u32 k = 0; k < P.n_keys; ++k) {
    if (P.key_nulls[k] && P.key_nulls[k][i]) return false;
    u64 v;
    if (P.width[k] == 8) v = reinterpret_cast<const u64*>(P.key_data[k])[i];
    else if (P.width[k] == 4) v = reinterpret_cast<const u32*>(P.key_data[k])[i];
    else v = reinterpret_cast<const u8*>(P.key_data[k])[i];
    const u64 vmask = P.bits[k] >= 64 ? ~0ull : ((1ull << P.bits[k]) - 1ull);
    w[P.word[k] & 1u] |= (v & vmask) << P.shift[k];
  }
  *lo = w[0]; *hi = w[1];
  return true;
}
__global__ __launch_bounds__(256) void ssgpu_join_build_kernel(const JoinBuildParams P) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= P.n_rows) return;
  if (P.keys_hi) {
    // Two-word keys.  An inserter claims a slot by putting its ROW into rows[slot] (one 32-bit CAS); whoever meets a
    // claimed slot compares its key with the claimant's, recomputed from the (immutable) key columns -- the key words the
    // claimant stores into the table afterwards are only read by the probing kernel, so no ordering between the claim and
    // those stores is needed, and there is no reserved key value.
    u64 lo, hi;
    if (!join_wide_key_of_row(P, i, &lo, &hi)) { if (P.slot_of_row) P.slot_of_row[i] = VM_NONE; return; }
    const bool multi = P.counts != nullptr;
    u32 slot = hash64(lo ^ hash64(hi)) & P.capacity_mask;
    for (u32 probe = 0; probe <= P.capacity_mask; ++probe) {
      const u32 old = atomicCAS(&P.rows[slot], VM_NONE, (u32)i);
      if (old == VM_NONE) {
        P.keys[slot] = lo; P.keys_hi[slot] = hi;
        if (multi) { atomicAdd(&P.counts[slot], 1u); P.slot_of_row[i] = slot; }
        return;
      }
      u64 olo, ohi;
      (void)join_wide_key_of_row(P, old, &olo, &ohi);   // an indexed row: its key is not NULL
      if (olo == lo && ohi == hi) {
        if (multi) { atomicAdd(&P.counts[slot], 1u); P.slot_of_row[i] = slot; }
        else atomicExch(&P.flags[0], 1u);                // duplicate key in a UNIQUE rhs
        return;
      }
      slot = (slot + 1) & P.capacity_mask;
    }
    atomicExch(&P.flags[0], 2u);
    return;
  }
  u64 key = 0;
  for (u32 k = 0; k < P.n_keys; ++k) {
    if (P.key_nulls[k] && P.key_nulls[k][i]) { if (P.slot_of_row) P.slot_of_row[i] = VM_NONE; return; }   // NULL never equals anything: not indexed
    u64 v;
    if (P.width[k] == 8) v = reinterpret_cast<const u64*>(P.key_data[k])[i];
    else if (P.width[k] == 4) v = reinterpret_cast<const u32*>(P.key_data[k])[i];
    else v = reinterpret_cast<const u8*>(P.key_data[k])[i];
    const u64 vmask = P.bits[k] >= 64 ? ~0ull : ((1ull << P.bits[k]) - 1ull);
    key |= (v & vmask) << P.shift[k];
  }
  const bool multi = P.counts != nullptr;
  if (key == VM_KEY_EMPTY) {
    if (multi) {
      const u32 cap = P.capacity_mask + 1u;
      atomicExch(P.special, cap); atomicAdd(&P.counts[cap], 1u); P.slot_of_row[i] = cap;
    } else if (atomicCAS(P.special, VM_NONE, (u32)i) != VM_NONE) atomicExch(&P.flags[0], 1u);
    return;
  }
  // one-word keys: an entry is the pair {key, answer} in ONE 16-byte slot, so that the probe needs one load per slot
  u32 slot = hash64(key) & P.capacity_mask;
  for (u32 probe = 0; probe <= P.capacity_mask; ++probe) {
    const u64 old = atomicCAS(&P.keys[2ull * slot], VM_KEY_EMPTY, key);
    if (multi) {
      if (old == VM_KEY_EMPTY || old == key) {
        if (old == VM_KEY_EMPTY) P.keys[2ull * slot + 1] = slot;   // the probe answers with the key's slot, not a row
        atomicAdd(&P.counts[slot], 1u); P.slot_of_row[i] = slot;
        return;
      }
    } else {
      if (old == VM_KEY_EMPTY) { P.keys[2ull * slot + 1] = i; return; }
      if (old == key) { atomicExch(&P.flags[0], 1u); return; }     // duplicate key in a UNIQUE rhs
    }
    slot = (slot + 1) & P.capacity_mask;
  }
  atomicExch(&P.flags[0], 2u);
}
__global__ void ssgpu_fill_u32_kernel(u32* __restrict__ p, u32 v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// fill helpers (table initialisation without a host round trip)
__global__ void ssgpu_fill_u64_kernel(u64* __restrict__ p, u64 v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
// acc init: per-slot pattern of n_gaggs identities
// Everything a GroupAggregate run needs cleared, in ONE launch (keys -> EMPTY, accumulators -> their initial pattern,
// contribution counts and up to three small word arrays -> 0): these were five or six tiny launches per run, and at the
// shard sizes of a strong-scaling job (a 0.3 ms scan) the launches, not the kernels, are what a step waits for.
__global__ __launch_bounds__(256) void ssgpu_group_init_kernel(const GroupInitParams P) {
  const u64 first = (u64)blockIdx.x * 256 + threadIdx.x, stride = (u64)gridDim.x * 256;
  for (u32 r = 0; r <= P.n_rep; ++r) {   // (the chunks of a dense table buffer: the same three ranges, rep_stride bytes apart)
    u64* const keys = reinterpret_cast<u64*>(reinterpret_cast<char*>(P.keys) + (u64)r * P.rep_stride);
    u64* const acc = reinterpret_cast<u64*>(reinterpret_cast<char*>(P.acc) + (u64)r * P.rep_stride);
    u32* const cnt = reinterpret_cast<u32*>(reinterpret_cast<char*>(P.cnt) + (u64)r * P.rep_stride);
    for (u64 i = first; i < P.n_keys; i += stride) keys[i] = VM_KEY_EMPTY;
    for (u64 i = first; i < P.n_acc; i += stride) acc[i] = P.pattern[i % P.ng];
    for (u64 i = first; i < P.n_cnt; i += stride) cnt[i] = 0u;
  }
  for (int q = 0; q < 4; ++q) for (u64 i = first; i < P.nz[q]; i += stride) P.z[q][i] = 0u;
}
__global__ void ssgpu_fill_pattern_u64_kernel(u64* __restrict__ p, const u64* __restrict__ pattern, u32 plen, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = pattern[i % plen];
}

// ---------------------------------------------------------------------------
// host-callable launchers (C++ linkage inside the library)
// ---------------------------------------------------------------------------
hipError_t ssgpu_launch_pipeline(const VmParams& P, int K, int grid, hipStream_t stream) {
  dim3 g(grid), b(VM_WG_THREADS);  // 4 waves
  size_t lds = P.lds_bytes;
  if (P.uses_math) {
    switch (K) {
      case 1: hipLaunchKernelGGL((ssgpu_pipeline_kernel<1, true>), g, b, lds, stream, P); break;
      case 2: hipLaunchKernelGGL((ssgpu_pipeline_kernel<2, true>), g, b, lds, stream, P); break;
      case 4: hipLaunchKernelGGL((ssgpu_pipeline_kernel<4, true>), g, b, lds, stream, P); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (K) {
    case 1: hipLaunchKernelGGL((ssgpu_pipeline_kernel<1, false>), g, b, lds, stream, P); break;
    case 2: hipLaunchKernelGGL((ssgpu_pipeline_kernel<2, false>), g, b, lds, stream, P); break;
    case 4: hipLaunchKernelGGL((ssgpu_pipeline_kernel<4, false>), g, b, lds, stream, P); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
// workgroups of the pipeline kernel one CU can hold at once for this program (registers and LDS): programs whose tiles
// wait for each other (SEL_RANK_LB) must not launch more workgroups than are resident together
int ssgpu_pipeline_resident_per_cu(const VmParams& P, int K) {
  int n = 0;
  hipError_t e = hipErrorInvalidValue;
  const size_t lds = P.lds_bytes;
  if (P.uses_math) {
    switch (K) {
      case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ssgpu_pipeline_kernel<1, true>, VM_WG_THREADS, lds); break;
      case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ssgpu_pipeline_kernel<2, true>, VM_WG_THREADS, lds); break;
      case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ssgpu_pipeline_kernel<4, true>, VM_WG_THREADS, lds); break;
      default: break;
    }
  } else {
    switch (K) {
      case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ssgpu_pipeline_kernel<1, false>, VM_WG_THREADS, lds); break;
      case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ssgpu_pipeline_kernel<2, false>, VM_WG_THREADS, lds); break;
      case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ssgpu_pipeline_kernel<4, false>, VM_WG_THREADS, lds); break;
      default: break;
    }
  }
  if (e != hipSuccess) { (void)hipGetLastError(); return 1; }
  return n < 1 ? 1 : n;
}
// NOT_UNIQUE hash join, expansion: output row o belongs to the lhs row i whose run [offsets[i],
// offsets[i] + count[i]) contains it (binary search over the exclusive scan of the run counts) and to
// rhs row rows_sorted[run_start[i] + (o - offsets[i])] -- or to no rhs row (LEFT_OUTER, unmatched).
__global__ __launch_bounds__(256) void ssgpu_join_expand_kernel(const u32* __restrict__ offsets, const u32* __restrict__ run_start,
                                                                const u32* __restrict__ rows_sorted, u64 n_lhs, u64 n_out,
                                                                u32* __restrict__ lhs_idx, u32* __restrict__ rhs_row) {
  const u64 o = (u64)blockIdx.x * 256 + threadIdx.x;
  if (o >= n_out) return;
  u64 lo = 0, hi = n_lhs - 1;
  while (lo < hi) {                       // the largest i with offsets[i] <= o (rows with an empty run share
    const u64 mid = (lo + hi + 1) >> 1;   // their successor's offset and are skipped by "largest")
    if ((u64)offsets[mid] <= o) lo = mid; else hi = mid - 1;
  }
  const u32 s = run_start[lo];
  lhs_idx[o] = (u32)lo;
  rhs_row[o] = s == VM_NONE ? VM_NONE : rows_sorted[(u64)s + (o - (u64)offsets[lo])];
}
hipError_t ssgpu_launch_join_expand(const unsigned int* offsets, const unsigned int* run_start, const unsigned int* rows_sorted,
                                    unsigned long long n_lhs, unsigned long long n_out, unsigned int* lhs_idx, unsigned int* rhs_row, hipStream_t stream) {
  if (n_out) hipLaunchKernelGGL(ssgpu_join_expand_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, stream,
                                offsets, run_start, rows_sorted, (u64)n_lhs, (u64)n_out, lhs_idx, rhs_row);
  return hipGetLastError();
}
hipError_t ssgpu_launch_join_build(const JoinBuildParams& P, hipStream_t stream) {
  if (P.n_rows) hipLaunchKernelGGL(ssgpu_join_build_kernel, dim3((unsigned)((P.n_rows + 255) / 256)), dim3(256), 0, stream, P);
  return hipGetLastError();
}
// GroupAggregate FIRST / LAST: the group's accumulator holds the smallest / largest contributing
// row id; the result is the input column's value at that row.
// Value kinds: 0 i32, 1 u32, 2 i64, 3 u64, 4 f32, 5 f64, 6 one byte.  When the result type differs from the column's
// (AddAggregationWithDefinedOutputType) the picked value is converted like the assignment the reference performs
// (AssignmentOperator, aggregation_operators.h:100-122: a C++ conversion; floating -> integer truncates, cvttsd2si's
// INT64_MIN for NaN / out of range).
__global__ __launch_bounds__(256) void ssgpu_gather_rowid_kernel(void* __restrict__ dst, const u8* __restrict__ dst_null, const void* __restrict__ src,
                                                                 u32 width, int src_kind, int dst_kind, const u64* __restrict__ rowids, u64 rowid_mask, i64 row_id_base,
                                                                 const u64* __restrict__ n_rows_dev, u64 n_rows_max) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  const u64 n = n_rows_dev ? *n_rows_dev : n_rows_max;
  if (i >= n || i >= n_rows_max) return;
  const bool isnull = dst_null && dst_null[i];
  const u64 r = isnull ? 0ull : (u64)((i64)(rowids[i] & rowid_mask) - row_id_base);
  if (src_kind == dst_kind) {
    if (width == 8) reinterpret_cast<u64*>(dst)[i] = isnull ? 0ull : reinterpret_cast<const u64*>(src)[r];
    else if (width == 4) reinterpret_cast<u32*>(dst)[i] = isnull ? 0u : reinterpret_cast<const u32*>(src)[r];
    else reinterpret_cast<u8*>(dst)[i] = isnull ? (u8)0 : reinterpret_cast<const u8*>(src)[r];
    return;
  }
  // widen to (i64 | u64 | f64), then narrow to the result type
  i64 vi = 0; u64 vu = 0; double vf = 0.0; int fam = 0;     // 0 signed, 1 unsigned, 2 floating
  if (!isnull) switch (src_kind) {
    case 0: vi = reinterpret_cast<const i32*>(src)[r]; fam = 0; break;
    case 1: vu = reinterpret_cast<const u32*>(src)[r]; fam = 1; break;
    case 2: vi = reinterpret_cast<const i64*>(src)[r]; fam = 0; break;
    case 3: vu = reinterpret_cast<const u64*>(src)[r]; fam = 1; break;
    case 4: vf = (double)reinterpret_cast<const float*>(src)[r]; fam = 2; break;
    case 5: vf = reinterpret_cast<const double*>(src)[r]; fam = 2; break;
    default: vu = reinterpret_cast<const u8*>(src)[r]; fam = 1; break;
  }
  if (dst_kind == 4 || dst_kind == 5) {
    const double d = fam == 2 ? vf : (fam == 1 ? (double)vu : (double)vi);
    if (dst_kind == 5) reinterpret_cast<double*>(dst)[i] = d;
    else reinterpret_cast<float*>(dst)[i] = fam == 2 ? (float)vf : (fam == 1 ? (float)vu : (float)vi);
    return;
  }
  u64 bits;
  if (fam == 2) {
    const double tr = trunc(vf);
    bits = (tr >= -9223372036854775808.0 && tr < 9223372036854775808.0) ? (u64)(i64)tr : 0x8000000000000000ull;   // NaN fails both tests
  } else bits = fam == 1 ? vu : (u64)vi;
  if (dst_kind == 2 || dst_kind == 3) reinterpret_cast<u64*>(dst)[i] = bits;
  else if (dst_kind == 0 || dst_kind == 1) reinterpret_cast<u32*>(dst)[i] = (u32)bits;
  else reinterpret_cast<u8*>(dst)[i] = (u8)bits;
}
hipError_t ssgpu_launch_gather_rowid(void* dst, const uint8_t* dst_null, const void* src, uint32_t width, int src_kind, int dst_kind, const uint64_t* rowids, uint64_t rowid_mask,
                                     int64_t row_id_base, const uint64_t* n_rows_dev, uint64_t n_rows_max, hipStream_t stream) {
  if (n_rows_max) hipLaunchKernelGGL(ssgpu_gather_rowid_kernel, dim3((unsigned)((n_rows_max + 255) / 256)), dim3(256), 0, stream,
                                     dst, dst_null, src, width, src_kind, dst_kind, (const u64*)rowids, (u64)rowid_mask, (i64)row_id_base, (const u64*)n_rows_dev, (u64)n_rows_max);
  return hipGetLastError();
}
hipError_t ssgpu_launch_fill_u32(unsigned int* p, unsigned int v, size_t n, hipStream_t stream) {
  if (n) hipLaunchKernelGGL(ssgpu_fill_u32_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, stream, p, v, n);
  return hipGetLastError();
}
hipError_t ssgpu_launch_part_agg(const PartAggParams& P, unsigned int lds_bytes, hipStream_t stream) {
  if (P.rec_words <= 8) hipLaunchKernelGGL(ssgpu_part_agg_kernel<8>, dim3(P.n_parts), dim3(SSGPU_PART_THREADS), lds_bytes, stream, P);
  else if (P.rec_words <= 16) hipLaunchKernelGGL(ssgpu_part_agg_kernel<16>, dim3(P.n_parts), dim3(SSGPU_PART_THREADS), lds_bytes, stream, P);
  else if (P.rec_words <= SSGPU_PART_MAX_WORDS) hipLaunchKernelGGL(ssgpu_part_agg_kernel<SSGPU_PART_MAX_WORDS>, dim3(P.n_parts), dim3(SSGPU_PART_THREADS), lds_bytes, stream, P);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
hipError_t ssgpu_launch_group_resident(const PartAggParams& P, const PlainScatterParams& S, unsigned int lds_bytes, int grid, hipStream_t stream) {
  if (P.rec_words <= 8) hipLaunchKernelGGL(ssgpu_group_resident_kernel<8>, dim3(grid), dim3(SSGPU_PART_THREADS), lds_bytes, stream, P, S);
  else hipLaunchKernelGGL(ssgpu_group_resident_kernel<16>, dim3(grid), dim3(SSGPU_PART_THREADS), lds_bytes, stream, P, S);
  return hipGetLastError();
}
hipError_t ssgpu_part_agg_set_max_lds(int bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ssgpu_part_agg_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(ssgpu_group_resident_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(ssgpu_group_resident_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(ssgpu_part_agg_kernel<SSGPU_PART_MAX_WORDS>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(ssgpu_part_agg_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
hipError_t ssgpu_pipeline_set_max_lds(int bytes) {
  hipError_t e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(ssgpu_pipeline_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(ssgpu_pipeline_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(ssgpu_pipeline_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(ssgpu_pipeline_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(ssgpu_pipeline_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(ssgpu_pipeline_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return e;
}
hipError_t ssgpu_launch_finish_slots(const VmAccRec* partials, int n_slots, int n_parts, const int* slot_kind,
                                     VmAccRec* out, uint64_t* state, hipStream_t stream, const EmitDesc* descs, int n_out) {
  hipLaunchKernelGGL(ssgpu_finish_slots_kernel, dim3(n_slots), dim3(256), 0, stream, partials, n_slots, n_parts, slot_kind, out, (u64*)state, descs, n_out);
  return hipGetLastError();
}
hipError_t ssgpu_launch_slots_to_state(const VmAccRec* recs, int n_slots, const int* slot_kind, uint64_t* state, hipStream_t stream) {
  int blocks = (n_slots + 63) / 64;
  hipLaunchKernelGGL(ssgpu_slots_to_state_kernel, dim3(blocks), dim3(64), 0, stream, recs, n_slots, slot_kind, (u64*)state);
  return hipGetLastError();
}
// Cross-rank fold of the partial-aggregate state (ssgpu_plan_fold_partials): one thread per slot
// combines the images array by array with the array's operator -- wrapping integer sum (sums,
// counts), double sum (double-double hi / lo), signed min / max (integer extrema and row ids in
// their sign-corrected domain), double min / max.  FIRST / LAST slots take the value of the image
// that holds the smallest / largest contributing row id.
__device__ __forceinline__ void fold_state_slot(const u64* __restrict__ images, int n_images, u64* __restrict__ state, int n_slots, int kind, int s);
__global__ __launch_bounds__(64) void ssgpu_fold_state_kernel(const u64* __restrict__ images, int n_images, u64* __restrict__ state,
                                                              int n_slots, const int* __restrict__ slot_kind) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= n_slots) return;
  fold_state_slot(images, n_images, state, n_slots, slot_kind[s], s);
}
// fold (as above) + state -> slot records + the result row, in ONE launch of one workgroup (ssgpu_plan_fold_finalize): what
// follows the collective of a sharded scalar aggregate is a few hundred bytes of work, and three dependent launches of it
// cost more than the 12.5 M-row scan of an 8-GPU shard is allowed to lose
__device__ __forceinline__ void fold_state_slot(const u64* __restrict__ images, int n_images, u64* __restrict__ state, int n_slots, int kind, int s) {
  const size_t total = (size_t)SSGPU_STATE_ARRAYS * n_slots;
  u64 a[SSGPU_STATE_ARRAYS];
  for (int k = 0; k < SSGPU_STATE_ARRAYS; ++k) a[k] = images[(size_t)k * n_slots + s];
  for (int r = 1; r < n_images; ++r) {
    u64 v[SSGPU_STATE_ARRAYS];
    for (int k = 0; k < SSGPU_STATE_ARRAYS; ++k) v[k] = images[(size_t)r * total + (size_t)k * n_slots + s];
    if (kind == SLOT_FIRST) { if ((i64)v[4] < (i64)a[4]) a[0] = v[0]; }
    else if (kind == SLOT_LAST) { if ((i64)v[5] > (i64)a[5]) a[0] = v[0]; }
    else a[0] += v[0];
    a[1] += v[1];
    a[2] = d2u(u2d(a[2]) + u2d(v[2]));
    a[3] = d2u(u2d(a[3]) + u2d(v[3]));
    a[4] = (i64)v[4] < (i64)a[4] ? v[4] : a[4];
    a[5] = (i64)v[5] > (i64)a[5] ? v[5] : a[5];
    a[6] = u2d(v[6]) < u2d(a[6]) ? v[6] : a[6];
    a[7] = u2d(v[7]) > u2d(a[7]) ? v[7] : a[7];
  }
  for (int k = 0; k < SSGPU_STATE_ARRAYS; ++k) state[(size_t)k * n_slots + s] = a[k];
}
__global__ __launch_bounds__(256) void ssgpu_fold_emit_kernel(const u64* __restrict__ images, int n_images, u64* __restrict__ state, int n_slots,
                                                              const int* __restrict__ slot_kind, VmAccRec* __restrict__ recs,
                                                              const EmitDesc* __restrict__ descs, int n_out) {
  for (int s = threadIdx.x; s < n_slots; s += 256) {
    fold_state_slot(images, n_images, state, n_slots, slot_kind[s], s);
    recs[s] = state_to_slot(state, slot_kind[s], s, n_slots);   // (reads back only what this thread has just written)
  }
  __threadfence_block();
  __syncthreads();
  for (int i = threadIdx.x; i < n_out; i += 256) { const EmitDesc d = descs[i]; emit_scalar_one(d, recs[d.slot]); }
}
hipError_t ssgpu_launch_fold_emit(const uint64_t* images, int n_images, uint64_t* state, int n_slots, const int* slot_kind, VmAccRec* recs,
                                  const EmitDesc* descs, int n_out, hipStream_t stream) {
  hipLaunchKernelGGL(ssgpu_fold_emit_kernel, dim3(1), dim3(256), 0, stream, (const u64*)images, n_images, (u64*)state, n_slots, slot_kind, recs, descs, n_out);
  return hipGetLastError();
}
hipError_t ssgpu_launch_fold_state(const uint64_t* images, int n_images, uint64_t* state, int n_slots, const int* slot_kind, hipStream_t stream) {
  if (n_slots > 0) hipLaunchKernelGGL(ssgpu_fold_state_kernel, dim3((n_slots + 63) / 64), dim3(64), 0, stream, (const u64*)images, n_images, (u64*)state, n_slots, slot_kind);
  return hipGetLastError();
}
hipError_t ssgpu_launch_state_to_slots(const uint64_t* state, int n_slots, const int* slot_kind, VmAccRec* recs, hipStream_t stream) {
  int blocks = (n_slots + 63) / 64;
  hipLaunchKernelGGL(ssgpu_state_to_slots_kernel, dim3(blocks), dim3(64), 0, stream, (const u64*)state, n_slots, slot_kind, recs);
  return hipGetLastError();
}
hipError_t ssgpu_launch_emit_scalar(const VmAccRec* recs, const EmitDesc* descs, int n_out, hipStream_t stream) {
  int blocks = (n_out + 63) / 64;
  hipLaunchKernelGGL(ssgpu_emit_scalar_kernel, dim3(blocks), dim3(64), 0, stream, recs, descs, n_out);
  return hipGetLastError();
}
// large inputs: chunk totals (one workgroup per 8192-entry chunk) -> scan of the totals ->
// per-chunk exclusive scans seeded with their base; small inputs: one workgroup does it all
__global__ __launch_bounds__(1024) void ssgpu_scan_chunk_totals_kernel(const u32* __restrict__ in, int n, u32* __restrict__ totals) {
  __shared__ u64 wave_sums[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i0 = blockIdx.x * 8192 + t * 8;
  u64 sum = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) sum += (i0 + j < n) ? in[i0 + j] : 0u;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
  if (lane == 0) wave_sums[wave] = sum;
  __syncthreads();
  if (t == 0) { u64 s = 0; for (int w = 0; w < 16; ++w) s += wave_sums[w]; totals[blockIdx.x] = (u32)s; }
}
__global__ __launch_bounds__(1024) void ssgpu_scan_chunks_kernel(const u32* __restrict__ in, u32* __restrict__ out, int n,
                                                                  const u32* __restrict__ chunk_base) {
  __shared__ u64 wave_sums[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i0 = blockIdx.x * 8192 + t * 8;
  u32 v[8]; u64 sum = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { v[j] = (i0 + j < n) ? in[i0 + j] : 0u; sum += v[j]; }
  u64 inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { u64 o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  if (lane == 63) wave_sums[wave] = inc;
  __syncthreads();
  u64 run = chunk_base[blockIdx.x] + (inc - sum);
  for (int w = 0; w < wave; ++w) run += wave_sums[w];
#pragma unroll
  for (int j = 0; j < 8; ++j) { if (i0 + j < n) out[i0 + j] = (u32)run; run += v[j]; }
}

hipError_t ssgpu_launch_scan_counts(const uint32_t* in, uint32_t* out, int n, uint64_t* total, hipStream_t stream) {
  if (n <= 65536) {
    hipLaunchKernelGGL(ssgpu_scan_counts_kernel, dim3(1), dim3(1024), 0, stream, in, out, n, (u64*)total);
    return hipGetLastError();
  }
  // scratch for the chunk totals / bases lives behind the output array's last element? no: own static buffers
  static thread_local u32* d_totals = nullptr; static thread_local u32* d_bases = nullptr; static thread_local int cap = 0;
  const int chunks = (n + 8191) / 8192;
  if (chunks > cap) {
    if (d_totals) { (void)hipFree(d_totals); (void)hipFree(d_bases); }
    hipError_t e = hipMalloc((void**)&d_totals, (size_t)chunks * 4); if (e != hipSuccess) return e;
    e = hipMalloc((void**)&d_bases, (size_t)chunks * 4); if (e != hipSuccess) return e;
    cap = chunks;
  }
  hipLaunchKernelGGL(ssgpu_scan_chunk_totals_kernel, dim3(chunks), dim3(1024), 0, stream, in, n, d_totals);
  hipLaunchKernelGGL(ssgpu_scan_counts_kernel, dim3(1), dim3(1024), 0, stream, d_totals, d_bases, chunks, (u64*)total);
  hipLaunchKernelGGL(ssgpu_scan_chunks_kernel, dim3(chunks), dim3(1024), 0, stream, in, out, n, d_bases);
  return hipGetLastError();
}
hipError_t ssgpu_launch_group_count(const GroupExtractParams& P, uint32_t* tile_counts, hipStream_t stream) {
  int blocks = (int)(((size_t)P.capacity + 1 + 511) / 512);
  hipLaunchKernelGGL(ssgpu_group_count_kernel, dim3(blocks), dim3(256), 0, stream, P, tile_counts);
  return hipGetLastError();
}
hipError_t ssgpu_launch_group_extract_lb(const GroupExtractParams& P, unsigned long long* status, uint64_t epoch, unsigned int* ctrl, unsigned int* error_flag, hipStream_t stream) {
  int blocks = (int)(((size_t)P.capacity + 1 + P.extra_slots + 511) / 512);
  hipLaunchKernelGGL(ssgpu_group_extract_lb_kernel, dim3(blocks), dim3(256), 0, stream, P, status, (u64)((epoch & 0x3FFFFFFFull) << 32), ctrl, error_flag);
  return hipGetLastError();
}
hipError_t ssgpu_launch_group_extract(const GroupExtractParams& P, hipStream_t stream) {
  int blocks = (int)(((size_t)P.capacity + 1 + 511) / 512);
  hipLaunchKernelGGL(ssgpu_group_extract_kernel, dim3(blocks), dim3(256), 0, stream, P);
  return hipGetLastError();
}
hipError_t ssgpu_launch_fill_u64(uint64_t* p, uint64_t v, size_t n, hipStream_t stream) {
  int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(ssgpu_fill_u64_kernel, dim3(blocks), dim3(256), 0, stream, (u64*)p, (u64)v, n);
  return hipGetLastError();
}
hipError_t ssgpu_launch_group_init(const GroupInitParams& P, hipStream_t stream) {
  const uint64_t n = std::max<uint64_t>(std::max<uint64_t>(P.n_keys, P.n_acc), std::max<uint64_t>(P.n_cnt, std::max<uint64_t>(std::max<uint64_t>(P.nz[0], P.nz[3]), std::max<uint64_t>(P.nz[1], P.nz[2]))));
  int blocks = (int)std::min<uint64_t>((n + 255) / 256, 4096); if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(ssgpu_group_init_kernel, dim3(blocks), dim3(256), 0, stream, P);
  return hipGetLastError();
}
hipError_t ssgpu_launch_fill_pattern_u64(uint64_t* p, const uint64_t* pattern, uint32_t plen, size_t n, hipStream_t stream) {
  int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(ssgpu_fill_pattern_u64_kernel, dim3(blocks), dim3(256), 0, stream, (u64*)p, (const u64*)pattern, plen, n);
  return hipGetLastError();
}
#endif  // __HIPCC_RTC__
