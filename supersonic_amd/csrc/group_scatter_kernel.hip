// group_scatter_kernel.hip -- the partition scatter of a "plain" GroupAggregate stage as a kernel of its own.
//
// A translation unit of its own because it is compiled WITHOUT -structurizecfg-skip-uniform-regions (the tile VM's
// kernels need that flag for their uniform opcode switch): this kernel's control flow is divergent by nature -- rows
// that a predicate drops sit in random lanes -- and with the flag it produced wrong partitions as soon as a Filter
// dropped rows (uniform loops nested in divergent regions were left unstructurized).
//
// Specialised by runtime compilation for plans that ask for it (rtc.cpp: ssgpu_rtc_specialize_pscat, -DSSGPU_RTC_PSCAT):
// the key packing, the record's field list, the predicates, the record size and the LDS carve-up are constants of
// rtc_pscat.h, every descriptor loop is unrolled and its kernel-argument reads fold away; only the column pointers stay
// run-time values.
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif
#include "vm.h"
#include "launch.h"
#ifdef SSGPU_RTC_PSCAT
#include "rtc_pscat.h"   // kPsNKeys, kPsKey{Width,Shift,Bits,Nullbit}[], kPsNFields, kPsField{Width,Off}[], kPsNPreds, kPsPred{Kind,Cmp,ColLeft}[],
                         // kPsRecWords, kPsRecInv, kPsNParts, kPsRows, kPsLdsBytes
#define PS_NKEYS kPsNKeys
#define PS_KEY_WIDTH(k) kPsKeyWidth[k]
#define PS_KEY_SHIFT(k) kPsKeyShift[k]
#define PS_KEY_BITS(k) kPsKeyBits[k]
#define PS_KEY_NULLBIT(k) kPsKeyNullbit[k]
#define PS_NFIELDS kPsNFields
#define PS_FIELD_WIDTH(f) kPsFieldWidth[f]
#define PS_FIELD_OFF(f) kPsFieldOff[f]
#define PS_NPREDS kPsNPreds
#define PS_PRED_KIND(q) kPsPredKind[q]
#define PS_PRED_CMP(q) kPsPredCmp[q]
#define PS_PRED_COL_LEFT(q) (kPsPredColLeft[q] != 0u)
#define PS_REC_WORDS kPsRecWords
#define PS_REC_INV kPsRecInv
#define PS_NPARTS kPsNParts
#define PS_DENSE (kPsDense != 0u)
#define PS_SPLIT (kPsSplit != 0u)
#define PS_PAY_INV kPsPayInv
#define PS_UNROLL _Pragma("unroll")
#else
#define PS_NKEYS P.n_keys
#define PS_KEY_WIDTH(k) P.keys[k].width
#define PS_KEY_SHIFT(k) P.keys[k].shift
#define PS_KEY_BITS(k) P.keys[k].bits
#define PS_KEY_NULLBIT(k) P.keys[k].nullbit
#define PS_NFIELDS P.n_fields
#define PS_FIELD_WIDTH(f) P.fields[f].width
#define PS_FIELD_OFF(f) P.fields[f].off
#define PS_NPREDS P.n_preds
#define PS_PRED_KIND(q) P.preds[q].kind
#define PS_PRED_CMP(q) P.preds[q].cmp
#define PS_PRED_COL_LEFT(q) (P.preds[q].col_on_left != 0u)
#define PS_REC_WORDS P.rec_words
#define PS_REC_INV P.rec_inv
#define PS_NPARTS P.n_parts
#define PS_DENSE (P.dense.on != 0u)
#define PS_SPLIT (P.split != 0u)
#define PS_PAY_INV P.pay_inv
#define PS_UNROLL
#endif
// Input columns are read once and never again; non-temporal loads (-DPS_NT=1) were meant to keep them from pushing the half-written
// record lines out of the XCD's L2.  Measured (profiles/r06_split_ab.txt): config #3 2.60 -> 2.73 ms, config #4 unchanged: off.
#ifndef PS_NT
#define PS_NT 0
#endif
template <typename T> __device__ __forceinline__ T ps_ld(const T* p) {
#if PS_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int u32;
typedef int i32;
typedef unsigned char u8;
__device__ __forceinline__ double u2d(u64 v) { return __longlong_as_double((i64)v); }
// The hash partition of a packed key.  Any function of the key would be correct (phase 2 only needs equal keys in one
// partition); this is pipeline_kernels.hip's part_of, whose high bits are decorrelated from the in-table home slot.
__device__ __forceinline__ u32 hash_local(u64 key) {
  u32 h = (u32)key ^ ((u32)(key >> 32) * 0x9E3779B1u);
  h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13;
  return h;
}
__device__ __forceinline__ u32 part_of(u64 key, u32 n_parts) {
  u32 h = hash_local(key) * 0x2C1B3C6Du; h ^= h >> 16;
  return __umulhi(h * 0x297A2D39u, n_parts);
}

// ---------------------------------------------------------------------------
// Partitioned GroupAggregate, phase 1 for plain stages (PlainScatterParams in launch.h; reference loop being replaced:
// the per-row insert of cursor/core/aggregate_groups.cc:342-371 -> row_hash_set.cc:458-517).
// Per tile of THREADS x R rows:  (1) every thread loads its rows' predicate and key columns, packs the key exactly as
// KEY_APPEND_* do, and takes a rank inside its partition with one returning LDS atomic; the first 8-byte fields of the
// record are fetched into registers meanwhile;  (2) one global atomic per (partition with rows, tile) reserves the tile's
// run in the segment of (partition, this workgroup's XCD), while wave 0 scans the per-partition counts into staging
// offsets;  (3) records are assembled in LDS in partition order;  (4) the staged words leave with consecutive lanes on
// consecutive 8-byte words: a partition's run is one contiguous piece of its segment.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool pscat_pred(const void* data, const u8* nulls, u64 c, u32 kind, u32 cmp, bool col_on_left, u64 row) {
  const bool is_null = nulls ? nulls[row] != 0 : false;    // a NULL predicate drops the row (filter.cc:170-199)
  bool lt, gt, eq;   // column < constant, column > constant, column == constant
  switch (kind) {
    case 0: { const i32 v = ps_ld(reinterpret_cast<const i32*>(data) + row), k = (i32)(u32)c; lt = v < k; gt = v > k; eq = v == k; } break;
    case 1: { const u32 v = ps_ld(reinterpret_cast<const u32*>(data) + row), k = (u32)c; lt = v < k; gt = v > k; eq = v == k; } break;
    case 2: { const i64 v = ps_ld(reinterpret_cast<const i64*>(data) + row), k = (i64)c; lt = v < k; gt = v > k; eq = v == k; } break;
    case 3: { const u64 v = ps_ld(reinterpret_cast<const u64*>(data) + row), k = c; lt = v < k; gt = v > k; eq = v == k; } break;
    case 4: { const float v = ps_ld(reinterpret_cast<const float*>(data) + row), k = __uint_as_float((u32)c); lt = v < k; gt = v > k; eq = v == k; } break;
    case 6: { const u8 v = ps_ld(reinterpret_cast<const u8*>(data) + row), k = (u8)c; lt = v < k; gt = v > k; eq = v == k; } break;   // a BOOL column as the predicate: (column != FALSE)
    default: { const double v = ps_ld(reinterpret_cast<const double*>(data) + row), k = u2d(c); lt = v < k; gt = v > k; eq = v == k; } break;
  }
  bool r;
  switch (cmp) {
    case 0: r = col_on_left ? lt : gt; break;                  // col < k   |  k < col   (a NaN compares false either way)
    case 1: r = col_on_left ? (lt || eq) : (gt || eq); break;  // col <= k  |  k <= col
    case 2: r = eq; break;
    default: r = !eq; break;
  }
  return r && !is_null;
}

#ifndef PS_PAIR_LOADS
#define PS_PAIR_LOADS 0     /* 1: the pipelined form with two rows per thread loads pairs of consecutive rows (A/B: profiles/r06_pair_loads.txt) */
#endif
#if PS_PAIR_LOADS
#define PS_ROW(base, j, t) ((R) == 2 ? (base) + 2ull * (t) + (u64)(j) : (base) + (u64)(j) * THREADS + (t))
#else
#define PS_ROW(base, j, t) ((base) + (u64)(j) * THREADS + (t))
#endif
#ifdef SSGPU_RTC_PSCAT
// the same predicate on a value that is already in a register (the pipelined form loads a tile's columns one tile ahead)
__device__ __forceinline__ bool pscat_pred_value(u64 raw, bool is_null, u64 c, u32 kind, u32 cmp, bool col_on_left) {
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
__device__ __forceinline__ u64 pscat_load_width(const void* p, u32 width, u64 row) {
  return width == 8u ? ps_ld(reinterpret_cast<const u64*>(p) + row) : width == 4u ? (u64)ps_ld(reinterpret_cast<const u32*>(p) + row) : (u64)ps_ld(reinterpret_cast<const u8*>(p) + row);
}
#endif

template <int R, int NT>
__global__ __launch_bounds__(NT) void ssgpu_part_scatter_plain_kernel(const PlainScatterParams P) {
  constexpr u32 THREADS = NT, T = THREADS * R, REGF = 6;
  const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const u32 NP = PS_NPARTS, cap = P.seg_cap, wpr = PS_REC_WORDS, rb = wpr * 8u;
  const u32 xcd = blockIdx.x & (SSGPU_PSCAT_XCDS - 1u);
  // LDS: per-partition counts of the tile | global base of the tile's run | staging
  // offset of the run | global record index by staging position | the staged records
#ifdef SSGPU_RTC_PSCAT
  __shared__ __attribute__((aligned(16))) char pscat_lds[kPsLdsBytes];   // (a module-loaded kernel cannot ask for > 64 KiB of dynamic LDS)
#else
  extern __shared__ __attribute__((aligned(16))) char pscat_lds[];
#endif
  u32* const cnt = reinterpret_cast<u32*>(pscat_lds);
  u32* const gbase = cnt + NP;
  u32* const start = gbase + NP;
  u32* const grec = start + NP + 2u;   // start[NP] = the tile's record count (+ one word of padding: the records stay 8-byte aligned)
  char* const stage = reinterpret_cast<char*>(grec + T);
  for (u32 i = t; i < NP; i += THREADS) cnt[i] = 0u;
  // the heavy hitters' keys as a small open-addressing set (<= SSGPU_HOT_MAX keys in SSGPU_HOT_SLOTS slots)
  __shared__ u64 hot_tab[SSGPU_HOT_SLOTS];
  if (t < SSGPU_HOT_SLOTS) hot_tab[t] = VM_KEY_EMPTY;
  __syncthreads();
  if (t == 0)
    for (u32 h = 0; h < P.n_hot; ++h) {
      u32 i = hash_local(P.hot_keys[h]) & (SSGPU_HOT_SLOTS - 1u);
      while (hot_tab[i] != VM_KEY_EMPTY && hot_tab[i] != P.hot_keys[h]) i = (i + 1u) & (SSGPU_HOT_SLOTS - 1u);
      hot_tab[i] = P.hot_keys[h];
    }
  __syncthreads();
  const u64 n = P.n_rows, n_tiles = (n + T - 1) / T;
  const u32 nf = PS_NFIELDS;
#ifdef SSGPU_RTC_PSCAT
  if constexpr (kPsPipe != 0u) {
    // The software-pipelined form (specialised builds; profiles/r06_pscat_pipe.txt).  A tile's work is a chain of latencies -- column
    // loads, ranking, one returning global atomic per partition, staging, flush -- and with ONE workgroup per CU (the staging area
    // takes most of the LDS) nothing overlaps them: measured, a tile of 2048 rows takes 10 us whatever its size, half of it waiting.
    // Here tile k + 1's columns are loaded and ranked, and its runs reserved, while tile k is staged and flushed:
    //   [A] gbase <- the reservations issued one trip ago | barrier | stage(k), rank(k + 1) <- the columns loaded one trip ago
    //   [B] barrier | scan + reserve(k + 1) (atomics issued, not waited for), load(k + 2), flush(k)
    // Two counter arrays alternate (rank(k + 1) runs while tile k's counts are being cleared); two barriers per tile instead of three.
    if (n == 0) return;     // (the loads below are unconditional on a row that exists: row 0)
    constexpr u32 NK = kPsNKeys ? kPsNKeys : 1u, NQ = kPsNPreds ? kPsNPreds : 1u, RES = THREADS - 64u, NGB = (kPsNParts + RES - 1u) / RES;
    u32* const cnt_a = cnt; u32* const cnt_b = reinterpret_cast<u32*>(stage + (size_t)T * rb);
    for (u32 i = t; i < NP; i += THREADS) cnt_b[i] = 0u;
    struct Raw { u64 k[R][NK]; u32 kn[R][NK]; u64 p[R][NQ]; u32 pn[R][NQ]; u64 f[R][REGF]; bool in[R]; };
    struct St { u64 key[R]; u32 pt[R], pos[R]; bool ok[R]; u64 fv[R][REGF]; };
    auto issue = [&](u64 tile, Raw& W) {
      const u64 base = tile * T;
#if PS_PAIR_LOADS
      // (-DPS_PAIR_LOADS=1, two rows per thread) a thread takes two CONSECUTIVE rows of the tile, so that an 8-byte column leaves
      // memory as 16 bytes per lane and a 4-byte one as 8; which rows of a tile a thread holds is immaterial to everything downstream
      if constexpr (R == 2) {
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
        typedef u32 u32x2 __attribute__((ext_vector_type(2)));
        const u64 row0 = base + 2ull * t;
        const bool both = tile < n_tiles && row0 + 1ull < n;
        if (both) {
          W.in[0] = W.in[1] = true;
#pragma unroll
          for (u32 q = 0; q < kPsNPreds; ++q) {
            const u32 pw = kPsPredKind[q] == 6u ? 1u : (kPsPredKind[q] == 0u || kPsPredKind[q] == 1u || kPsPredKind[q] == 4u) ? 4u : 8u;
            if (pw == 8u) { const u64x2 v = *reinterpret_cast<const u64x2*>(reinterpret_cast<const u64*>(P.preds[q].data) + row0); W.p[0][q] = v.x; W.p[1][q] = v.y; }
            else if (pw == 4u) { const u32x2 v = *reinterpret_cast<const u32x2*>(reinterpret_cast<const u32*>(P.preds[q].data) + row0); W.p[0][q] = v.x; W.p[1][q] = v.y; }
            else { W.p[0][q] = reinterpret_cast<const u8*>(P.preds[q].data)[row0]; W.p[1][q] = reinterpret_cast<const u8*>(P.preds[q].data)[row0 + 1]; }
            W.pn[0][q] = P.preds[q].nulls ? (u32)P.preds[q].nulls[row0] : 0u; W.pn[1][q] = P.preds[q].nulls ? (u32)P.preds[q].nulls[row0 + 1] : 0u;
          }
#pragma unroll
          for (u32 k = 0; k < kPsNKeys; ++k) {
            if (kPsKeyWidth[k] == 8u) { const u64x2 v = *reinterpret_cast<const u64x2*>(reinterpret_cast<const u64*>(P.keys[k].data) + row0); W.k[0][k] = v.x; W.k[1][k] = v.y; }
            else if (kPsKeyWidth[k] == 4u) { const u32x2 v = *reinterpret_cast<const u32x2*>(reinterpret_cast<const u32*>(P.keys[k].data) + row0); W.k[0][k] = v.x; W.k[1][k] = v.y; }
            else { W.k[0][k] = reinterpret_cast<const u8*>(P.keys[k].data)[row0]; W.k[1][k] = reinterpret_cast<const u8*>(P.keys[k].data)[row0 + 1]; }
            W.kn[0][k] = P.keys[k].nulls ? (u32)P.keys[k].nulls[row0] : 0u; W.kn[1][k] = P.keys[k].nulls ? (u32)P.keys[k].nulls[row0 + 1] : 0u;
          }
#pragma unroll
          for (u32 f = 0; f < REGF; ++f) {
            if (f < nf && PS_FIELD_WIDTH(f < nf ? f : 0u) == 8u && P.fields[f].src) { const u64x2 v = *reinterpret_cast<const u64x2*>(reinterpret_cast<const u64*>(P.fields[f].src) + row0); W.f[0][f] = v.x; W.f[1][f] = v.y; }
            else { W.f[0][f] = 0ull; W.f[1][f] = 0ull; }
          }
          return;
        }
      }
#endif
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const u64 row = PS_ROW(base, j, t);
        W.in[j] = tile < n_tiles && row < n;
        const u64 rowc = W.in[j] ? row : 0ull;     // (unconditional loads, on a row that exists)
#pragma unroll
        for (u32 q = 0; q < kPsNPreds; ++q) {
          W.p[j][q] = pscat_load_width(P.preds[q].data, kPsPredKind[q] == 6u ? 1u : (kPsPredKind[q] == 0u || kPsPredKind[q] == 1u || kPsPredKind[q] == 4u) ? 4u : 8u, rowc);
          W.pn[j][q] = P.preds[q].nulls ? (u32)P.preds[q].nulls[rowc] : 0u;
        }
#pragma unroll
        for (u32 k = 0; k < kPsNKeys; ++k) {
          W.k[j][k] = pscat_load_width(P.keys[k].data, kPsKeyWidth[k], rowc);
          W.kn[j][k] = P.keys[k].nulls ? (u32)P.keys[k].nulls[rowc] : 0u;
        }
#pragma unroll
        for (u32 f = 0; f < REGF; ++f)
          W.f[j][f] = (f < nf && PS_FIELD_WIDTH(f < nf ? f : 0u) == 8u && P.fields[f].src) ? ps_ld(reinterpret_cast<const u64*>(P.fields[f].src) + rowc) : 0ull;
      }
    };
    bool miss = false, over = false;
    auto rank = [&](const Raw& W, St& S, u32* c) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        bool ok = W.in[j];
#pragma unroll
        for (u32 q = 0; q < kPsNPreds; ++q) ok = ok & pscat_pred_value(W.p[j][q], W.pn[j][q] != 0u, P.preds[q].bits, kPsPredKind[q], kPsPredCmp[q], kPsPredColLeft[q] != 0u);
        u64 key = 0ull;
#pragma unroll
        for (u32 k = 0; k < kPsNKeys; ++k) {
          u64 a = W.k[j][k] & (kPsKeyBits[k] >= 64u ? ~0ull : ((1ull << (kPsKeyBits[k] & 63u)) - 1ull));
          if (W.kn[j][k]) a = 1ull << ((kPsKeyNullbit[k] - kPsKeyShift[k]) & 63u);
          key |= a << kPsKeyShift[k];
        }
        if (P.n_hot) {   // (uniform) heavy hitters are aggregated apart
          u32 i = hash_local(key) & (SSGPU_HOT_SLOTS - 1u);
          for (u32 probe = 0; probe < SSGPU_HOT_SLOTS; ++probe) {
            const u64 cur = hot_tab[i];
            if (cur == VM_KEY_EMPTY) break;
            if (cur == key) { ok = false; break; }
            i = (i + 1u) & (SSGPU_HOT_SLOTS - 1u);
          }
        }
        u32 pt;
        if (PS_DENSE) {
          u32 idx;
          const bool in = ssgpu_dense_index(P.dense, key, &idx);
          if (ok && !in) miss = true;
          ok = ok && in;
          const u32 entry = ssgpu_dense_entry(P.dense, idx, &pt);
          key = PS_SPLIT ? (u64)entry : (u64)idx;
        } else pt = part_of(key, NP);
        S.key[j] = key; S.pt[j] = pt; S.ok[j] = ok; S.pos[j] = 0u;
        if (ok) S.pos[j] = atomicAdd(&c[pt], 1u);
#pragma unroll
        for (u32 f = 0; f < REGF; ++f) S.fv[j][f] = W.f[j][f];
      }
    };
    u32 gb[NGB];
    auto scan_reserve = [&](const u32* c) {
      if (wave == 0) {   // staging offsets: exclusive scan of the tile's per-partition counts
        const u32 per = (NP + 63u) / 64u;
        u32 s0 = 0;
        for (u32 i = 0; i < per; ++i) { const u32 q = lane * per + i; if (q < NP) s0 += c[q]; }
        u32 inc = s0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d); if (lane >= (u32)d) inc += o; }
        u32 ex = inc - s0;
        for (u32 i = 0; i < per; ++i) { const u32 q = lane * per + i; if (q < NP) { start[q] = ex; ex += c[q]; } }
        if (lane == 63u) start[NP] = ex;
      } else {           // every partition's run in its (partition, XCD) segment: the atomics leave now, their answers are looked at one trip later
#pragma unroll
        for (u32 g = 0; g < NGB; ++g) {
          const u32 i = t - 64u + g * RES;
          gb[g] = 0u;
          if (i < NP) { const u32 cc = c[i]; if (cc) gb[g] = atomicAdd(&P.counts[i * SSGPU_PSCAT_XCDS + xcd], cc); }
        }
      }
    };
    Raw raw; St cur;
    const u64 tile0 = blockIdx.x, step = gridDim.x;
    issue(tile0, raw);
    rank(raw, cur, cnt_a);
    __syncthreads();
    scan_reserve(cnt_a);
    issue(tile0 + step, raw);
    u32 par = 0u;
    for (u64 tile = tile0; tile < n_tiles; tile += step) {
      if (wave != 0) {
#pragma unroll
        for (u32 g = 0; g < NGB; ++g) { const u32 i = t - 64u + g * RES; if (i < NP) gbase[i] = gb[g]; }
      }
      __syncthreads();                                                   // [A]
      const u32 nrec = start[NP];
      u32* const c_cur = par ? cnt_b : cnt_a; u32* const c_nxt = par ? cnt_a : cnt_b;
      for (u32 i = t; i < NP; i += THREADS) c_cur[i] = 0u;              // (this tile's counts: scanned and reserved one trip ago)
      const u64 base = tile * T;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const u64 row = PS_ROW(base, j, t);
        const bool ok = cur.ok[j];
        const u64 rowc = ok ? row : 0ull;
        const u32 pt = cur.pt[j];
        const u32 s1 = ok ? start[pt] + cur.pos[j] : 0u, g = gbase[pt] + cur.pos[j];
        if (ok) { if (g < cap) grec[s1] = (pt * SSGPU_PSCAT_XCDS + xcd) * cap + g; else { grec[s1] = VM_NONE; over = true; } }
        char* r = stage + (size_t)s1 * rb;
        if (ok) *reinterpret_cast<u64*>(r) = cur.key[j];
#pragma unroll
        for (u32 f = 0; f < REGF; ++f) if (f < nf && PS_FIELD_WIDTH(f < nf ? f : 0u) == 8u) { if (ok) *reinterpret_cast<u64*>(r + PS_FIELD_OFF(f < nf ? f : 0u)) = cur.fv[j][f]; }
#pragma unroll
        for (u32 f = 0; f < nf; ++f) {
          const u32 fw = PS_FIELD_WIDTH(f), fo = PS_FIELD_OFF(f);
          const void* src = P.fields[f].src;
          if (fw == 8u) { if (f >= REGF) { const u64 v = src ? ps_ld(reinterpret_cast<const u64*>(src) + rowc) : 0ull; if (ok) *reinterpret_cast<u64*>(r + fo) = v; } }
          else if (fw == 4u) { const u32 v = src ? ps_ld(reinterpret_cast<const u32*>(src) + rowc) : 0u; if (ok) *reinterpret_cast<u32*>(r + fo) = v; }
          else { const u8 v = src ? ps_ld(reinterpret_cast<const u8*>(src) + rowc) : (u8)0; if (ok) *reinterpret_cast<u8*>(r + fo) = v; }
        }
      }
      St nxt;
      rank(raw, nxt, c_nxt);                                    // (tile + step: its columns were requested one trip ago; past the end: no row counts)
      __syncthreads();                                                   // [B]
      scan_reserve(c_nxt);
      issue(tile + 2u * step, raw);
      const u64* sw = reinterpret_cast<const u64*>(stage);
      if (PS_SPLIT) {
        const u32 pw = wpr - 1u, pwords = nrec * pw;
        for (u32 w = t; w < pwords; w += THREADS) {
          const u32 j = pw == 1u ? w : __umulhi(w, PS_PAY_INV), f = w - j * pw;
          const u32 g = grec[j];
          if (g != VM_NONE) P.recs[(u64)g * pw + f] = sw[j * wpr + 1u + f];
        }
        for (u32 j = t; j < nrec; j += THREADS) {
          const u32 g = grec[j];
          if (g != VM_NONE) P.recs_entry[g] = (unsigned short)sw[j * wpr];
        }
      } else {
        const u32 words = nrec * wpr;
        for (u32 w = t; w < words; w += THREADS) {
          const u32 j = wpr == 1u ? w : __umulhi(w, PS_REC_INV), f = w - j * wpr;
          const u32 g = grec[j];
          if (g != VM_NONE) P.recs[(u64)g * wpr + f] = sw[w];
        }
      }
      cur = nxt; par ^= 1u;
    }
    if (over) atomicExch(P.overflow, 1u);
    if (miss) atomicExch(P.overflow + 2, 1u);
    return;
  }
#endif
  for (u64 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const u64 base = tile * T;
    u64 key[R]; u64 fv[R][REGF]; u32 pt[R], pos[R]; bool ok[R]; bool miss = false;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const u64 row = base + (u64)j * THREADS + t;
      ok[j] = row < n;
      const u64 rowc = ok[j] ? row : 0ull;   // (every load below is unconditional: loops over descriptors stay out of divergent regions)
      PS_UNROLL for (u32 q = 0; q < PS_NPREDS; ++q)
        ok[j] = ok[j] & pscat_pred(P.preds[q].data, P.preds[q].nulls, P.preds[q].bits, PS_PRED_KIND(q), PS_PRED_CMP(q), PS_PRED_COL_LEFT(q), rowc);
      key[j] = 0ull;
      PS_UNROLL for (u32 k = 0; k < PS_NKEYS; ++k) {
        const u32 kw = PS_KEY_WIDTH(k), kbits = PS_KEY_BITS(k), kshift = PS_KEY_SHIFT(k);
        u64 a = kw == 8 ? ps_ld(reinterpret_cast<const u64*>(P.keys[k].data) + rowc) : kw == 4 ? (u64)ps_ld(reinterpret_cast<const u32*>(P.keys[k].data) + rowc)
                                                                                       : (u64)ps_ld(reinterpret_cast<const u8*>(P.keys[k].data) + rowc);
        a &= kbits >= 64 ? ~0ull : ((1ull << kbits) - 1ull);
        if (P.keys[k].nulls && P.keys[k].nulls[rowc]) a = 1ull << (PS_KEY_NULLBIT(k) - kshift);
        key[j] |= a << kshift;
      }
#pragma unroll
      for (u32 f = 0; f < REGF; ++f)   // the leading 8-byte fields travel through registers: their loads are in flight during the ranking
        fv[j][f] = (f < nf && PS_FIELD_WIDTH(f < nf ? f : 0u) == 8u && P.fields[f].src) ? ps_ld(reinterpret_cast<const u64*>(P.fields[f].src) + rowc) : 0ull;
      // heavy hitters are aggregated by the resident kernel (hot_only), not scattered: one key must not fill a partition's segments
      if (P.n_hot) {   // (uniform)
        u32 i = hash_local(key[j]) & (SSGPU_HOT_SLOTS - 1u);
        for (u32 probe = 0; probe < SSGPU_HOT_SLOTS; ++probe) {
          const u64 cur = hot_tab[i];
          if (cur == VM_KEY_EMPTY) break;      // (first: a row whose packed key IS the EMPTY value is never a heavy hitter -- it must not match a free slot)
          if (cur == key[j]) { ok[j] = false; break; }
          i = (i + 1u) & (SSGPU_HOT_SLOTS - 1u);
        }
      }
      if (PS_DENSE) {   // (uniform) dense slots: the record carries the group's dense index, partition = index % NP
        u32 idx;
        const bool in = ssgpu_dense_index(P.dense, key[j], &idx);
        if (ok[j] && !in) miss = true;
        ok[j] = ok[j] && in;
        const u32 entry = ssgpu_dense_entry(P.dense, idx, &pt[j]);
        key[j] = PS_SPLIT ? (u64)entry : (u64)idx;   // (split records: the partition's table entry, 16 bits in an array of its own)
      } else
      pt[j] = part_of(key[j], NP);
      pos[j] = 0u;
      if (ok[j]) pos[j] = atomicAdd(&cnt[pt[j]], 1u);
    }
    __syncthreads();
    if (wave == 0) {   // staging offsets: exclusive scan of the tile's per-partition counts
      const u32 per = (NP + 63u) / 64u;
      u32 s = 0;
      for (u32 i = 0; i < per; ++i) { const u32 q = lane * per + i; if (q < NP) s += cnt[q]; }
      u32 inc = s;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d); if (lane >= (u32)d) inc += o; }
      u32 ex = inc - s;
      for (u32 i = 0; i < per; ++i) { const u32 q = lane * per + i; if (q < NP) { start[q] = ex; ex += cnt[q]; } }
      if (lane == 63u) start[NP] = ex;   // the tile's record count
    } else {           // reserve every partition's run in its (partition, XCD) segment
      for (u32 i = t - 64u; i < NP; i += THREADS - 64u) { const u32 c = cnt[i]; gbase[i] = c ? atomicAdd(&P.counts[i * SSGPU_PSCAT_XCDS + xcd], c) : 0u; }
    }
    __syncthreads();
    bool over = false;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const u64 row = base + (u64)j * THREADS + t;
      const u64 rowc = ok[j] ? row : 0ull;
      const u32 s = ok[j] ? start[pt[j]] + pos[j] : 0u, g = gbase[pt[j]] + pos[j];
      if (ok[j]) { if (g < cap) grec[s] = (pt[j] * SSGPU_PSCAT_XCDS + xcd) * cap + g; else { grec[s] = VM_NONE; over = true; } }
      char* r = stage + (size_t)s * rb;
      if (ok[j]) *reinterpret_cast<u64*>(r) = key[j];
#pragma unroll
      for (u32 f = 0; f < REGF; ++f) if (f < nf && PS_FIELD_WIDTH(f < nf ? f : 0u) == 8u) { if (ok[j]) *reinterpret_cast<u64*>(r + PS_FIELD_OFF(f < nf ? f : 0u)) = fv[j][f]; }
      PS_UNROLL for (u32 f = 0; f < nf; ++f) {
        const u32 fw = PS_FIELD_WIDTH(f), fo = PS_FIELD_OFF(f);
        const void* src = P.fields[f].src;
        if (fw == 8u) { if (f >= REGF) { const u64 v = src ? ps_ld(reinterpret_cast<const u64*>(src) + rowc) : 0ull; if (ok[j]) *reinterpret_cast<u64*>(r + fo) = v; } }
        else if (fw == 4u) { const u32 v = src ? ps_ld(reinterpret_cast<const u32*>(src) + rowc) : 0u; if (ok[j]) *reinterpret_cast<u32*>(r + fo) = v; }
        else { const u8 v = src ? ps_ld(reinterpret_cast<const u8*>(src) + rowc) : (u8)0; if (ok[j]) *reinterpret_cast<u8*>(r + fo) = v; }
      }
    }
    if (over) atomicExch(P.overflow, 1u);
    if (miss) atomicExch(P.overflow + 2, 1u);   // a key outside the dense ranges: the host widens them and repeats the run
    for (u32 i = t; i < NP; i += THREADS) cnt[i] = 0u;   // (read last before the barrier above; next written after the one below)
    __syncthreads();
    const u64* sw = reinterpret_cast<const u64*>(stage);
    if (PS_SPLIT) {
      // split records (dense slots): the payload words of a record -- everything behind its key word -- go to the segment's record
      // array, (wpr - 1) words each, and the key word, which here is the entry of the partition's table (< 2^16), to the segment's
      // array of 16-bit entries: 34 bytes per row for four DOUBLE inputs where the whole record took 40
      const u32 pw = wpr - 1u, nrec = start[NP], pwords = nrec * pw;
      for (u32 w = t; w < pwords; w += THREADS) {
        const u32 j = pw == 1u ? w : __umulhi(w, PS_PAY_INV), f = w - j * pw;
        const u32 g = grec[j];
        if (g != VM_NONE) P.recs[(u64)g * pw + f] = sw[j * wpr + 1u + f];
      }
      for (u32 j = t; j < nrec; j += THREADS) {
        const u32 g = grec[j];
        if (g != VM_NONE) P.recs_entry[g] = (unsigned short)sw[j * wpr];
      }
      continue;   // (as below: no barrier needed before the next tile)
    }
    const u32 words = start[NP] * wpr;
    for (u32 w = t; w < words; w += THREADS) {
      const u32 j = wpr == 1u ? w : __umulhi(w, PS_REC_INV), f = w - j * wpr;   // (one-word records -- the key alone, COUNT(*) queries: 2^32 / 1 + 1 does not fit rec_inv)
      const u32 g = grec[j];
      if (g != VM_NONE) P.recs[(u64)g * wpr + f] = sw[w];
    }
    // (no barrier here: the next tile's staging writes come after two more barriers, which every wave reaches only
    //  after its part of this flush)
  }
}

#ifndef __HIPCC_RTC__
// ---------------------------------------------------------------------------
// max_unique_keys_in_result: the fold of the group table's tail (FoldTailColumn in launch.h; reference:
// cursor/infrastructure/row_hash_set.cc:500-511 -- an unseen key beyond the limit is answered with the set's last row, so
// its rows aggregate there).  The table arrives sorted by first-seen row id; column `blockIdx.x` of the result is its
// rows [0, limit] with rows (limit, n_in) merged into row `limit`.
// ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T fold_combine(T a, T b, u32 op) {
  if (op == 1u) return (T)(a + b);
  if (op == 2u) return b < a ? b : a;     // (a NaN never replaces: `val < result`, aggregation_operators.h:200,221)
  return a < b ? b : a;
}
template <typename T>
__device__ void fold_tail_column(const FoldTailColumn& C, u64 n_in, u64 limit, u64* lds_val, u32* lds_has) {
  const u32 t = threadIdx.x, NT = blockDim.x;
  const T* src = reinterpret_cast<const T*>(C.src);
  T* dst = reinterpret_cast<T*>(C.dst);
  const u64 keep = n_in < limit + 1ull ? n_in : limit + 1ull;
  for (u64 i = t; i < keep; i += NT) { dst[i] = src[i]; if (C.dst_nulls) C.dst_nulls[i] = C.src_nulls ? C.src_nulls[i] : (u8)0; }
  if (n_in <= limit + 1ull || C.op == 0u) return;
  if (C.op >= 4u) {
    // FIRST / LAST: the smallest / largest row id of the merged rows (ids of different groups differ), then its row's value
    const bool first = C.op == 4u;
    u64 best = first ? ~0ull : 0ull; u32 has = 0;
    for (u64 i = limit + t; i < n_in; i += NT) {
      if (C.by_nulls && C.by_nulls[i]) continue;
      const u64 r = C.by[i];
      best = has ? (first ? (r < best ? r : best) : (r > best ? r : best)) : r; has = 1u;
    }
    lds_val[t] = best; lds_has[t] = has;
    __syncthreads();
    for (u32 d = NT >> 1; d > 0; d >>= 1) {
      if (t < d && lds_has[t + d]) {
        const u64 b = lds_val[t + d];
        if (lds_has[t]) { const u64 a = lds_val[t]; lds_val[t] = first ? (b < a ? b : a) : (b > a ? b : a); }
        else { lds_val[t] = b; lds_has[t] = 1u; }
      }
      __syncthreads();
    }
    const u64 pick = lds_val[0]; const u32 any = lds_has[0];
    if (!any) { if (t == 0 && C.dst_nulls) C.dst_nulls[limit] = (u8)1; return; }
    for (u64 i = limit + t; i < n_in; i += NT)
      if (!(C.by_nulls && C.by_nulls[i]) && C.by[i] == pick) { dst[limit] = src[i]; if (C.dst_nulls) C.dst_nulls[limit] = C.src_nulls ? C.src_nulls[i] : (u8)0; }
    return;
  }
  T acc = T(); u32 has = 0;
  for (u64 i = limit + t; i < n_in; i += NT) {
    if (C.src_nulls && C.src_nulls[i]) continue;
    acc = has ? fold_combine<T>(acc, src[i], C.op) : src[i]; has = 1u;
  }
  u64 bits = 0; __builtin_memcpy(&bits, &acc, sizeof(T));
  lds_val[t] = bits; lds_has[t] = has;
  __syncthreads();
  for (u32 d = NT >> 1; d > 0; d >>= 1) {
    if (t < d && lds_has[t + d]) {
      T b; __builtin_memcpy(&b, &lds_val[t + d], sizeof(T));
      if (lds_has[t]) { T a; __builtin_memcpy(&a, &lds_val[t], sizeof(T)); a = fold_combine<T>(a, b, C.op); u64 w = 0; __builtin_memcpy(&w, &a, sizeof(T)); lds_val[t] = w; }
      else { lds_val[t] = lds_val[t + d]; lds_has[t] = 1u; }
    }
    __syncthreads();
  }
  if (t == 0) {
    T r; __builtin_memcpy(&r, &lds_val[0], sizeof(T));
    if (lds_has[0]) dst[limit] = r;
    if (C.dst_nulls) C.dst_nulls[limit] = lds_has[0] ? (u8)0 : (u8)1;
  }
}
__global__ __launch_bounds__(256) void ssgpu_fold_tail_kernel(const FoldTailColumn* cols, u64 n_in, u64 limit) {
  __shared__ u64 lds_val[256];
  __shared__ u32 lds_has[256];
  const FoldTailColumn C = cols[blockIdx.x];
  switch (C.kind) {
    case 0: fold_tail_column<i32>(C, n_in, limit, lds_val, lds_has); break;
    case 1: fold_tail_column<u32>(C, n_in, limit, lds_val, lds_has); break;
    case 2: fold_tail_column<i64>(C, n_in, limit, lds_val, lds_has); break;
    case 3: fold_tail_column<u64>(C, n_in, limit, lds_val, lds_has); break;
    case 4: fold_tail_column<float>(C, n_in, limit, lds_val, lds_has); break;
    case 5: fold_tail_column<double>(C, n_in, limit, lds_val, lds_has); break;
    default: fold_tail_column<u8>(C, n_in, limit, lds_val, lds_has); break;
  }
}
hipError_t ssgpu_launch_fold_tail(const FoldTailColumn* cols_dev, unsigned int n_cols, unsigned long long n_in, unsigned long long limit, hipStream_t stream) {
  if (n_cols == 0) return hipSuccess;
  hipLaunchKernelGGL(ssgpu_fold_tail_kernel, dim3(n_cols), dim3(256), 0, stream, cols_dev, (u64)n_in, (u64)limit);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Heavy-hitter detection (PlainScatterParams: n_hot / hot_keys).  A key that holds a large share of the rows would send
// that share into ONE hash partition: its segments overflow, and the stage used to fall back to the direct shape (every
// cold row a dozen global atomics: 26 ms for 100 M rows with 30 % in one group, against 3 ms for uniform keys).  When a
// partition segment overflows, this kernel looks at a regular sample of the rows -- predicates applied, keys packed exactly as
// the scatter packs them -- counts the keys in an LDS table and reports those seen at least `min_count` times (at most
// SSGPU_HOT_MAX: the threshold is doubled until no more qualify).  The reference's counterpart is the per-row insert of
// cursor/infrastructure/row_hash_set.cc:458-517; which keys repeat is a property of the data, found by looking.
// ---------------------------------------------------------------------------
#define HOT_TABLE 8192u
__global__ __launch_bounds__(1024) void ssgpu_hot_keys_kernel(const PlainScatterParams P, u64 n_sample, u32 min_count, u64* __restrict__ out) {
  __shared__ u64 hkey[HOT_TABLE];
  __shared__ u32 hcnt[HOT_TABLE];
  __shared__ u32 n_out, n_cand;
  const u32 t = threadIdx.x;
  for (u32 i = t; i < HOT_TABLE; i += 1024u) { hkey[i] = VM_KEY_EMPTY; hcnt[i] = 0u; }
  if (t == 0) { n_out = 0u; n_cand = 0u; }
  __syncthreads();
  const u64 n = P.n_rows;
  const u64 stride = n_sample ? (n / n_sample ? n / n_sample : 1ull) : 1ull;
  for (u64 sidx = t; sidx < n_sample; sidx += 1024u) {
    const u64 row = sidx * stride;
    if (row >= n) break;
    bool ok = true;
    for (u32 q = 0; q < P.n_preds; ++q)
      ok = ok & pscat_pred(P.preds[q].data, P.preds[q].nulls, P.preds[q].bits, P.preds[q].kind, P.preds[q].cmp, P.preds[q].col_on_left != 0u, row);
    u64 key = 0ull;
    for (u32 k = 0; k < P.n_keys; ++k) {
      const u32 kw = P.keys[k].width, kbits = P.keys[k].bits, kshift = P.keys[k].shift;
      u64 a = kw == 8 ? reinterpret_cast<const u64*>(P.keys[k].data)[row] : kw == 4 ? (u64)reinterpret_cast<const u32*>(P.keys[k].data)[row]
                                                                                    : (u64)reinterpret_cast<const u8*>(P.keys[k].data)[row];
      a &= kbits >= 64 ? ~0ull : ((1ull << kbits) - 1ull);
      if (P.keys[k].nulls && P.keys[k].nulls[row]) a = 1ull << (P.keys[k].nullbit - kshift);
      key |= a << kshift;
    }
    if (!ok || key == VM_KEY_EMPTY) continue;      // (the EMPTY-valued key has its own reserved slot everywhere: never treated as hot)
    u32 i = hash_local(key) & (HOT_TABLE - 1u);
    for (u32 probe = 0; probe < 64u; ++probe) {    // a full neighbourhood: the row is simply not counted (a sample, not a census)
      const u64 cur = hkey[i];
      if (cur == key) { atomicAdd(&hcnt[i], 1u); break; }
      if (cur == VM_KEY_EMPTY) {
        const u64 prev = atomicCAS(reinterpret_cast<unsigned long long*>(&hkey[i]), (unsigned long long)VM_KEY_EMPTY, (unsigned long long)key);
        if (prev == VM_KEY_EMPTY || prev == key) { atomicAdd(&hcnt[i], 1u); break; }
      }
      i = (i + 1u) & (HOT_TABLE - 1u);
    }
  }
  __syncthreads();
  u32 thr = min_count ? min_count : 1u;
  for (int round = 0; round < 24; ++round) {       // at most SSGPU_HOT_MAX keys: the most frequent ones
    if (t == 0) n_cand = 0u;
    __syncthreads();
    u32 mine = 0;
    for (u32 i = t; i < HOT_TABLE; i += 1024u) mine += hcnt[i] >= thr ? 1u : 0u;
    if (mine) atomicAdd(&n_cand, mine);
    __syncthreads();
    const u32 c = n_cand;
    __syncthreads();
    if (c <= SSGPU_HOT_MAX) break;
    thr *= 2u;
  }
  for (u32 i = t; i < HOT_TABLE; i += 1024u)
    if (hcnt[i] >= thr) { const u32 slot = atomicAdd(&n_out, 1u); if (slot < SSGPU_HOT_MAX) { out[1 + 2 * slot] = hkey[i]; out[2 + 2 * slot] = hcnt[i]; } }
  __syncthreads();
  if (t == 0) out[0] = n_out < SSGPU_HOT_MAX ? n_out : SSGPU_HOT_MAX;
}
hipError_t ssgpu_launch_hot_keys(const PlainScatterParams& S, unsigned long long n_sample, unsigned int min_count, unsigned long long* out, hipStream_t stream) {
  hipLaunchKernelGGL(ssgpu_hot_keys_kernel, dim3(1), dim3(1024), 0, stream, S, (u64)n_sample, min_count, (u64*)out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Value ranges of the key columns (DenseKeyMap in launch.h): one streaming read of the key columns, min / max in an
// order-preserving unsigned domain (signed columns: sign bit flipped), wave reduction, one atomic per wave and word.
// out = [min x n_keys][max x n_keys][non-NULL rows x n_keys], initialised by the launcher.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ssgpu_key_domain_kernel(const PlainScatterParams P, const u32 signed_mask, u64* __restrict__ out) {
  const u64 n = P.n_rows;
  const u32 nk = P.n_keys;
  u64 lo[SSGPU_PSCAT_MAX_KEYS], hi[SSGPU_PSCAT_MAX_KEYS], cnt[SSGPU_PSCAT_MAX_KEYS];
#pragma unroll
  for (u32 k = 0; k < SSGPU_PSCAT_MAX_KEYS; ++k) { lo[k] = ~0ull; hi[k] = 0ull; cnt[k] = 0ull; }
  for (u64 row = (u64)blockIdx.x * 1024u + threadIdx.x; row < n; row += (u64)gridDim.x * 1024u) {
#pragma unroll
    for (u32 k = 0; k < SSGPU_PSCAT_MAX_KEYS; ++k) {
      if (k >= nk) break;
      const u32 kw = P.keys[k].width;
      const bool sgn = (signed_mask >> k) & 1u;
      u64 a = kw == 8 ? reinterpret_cast<const u64*>(P.keys[k].data)[row]
            : kw == 4 ? (sgn ? (u64)(i64)reinterpret_cast<const i32*>(P.keys[k].data)[row] : (u64)reinterpret_cast<const u32*>(P.keys[k].data)[row])
                      : (u64)reinterpret_cast<const u8*>(P.keys[k].data)[row];
      if (sgn) a ^= 0x8000000000000000ull;
      const bool is_null = P.keys[k].nulls && P.keys[k].nulls[row];
      if (!is_null) { lo[k] = a < lo[k] ? a : lo[k]; hi[k] = a > hi[k] ? a : hi[k]; cnt[k] += 1ull; }
    }
  }
  // wave reduction, then the workgroup's 16 waves through LDS: ONE atomic per workgroup and word (one per wave was 49 k atomics on
  // six addresses for 12.5 M rows -- 0.6 ms of a pass that reads 100 MB)
  __shared__ u64 s_lo[16][SSGPU_PSCAT_MAX_KEYS], s_hi[16][SSGPU_PSCAT_MAX_KEYS], s_cnt[16][SSGPU_PSCAT_MAX_KEYS];
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
  for (u32 k = 0; k < SSGPU_PSCAT_MAX_KEYS; ++k) {
    if (k >= nk) break;
    u64 l = lo[k], h = hi[k], c = cnt[k];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const u64 ol = __shfl_xor(l, d), oh = __shfl_xor(h, d), oc = __shfl_xor(c, d);
      l = ol < l ? ol : l; h = oh > h ? oh : h; c += oc;
    }
    if (lane == 0) { s_lo[wave][k] = l; s_hi[wave][k] = h; s_cnt[wave][k] = c; }
  }
  __syncthreads();
  if (threadIdx.x < nk) {
    const u32 k = threadIdx.x;
    u64 l = ~0ull, h = 0ull, c = 0ull;
    for (u32 w = 0; w < 16u; ++w) { l = s_lo[w][k] < l ? s_lo[w][k] : l; h = s_hi[w][k] > h ? s_hi[w][k] : h; c += s_cnt[w][k]; }
    if (c) { atomicMin(&out[k], l); atomicMax(&out[nk + k], h); atomicAdd(&out[2u * nk + k], c); }
  }
}
hipError_t ssgpu_launch_key_domain(const PlainScatterParams& S, const unsigned int* is_signed, unsigned long long* out, int grid, hipStream_t stream) {
  unsigned int mask = 0;
  for (unsigned int k = 0; k < S.n_keys; ++k) if (is_signed[k]) mask |= 1u << k;
  hipError_t e = hipMemsetAsync(out, 0xFF, (size_t)S.n_keys * 8, stream);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(out + S.n_keys, 0, (size_t)S.n_keys * 16, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ssgpu_key_domain_kernel, dim3((unsigned)(grid > 0 ? grid : 1)), dim3(1024), 0, stream, S, mask, (u64*)out);
  return hipGetLastError();
}

unsigned int ssgpu_part_scatter_plain_lds(unsigned int n_parts, unsigned int rec_words, int rows_per_thread, int threads) {
  const unsigned int T = (unsigned)threads * (unsigned)rows_per_thread;
  return (3u * n_parts + 2u + T) * 4u + 16u + T * rec_words * 8u;   // (the pipelined form's second counter array -- n_parts more words behind the staging area -- is added by the caller that asks for it)
}
// The launch shape of the plain scatter: 1024 threads x 2 rows (1 when the staging area of 2048 records does not fit the LDS), one
// workgroup per CU -- or what the caller asks for (development options pscat_threads / pscat_rows / pscat_wgs): 512-thread
// workgroups of 2 - 4 rows per thread, several per CU, overlap one workgroup's load and reservation latencies with another's flush.
PscatGeom ssgpu_part_scatter_plain_geom(unsigned int n_parts, unsigned int rec_words, int threads, int rows, int wgs_per_cu) {
  PscatGeom g; g.pipe = 0u; g.threads = threads == 512 ? 512u : 1024u; g.wgs_per_cu = wgs_per_cu > 0 ? (unsigned)wgs_per_cu : 1u;
  const unsigned int max_rows = g.threads == 512u ? 6u : 3u;
  g.rows = rows > 0 ? (unsigned)rows : 2u;
  if (g.rows > max_rows) g.rows = max_rows;
  if (g.threads == 512u && g.rows < 2u) g.rows = 2u;
  const unsigned int budget = (156u * 1024u) / g.wgs_per_cu;     // (every resident workgroup needs its staging area)
  const unsigned int min_rows = g.threads == 512u ? 2u : 1u;
  while (g.rows > min_rows && ssgpu_part_scatter_plain_lds(n_parts, rec_words, (int)g.rows, (int)g.threads) > budget) --g.rows;
  g.lds = ssgpu_part_scatter_plain_lds(n_parts, rec_words, (int)g.rows, (int)g.threads);
  if (g.lds > 156u * 1024u && g.threads == 512u) { g.threads = 1024u; g.rows = 1u; g.lds = ssgpu_part_scatter_plain_lds(n_parts, rec_words, 1, 1024); }
  return g;
}
template <int R, int NT> static hipError_t pscat_launch(const PlainScatterParams& P, unsigned int lds, int grid, hipStream_t stream) {
  static bool attr_done = false;
  // (the kernel also has 512 bytes of static LDS -- the heavy hitters' key set: the dynamic part may take the rest of the 160 KiB)
  if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ssgpu_part_scatter_plain_kernel<R, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024); attr_done = true; }
  hipLaunchKernelGGL((ssgpu_part_scatter_plain_kernel<R, NT>), dim3((unsigned)grid), dim3(NT), lds, stream, P);
  return hipGetLastError();
}
hipError_t ssgpu_launch_part_scatter_plain(const PlainScatterParams& P, const PscatGeom& g, int grid, hipStream_t stream) {
  if (g.lds > 156u * 1024u) return hipErrorInvalidValue;
  if (g.threads == 1024u && g.rows == 2u) return pscat_launch<2, 1024>(P, g.lds, grid, stream);
  if (g.threads == 1024u && g.rows == 1u) return pscat_launch<1, 1024>(P, g.lds, grid, stream);
  if (g.threads == 512u && g.rows == 2u) return pscat_launch<2, 512>(P, g.lds, grid, stream);
  if (g.threads == 512u && g.rows == 3u) return pscat_launch<3, 512>(P, g.lds, grid, stream);
  if (g.threads == 512u && g.rows == 4u) return pscat_launch<4, 512>(P, g.lds, grid, stream);
  if (g.threads == 512u && g.rows == 6u) return pscat_launch<6, 512>(P, g.lds, grid, stream);
  if (g.threads == 1024u && g.rows == 3u) return pscat_launch<3, 1024>(P, g.lds, grid, stream);
  return hipErrorInvalidValue;
}
#endif  // !__HIPCC_RTC__
