// launch.h -- host-visible launchers and parameter blocks of the HIP kernels.
#ifndef SSGPU_LAUNCH_H_
#define SSGPU_LAUNCH_H_

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#endif
#include "vm.h"

// scalar-aggregate slot combine rules (finish kernel) ------------------------
enum SlotKind {
  SLOT_SUM_INT = 0, SLOT_SUM_DD = 1, SLOT_MIN_U64 = 2, SLOT_MAX_U64 = 3, SLOT_MIN_F64 = 4,
  SLOT_MAX_F64 = 5, SLOT_FIRST = 6, SLOT_LAST = 7, SLOT_COUNT = 8
};
// accumulator-domain -> column-type conversions (emit kernels) ----------------
enum EmitKind {
  EMIT_U64 = 0, EMIT_I64KEY = 1, EMIT_U32 = 2, EMIT_I32KEY = 3, EMIT_F64 = 4, EMIT_F32 = 5,
  EMIT_DD_F64 = 6, EMIT_DD_F32 = 7, EMIT_U8 = 8,
  EMIT_DDRES_F64 = 9, /* (hi, lo) accumulator pair -> the residual e of s = fl(hi + lo): s + e == hi + lo (SSGPU_SUM_RESIDUAL) */
  EMIT_CNT_U64 = 20, EMIT_CNT_U32 = 21, /* COUNT(*) = contribution count of another slot */
  EMIT_FKEY_F64 = 100, EMIT_FKEY_F32 = 101 /* group table: ordered-double key */
};
#define SSGPU_STATE_ARRAYS 8 /* reducible state: 8 u64 arrays of n_slots */

struct EmitDesc { void* data; uint8_t* is_null; int slot; int out_kind; };

struct GroupKeyOut { void* data; uint8_t* is_null; uint32_t shift, bits, nullbit, width; };
struct GroupAggOut { void* data; uint8_t* is_null; int s; int out_kind; int has_cnt; int pad; };
struct GroupExtractParams {
  const unsigned long long* keys;
  const unsigned long long* acc;
  const unsigned int* cnt;
  uint32_t capacity;      /* slots 0..capacity-1 + the special slot `capacity` */
  uint32_t n_gaggs;
  uint32_t n_keys;
  uint32_t n_aggs_out;
  uint32_t extra_slots;   /* occupied-or-empty slots behind the special one: capacity + 1 .. capacity + extra_slots (the heavy hitters' dense slots) */
  const unsigned int* tile_offsets;
  GroupKeyOut keys_out[16];
  GroupAggOut aggs_out[VM_MAX_AGG_SLOTS];
};

// Partitioned GroupAggregate, second phase: one workgroup aggregates one hash partition in LDS
// and dumps its table into slots [part * local_capacity, (part + 1) * local_capacity) of the
// global table (no global atomics; the usual extraction kernels run on the result).  The partition's
// records (rec_words x 8 bytes each, word 0 = packed key) sit in n_segs segments of seg_cap records:
// segment (part, g) at record (part * n_segs + g) * seg_cap holds counts[part * n_segs + g] records.
#define SSGPU_PART_THREADS 1024
// Dense slots ("slot = dense key index"): when the value ranges of a plain stage's key columns are small, a group's
// table slot is a mixed-radix number over them -- idx = sum_k off_k * stride_k, off_k = value - lo_k (the last of a NULLABLE
// key's span_k offsets is its NULL) -- instead of a hash of the packed key.  No key is stored or compared while rows are
// aggregated (a record carries idx where the hashed form carries the packed key), partition = idx % n_parts and LDS entry
// = idx / n_parts are exact, every table of a job -- every run, every shard -- has the SAME slot for a group, so partial
// tables of different ranks combine element by element.  The packed key is rebuilt from idx when a table is dumped.
// A row outside the ranges raises the stage's domain-miss flag (the host widens the ranges and repeats the run).
#define SSGPU_DENSE_MAX_SLOTS (1u << 21)
struct DenseKeyMap {
  unsigned int on, n_keys;
  unsigned int n_parts, parts_inv;    // parts_inv = floor(2^32 / n_parts) + 1
  unsigned int chunk_parts;           // partitions per table chunk: partition p dumps into chunk p / chunk_parts (1 chunk: all of them)
  unsigned int part_cap;              // table entries of a partition (index / n_parts < part_cap)
  unsigned int chunk_slots;           // regular slots of a chunk (chunk_parts * part_cap); its special slot follows them
  unsigned int pad0;
  unsigned long long chunk_stride;    // bytes from a chunk's keys / accumulators / counts to the next chunk's
  struct Key { unsigned long long lo; unsigned int span, stride, shift, bits, nullbit, pad; } keys[8];
};
// widest partition record, in 8-byte words: 8- and 16-word kernels serve the scans; the 20-word build exists for the merge of
// sharded partial tables (key + 16 value columns = 17 words), whose rows all belong to different groups -- through the global
// table that is one atomic per value, in partitions' LDS tables a tenth of it.  Records beyond 16 words take the plain scatter only.
#define SSGPU_PART_MAX_WORDS 20
struct PartAggParams {
  const unsigned long long* recs;
  const unsigned int* counts;
  unsigned int n_segs, seg_cap, rec_words, n_parts;
  unsigned int local_capacity;        // LDS table entries per partition
  unsigned int n_gaggs;               // accumulator words per group
  unsigned int n_aggs, any_cnt;
  unsigned int debug;                 // development switches (perf attribution): 1 = no aggregation, 2 = no probe
  unsigned int slab_segs;             // 0: workgroup = hash partition (its table is dumped to its own slots of T).  > 0: ONE
                                      // partition; workgroup j aggregates segments [j * slab_segs, (j + 1) * slab_segs) -- a slab of
                                      // the rows -- in a table that holds EVERY group, and merges it into T with atomics
  VmGroupTable T;                     // global table: capacity_mask + 1 == n_parts * local_capacity (any number)
  // one packed descriptor per aggregate: GAGG opcode (bits 0-15) | first accumulator word (16-23) | byte offset of
  // the value in the record, 0xFF = none (24-31) | value width (32-39) | byte offset of the NULL flag, 0xFF = never
  // NULL (40-47) | contribution count tracked (48)
  unsigned long long desc[VM_MAX_AGG_SLOTS];
  unsigned int* nan_flag;             // the stage's error word: SSGPU_FLAG_NAN_IN_MINMAX is set when a NaN reaches a floating MIN / MAX
  // Heavy hitters (resident form only): the table is seeded with the row source's hot_keys and takes NO other key -- rows of
  // every other key are skipped (they go through the partition scatter, which in turn skips the hot ones) -- and entry e is
  // merged into global slot hot_base + e (dense slots behind the partitions' ranges) instead of a hashed slot.
  unsigned int hot_only, hot_base;
  DenseKeyMap dense;                  // dense.on: word 0 of a record is the group's dense index, not its packed key
  // split records (PlainScatterParams::split): `recs` holds (rec_words - 1) payload words per record, word 0 -- the entry of this
  // partition's table the record belongs to -- comes from recs_entry
  const unsigned short* recs_entry;
  unsigned int split;
  // dense partitions, the input taken in several row ranges (the aggregation of range k runs beside the scatter of range k + 1): the
  // partition's table starts from what the launch before it dumped into T instead of from the empty table
  unsigned int accumulate;
};
hipError_t ssgpu_launch_part_agg(const PartAggParams& P, unsigned int lds_bytes, hipStream_t stream);
#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
// packed key -> dense index; false: the key lies outside the ranges (idx then addresses slot 0 and must not be used)
__device__ __forceinline__ bool ssgpu_dense_index(const DenseKeyMap& D, unsigned long long key, unsigned int* idx) {
  unsigned int i = 0; bool in = true;
  for (unsigned int k = 0; k < D.n_keys; ++k) {
    const DenseKeyMap::Key K = D.keys[k];
    const unsigned long long mask = K.bits >= 64u ? ~0ull : ((1ull << K.bits) - 1ull);
    unsigned long long off = ((key >> K.shift) - K.lo) & mask;     // value - lo modulo 2^bits: the offset inside the range, in either signedness
    const bool nullable = K.nullbit != 0xFFu;
    const bool is_null = nullable && ((key >> K.nullbit) & 1ull);
    const unsigned int values = nullable ? K.span - 1u : K.span;
    if (is_null) off = K.span - 1u;
    else if (off >= values) { in = false; off = 0; }
    i += (unsigned int)off * K.stride;
  }
  *idx = i;
  return in;
}
__device__ __forceinline__ unsigned long long ssgpu_dense_key_of(const DenseKeyMap& D, unsigned int idx) {
  unsigned long long key = 0ull;
  for (unsigned int k = 0; k < D.n_keys; ++k) {
    const DenseKeyMap::Key K = D.keys[k];
    const unsigned long long mask = K.bits >= 64u ? ~0ull : ((1ull << K.bits) - 1ull);
    const unsigned int off = (idx / K.stride) % K.span;
    if (K.nullbit != 0xFFu && off == K.span - 1u) key |= 1ull << K.nullbit;
    else key |= ((K.lo + off) & mask) << K.shift;
  }
  return key;
}
// idx -> (partition, entry): partition = idx % n_parts, entry = idx / n_parts
__device__ __forceinline__ unsigned int ssgpu_dense_entry(const DenseKeyMap& D, unsigned int idx, unsigned int* part) {
  if (D.n_parts == 1u) { *part = 0u; return idx; }   // (one table of all slots; 2^32 / 1 + 1 does not fit parts_inv)
  unsigned int q = __umulhi(idx, D.parts_inv);
  if (q * D.n_parts > idx) --q;
  *part = idx - q * D.n_parts;
  return q;
}
#endif

// Partitioned GroupAggregate, first phase as a kernel of its own (not a tile-VM program) for "plain" stages: records are
// assembled straight from the input columns.  One 1024-thread workgroup per CU takes tiles of 2048 (1024) rows, ranks
// them by hash partition in LDS, stages the records in partition order and appends every partition's run of the tile to
// the segment of (partition, XCD) with ONE global atomic -- the workgroups of an XCD share a segment, so the lines being
// written at any time are few and complete inside that XCD's L2 instead of being evicted half-written (with one segment
// per (partition, workgroup) the open lines exceed the L2 and the pass runs at the part's scattered-write rate).
#define SSGPU_PSCAT_THREADS 1024
#define SSGPU_PSCAT_XCDS 8
#define SSGPU_PSCAT_MAX_KEYS 8
#define SSGPU_PSCAT_MAX_FIELDS 24
#define SSGPU_PSCAT_MAX_PREDS 4
#define SSGPU_HOT_MAX 32     /* heavy-hitter keys a stage handles apart from the hash partitions */
#define SSGPU_HOT_SLOTS 64   /* entries of the heavy hitters' LDS table = dense global slots reserved for them */
struct PlainScatterParams {
  unsigned long long n_rows;
  unsigned int n_parts, seg_cap, rec_words, rec_inv;   // rec_inv = floor(2^32 / rec_words) + 1
  unsigned int n_keys, n_fields, n_preds;
  unsigned int split;           // dense slots only: records leave as (rec_words - 1) payload words in `recs` + a 16-bit table entry in `recs_entry`
  struct Key { const void* data; const unsigned char* nulls; unsigned int width, shift, bits, nullbit; } keys[SSGPU_PSCAT_MAX_KEYS];
  struct Field { const void* src; unsigned int width, off; } fields[SSGPU_PSCAT_MAX_FIELDS];   // src NULL: an absent NULL mask (zeros)
  struct Pred { const void* data; const unsigned char* nulls; unsigned int kind, cmp, col_on_left, pad; unsigned long long bits; } preds[SSGPU_PSCAT_MAX_PREDS];
  unsigned long long* recs;     // n_parts * SSGPU_PSCAT_XCDS segments of seg_cap records
  unsigned short* recs_entry;   // split: the same segments' table entries, one per record
  unsigned int pay_inv, pad1;   // split: floor(2^32 / (rec_words - 1)) + 1
  unsigned int* counts;         // [n_parts * SSGPU_PSCAT_XCDS] records appended to each segment; zero at launch
  unsigned int* overflow;       // set when a segment ran full
  // heavy hitters: rows whose packed key is one of these are NOT scattered (ssgpu_group_resident_kernel, hot_only, aggregates them)
  unsigned int n_hot, hot_pad;
  unsigned long long hot_keys[SSGPU_HOT_MAX];
  DenseKeyMap dense;            // dense.on: partition and record key word are the group's dense index (overflow[2] = a row outside the ranges)
};
// Value ranges of a plain stage's key columns over ALL rows (predicates not applied: a superset is as good), for DenseKeyMap:
// out[k] = smallest value of key k, out[n_keys + k] = largest -- in an order-preserving UNSIGNED domain (a signed column's,
// is_signed[k], values with their sign bit flipped) -- out[2 n_keys + k] = its rows that are not NULL (0: no value at all).
// out must hold 3 * n_keys words; the launcher initialises them.
hipError_t ssgpu_launch_key_domain(const PlainScatterParams& S, const unsigned int* is_signed, unsigned long long* out, int grid, hipStream_t stream);
// Sharded dense tables (ssgpu_plan_run_dense / ssgpu_plan_fold_dense): every chunk of a table buffer starts with a 64-byte header
// -- [0] flags of the run that filled it (bit 0 table overflow, bit 1 segment overflow, bit 2 a row outside the key ranges),
// [1] the stage's evaluation-error word -- then keys[slots + 1], acc[(slots + 1) * ng], cnt[(slots + 1) * ng] (when counts are kept).
#define SSGPU_DENSE_HEADER 64
struct DenseFoldParams {
  const void* chunks; unsigned int n_chunks, n_gaggs, any_cnt, slots;   // n_chunks images of ONE slot range (slots regular slots + the special one)
  unsigned long long chunk_bytes;
  unsigned long long* keys; unsigned long long* acc; unsigned int* cnt;   // the folded table (slots + 1 entries)
  const unsigned int* merge_op;
  unsigned int* flags_out;      // [0] |= the chunks' flags, [1] |= their error words
  unsigned int* clear[2]; unsigned int n_clear[2];   // words this launch also zeroes (the extraction's control words)
};
hipError_t ssgpu_launch_dense_fold(const DenseFoldParams& P, hipStream_t stream);
// forced_flags != 0: every header gets exactly these flags (a rank whose run failed: 8 | return code << 8) and a zero error word
hipError_t ssgpu_launch_dense_headers(void* chunks, unsigned int n_chunks, unsigned long long chunk_bytes, const unsigned int* overflow4, const unsigned int* error_flag, hipStream_t stream,
                                      unsigned int forced_flags = 0);
// Heavy-hitter detection: one workgroup counts the packed keys of `n_sample` rows taken at a regular stride (predicates
// applied) and reports the keys seen at least `min_count` times -- at most SSGPU_HOT_MAX, the most frequent ones.
// out[0] = number of keys, out[1 + 2 i] = key i, out[2 + 2 i] = its count in the sample.
hipError_t ssgpu_launch_hot_keys(const PlainScatterParams& S, unsigned long long n_sample, unsigned int min_count, unsigned long long* out, hipStream_t stream);
struct PscatGeom { unsigned int threads, rows, wgs_per_cu, lds, pipe; };   // workgroup size, rows per thread (tile = threads x rows), resident workgroups per CU the LDS is sized for, LDS bytes, software-pipelined form (specialised builds)
PscatGeom ssgpu_part_scatter_plain_geom(unsigned int n_parts, unsigned int rec_words, int threads, int rows, int wgs_per_cu);   // 0 = the default of each (1024 threads, 2 rows if they fit, 1 workgroup per CU)
hipError_t ssgpu_launch_part_scatter_plain(const PlainScatterParams& P, const PscatGeom& g, int grid, hipStream_t stream);
// GroupAggregateOptions::max_unique_keys_in_result, last step (group_scatter_kernel.hip): one workgroup per column copies rows
// [0, min(n_in, limit + 1)) and merges every row beyond `limit` into row `limit` (op: 0 keep, 1 sum, 2 min, 3 max; kind:
// 0 i32, 1 u32, 2 i64, 3 u64, 4 f32, 5 f64, 6 one byte), NULL partial results skipped
struct FoldTailColumn {
  const void* src; const unsigned char* src_nulls; void* dst; unsigned char* dst_nulls; unsigned int op, kind;
  const unsigned long long* by; const unsigned char* by_nulls;   // op 4 / 5 (FIRST / LAST): the row-id column the value is picked by
};
hipError_t ssgpu_launch_fold_tail(const FoldTailColumn* cols_dev, unsigned int n_cols, unsigned long long n_in, unsigned long long limit, hipStream_t stream);
unsigned int ssgpu_part_scatter_plain_lds(unsigned int n_parts, unsigned int rec_words, int rows_per_thread, int threads = SSGPU_PSCAT_THREADS);
hipError_t ssgpu_part_agg_set_max_lds(int bytes);
// The LDS-resident form of a plain GroupAggregate stage (few groups: ONE LDS table holds them all): `grid` 1024-thread
// workgroups read the input columns themselves (S: keys, fields, predicates; recs / counts unused), aggregate into a
// table of all groups each and merge it into the global table (A as for the slab form, slab_segs != 0).  No scatter.
hipError_t ssgpu_launch_group_resident(const PartAggParams& A, const PlainScatterParams& S, unsigned int lds_bytes, int grid, hipStream_t stream);

// HashJoin index over the rhs table: packed 64-bit key (same packing as the lhs KEY_APPEND
// instructions) -> rhs row.  Rows with a NULL key are not indexed (they can never match);
// a second row with an already indexed key raises flags[0] (the join was declared UNIQUE).
struct JoinBuildParams {
  const void* key_data[8];
  const unsigned char* key_nulls[8];
  unsigned int width[8], shift[8], bits[8];
  unsigned int n_keys;
  unsigned int capacity_mask;
  unsigned long long n_rows;
  unsigned long long* keys;     // one-word keys: capacity {key, answer} pairs, pre-filled with VM_KEY_EMPTY; two-word keys: first words
  unsigned int* rows;           // two-word keys only
  unsigned int* special;        // pre-filled with VM_NONE
  unsigned int* flags;          // pre-zeroed
  // NOT_UNIQUE keys (both null for a UNIQUE index): the table maps a key to its own slot, counts[slot]
  // (capacity + 1 entries, pre-zeroed; entry `capacity` belongs to the EMPTY-valued key) counts the key's
  // rows and slot_of_row[i] is row i's slot (VM_NONE for a NULL key)
  unsigned int* counts;
  unsigned int* slot_of_row;
  // keys of 65..128 packed bits: word[k] says which of the two key words field k belongs to, keys_hi holds the second
  // word of every entry and `rows` (pre-filled with VM_NONE) is what an inserter claims (see ssgpu_join_build_kernel)
  unsigned int word[8];
  unsigned long long* keys_hi;
};
hipError_t ssgpu_launch_join_expand(const unsigned int* offsets, const unsigned int* run_start, const unsigned int* rows_sorted,
                                    unsigned long long n_lhs, unsigned long long n_out, unsigned int* lhs_idx, unsigned int* rhs_row, hipStream_t stream);
hipError_t ssgpu_launch_join_build(const JoinBuildParams& P, hipStream_t stream);
hipError_t ssgpu_launch_fill_u32(unsigned int* p, unsigned int v, size_t n, hipStream_t stream);
// one launch for all a GroupAggregate run clears: keys[n_keys] = EMPTY, acc[i] = pattern[i % ng] (n_acc words), cnt[n_cnt] = 0,
// z[q][0 .. nz[q]) = 0 (32-bit words; unused: nz = 0)
struct GroupInitParams {
  unsigned long long* keys; unsigned long long n_keys;
  unsigned long long* acc; const unsigned long long* pattern; unsigned int ng; unsigned long long n_acc;
  unsigned int* cnt; unsigned long long n_cnt;
  unsigned int* z[4]; unsigned long long nz[4];
  unsigned int n_rep, pad; unsigned long long rep_stride;   // keys / acc / cnt ranges repeated n_rep MORE times, rep_stride bytes apart (the chunks of a dense table buffer)
};
hipError_t ssgpu_launch_group_init(const GroupInitParams& P, hipStream_t stream);

hipError_t ssgpu_launch_pipeline(const VmParams& P, int K, int grid, hipStream_t stream);
int ssgpu_pipeline_resident_per_cu(const VmParams& P, int K);
#ifndef __HIPCC_RTC__
#include <string>
// rtc.cpp: kernels specialised by runtime compilation.  The specialize calls return a HANDLE to a cached, reference-counted
// kernel (NULL + *why: keep the interpreter / the generic kernel); ssgpu_rtc_release drops the reference and the module
// is unloaded with the last one.  static_lds > 0: the pipeline kernel's LDS is a static array of that size (launches of
// more than 64 KiB of LDS, which a module-loaded kernel cannot get dynamically).
void* ssgpu_rtc_specialize(int device, const VmInstr* prog, int n_instr, int K, bool math, const uint32_t* staged_width, const uint32_t* staged_lds_off,
                           int n_staged, uint32_t static_lds, std::string* why);
hipError_t ssgpu_launch_pipeline_rtc(void* handle, const VmParams& P, int grid, bool static_lds, hipStream_t stream);
void* ssgpu_rtc_specialize_part_agg(int device, const unsigned long long* desc, int n_aggs, unsigned int rec_words, unsigned int n_gaggs, bool any_cnt,
                                    unsigned int lds_bytes, std::string* why, const PlainScatterParams* source = nullptr,   // source: the resident form (reads the input columns)
                                    bool dense = false,                                                                     // dense: records carry dense indices (DenseKeyMap), no probe
                                    bool split = false,                                                                     // split: records arrive as payload words + 16-bit table entries
                                    bool prefetch = false);                                                                 // prefetch: the record form loads a trip ahead (narrow records)
hipError_t ssgpu_launch_group_resident_rtc(void* handle, const PartAggParams& A, const PlainScatterParams& S, int grid, hipStream_t stream);
hipError_t ssgpu_launch_part_agg_rtc(void* handle, const PartAggParams& P, hipStream_t stream);
void* ssgpu_rtc_specialize_pscat(int device, const PlainScatterParams& S, const PscatGeom& g, std::string* why);
hipError_t ssgpu_launch_part_scatter_plain_rtc(void* handle, const PlainScatterParams& P, const PscatGeom& g, int grid, hipStream_t stream);
void* ssgpu_rtc_function(void* handle);
void ssgpu_rtc_release(void* handle);
void ssgpu_rtc_cached_only(bool on);   // the calling thread's requests from now on: find a kernel (memory, disk) or fail -- never compile
void ssgpu_rtc_mode(int mode);         // ... in general: 0 compile what is missing now, 1 never compile, 2 leave what is missing to the worker thread
int ssgpu_rtc_current_mode();         // the calling thread's mode (a slot remembers the mode it missed under: a stronger one asks again)
bool ssgpu_rtc_pending();              // the calling thread's last request came back empty-handed because its kernel is being compiled
void ssgpu_rtc_trim(int keep);   // unloads kernels without a user down to `keep` of them
void ssgpu_rtc_stats(long long* modules, long long* code_bytes, long long* compilations, long long* disk_hits);
#endif
hipError_t ssgpu_pipeline_set_max_lds(int bytes);
hipError_t ssgpu_launch_finish_slots(const VmAccRec* partials, int n_slots, int n_parts, const int* slot_kind,
                                     VmAccRec* out, uint64_t* state /* NULL, or the reducible state of a partial run */, hipStream_t stream,
                                     const EmitDesc* descs = nullptr, int n_out = 0 /* descs: the result columns are emitted by the same launch */);
hipError_t ssgpu_launch_slots_to_state(const VmAccRec* recs, int n_slots, const int* slot_kind, uint64_t* state,
                                       hipStream_t stream);
hipError_t ssgpu_launch_fold_state(const uint64_t* images, int n_images, uint64_t* state, int n_slots, const int* slot_kind, hipStream_t stream);
hipError_t ssgpu_launch_fold_emit(const uint64_t* images, int n_images, uint64_t* state, int n_slots, const int* slot_kind, VmAccRec* recs,
                                  const EmitDesc* descs, int n_out, hipStream_t stream);
hipError_t ssgpu_launch_state_to_slots(const uint64_t* state, int n_slots, const int* slot_kind, VmAccRec* recs,
                                       hipStream_t stream);
hipError_t ssgpu_launch_emit_scalar(const VmAccRec* recs, const EmitDesc* descs, int n_out, hipStream_t stream);
hipError_t ssgpu_launch_scan_counts(const uint32_t* in, uint32_t* out, int n, uint64_t* total, hipStream_t stream);
hipError_t ssgpu_launch_group_count(const GroupExtractParams& P, uint32_t* tile_counts, hipStream_t stream);
// count + scan + extract in one launch (decoupled look-back over ticket-ordered 512-slot tiles); ctrl = [rows u64][ticket u32][gave-up u32], zero at launch
hipError_t ssgpu_launch_group_extract_lb(const GroupExtractParams& P, unsigned long long* status, uint64_t epoch, unsigned int* ctrl, unsigned int* error_flag, hipStream_t stream);
hipError_t ssgpu_launch_group_extract(const GroupExtractParams& P, hipStream_t stream);
hipError_t ssgpu_launch_fill_u64(uint64_t* p, uint64_t v, size_t n, hipStream_t stream);
hipError_t ssgpu_launch_fill_pattern_u64(uint64_t* p, const uint64_t* pattern, uint32_t plen, size_t n,
                                         hipStream_t stream);

// sort / clusters (sort_kernels.hip)
hipError_t ssgpu_launch_sort_iota(uint32_t* idx, uint64_t n, hipStream_t s);
hipError_t ssgpu_launch_sort_load_keys(uint64_t* keys, const uint32_t* idx, const void* col, const uint8_t* nulls, uint32_t width,
                                       int kind, int descending, int null_pass, uint64_t n, unsigned long long* bits, hipStream_t s);
uint32_t ssgpu_sort_tiles(uint64_t n);
hipError_t ssgpu_launch_sort_hist(const uint64_t* keys, uint32_t shift, uint64_t n, uint32_t* hist, hipStream_t s);
hipError_t ssgpu_launch_sort_scatter(const uint64_t* keys_in, const uint32_t* idx_in, uint64_t* keys_out, uint32_t* idx_out,
                                     uint32_t shift, uint64_t n, const uint32_t* offsets, hipStream_t s);
// one-sweep radix passes + record payload (sort_kernels.hip)
hipError_t ssgpu_launch_sort_load_keys_hist(uint64_t* keys, const uint32_t* idx, const void* col, const uint8_t* nulls, uint32_t width, int kind,
                                            int descending, int null_pass, uint64_t n, unsigned long long* bits, uint32_t* hist8, uint32_t* base8, hipStream_t s,
                                            uint64_t* compact_out = nullptr);
hipError_t ssgpu_launch_sort_onesweep(const uint64_t* keys_in, const uint32_t* idx_in, uint64_t* keys_out, uint32_t* idx_out, uint32_t shift, uint64_t n,
                                      const uint32_t* digit_base, unsigned long long* status, uint32_t* ticket, uint64_t epoch, uint32_t* stuck, hipStream_t s);
hipError_t ssgpu_launch_sort_fix_ties(uint64_t* keys, uint32_t* idx, uint64_t n, uint32_t hi_shift, uint32_t* too_long, hipStream_t s);
struct SortRecField { const void* src; void* dst; unsigned int off, width; };   // src: input column (pack), dst: output column (gather; null = not wanted)
#define SSGPU_SORT_MAX_FIELDS 96
struct SortRecParams { void* recs; unsigned long long n; unsigned int stride, n_fields; SortRecField fields[SSGPU_SORT_MAX_FIELDS]; };
hipError_t ssgpu_launch_sort_pack(const SortRecParams& P, hipStream_t s);
hipError_t ssgpu_launch_sort_gather_rec(const SortRecParams& P, const uint32_t* idx, uint32_t idx_stride, hipStream_t s);
hipError_t ssgpu_launch_sort_fix_ties_compact(uint64_t* kc, const uint64_t* keys, uint64_t n, uint32_t* too_long, hipStream_t s);
hipError_t ssgpu_launch_sort_extract_idx(uint32_t* idx, const uint64_t* kc, uint64_t n, hipStream_t s);
// View-file loader: one piece = one column (or NULL-mask) segment of one file chunk inside a staged slab
struct UnpackPiece { unsigned long long src_off; void* dst; unsigned long long bytes; };
hipError_t ssgpu_launch_unpack(const char* slab, const UnpackPiece* pieces, unsigned int n_pieces, hipStream_t s);
hipError_t ssgpu_launch_sort_unkey(void* out, const uint64_t* keys, uint32_t width, int kind, int descending, uint64_t n, hipStream_t s);
hipError_t ssgpu_launch_sort_gather(void* out, uint8_t* out_nulls, const void* col, const uint8_t* nulls, uint32_t width,
                                    const uint32_t* idx, uint64_t n, hipStream_t s);
hipError_t ssgpu_launch_cluster_count(const void* const* data, const uint8_t* const* nulls, const uint32_t* width, uint32_t nkeys,
                                      uint64_t n, uint32_t* tile_counts, hipStream_t s);
hipError_t ssgpu_launch_cluster_flags(const void* const* data, const uint8_t* const* nulls, const uint32_t* width, uint32_t nkeys,
                                      uint64_t n, uint8_t* flag, hipStream_t s);
hipError_t ssgpu_launch_cluster_assign(const void* const* data, const uint8_t* const* nulls, const uint32_t* width, uint32_t nkeys,
                                       void* const* out_data, uint8_t* const* out_nulls, uint64_t n, const uint32_t* tile_offsets,
                                       uint32_t* seg_id, hipStream_t s);
hipError_t ssgpu_launch_gather_rowid(void* dst, const uint8_t* dst_null, const void* src, uint32_t width, int src_kind, int dst_kind, const uint64_t* rowids,
                                     uint64_t rowid_mask /* bits of a rowids word that ARE the row id */, int64_t row_id_base, const uint64_t* n_rows_dev, uint64_t n_rows_max, hipStream_t stream);
// SUM of a floating column into an integer result, row after row in the rows' order (sort_kernels.hip: SeqSumParams); one result
// per segment of seg_id (nullptr: one segment of all n rows); kinds 0 i32, 1 u32, 2 i64, 3 u64, 4 f32, 5 f64
// (one segment only) first / state / resume: rows [first, n) continue the fold from state[0..1] -- the host cuts a long fold into
// launches and looks at the plan's interrupt flag between them
hipError_t ssgpu_launch_seq_sum(const void* src, const uint8_t* src_nulls, int src_kind, const uint32_t* seg_id, uint64_t n,
                                void* dst, uint8_t* dst_nulls, int dst_kind, hipStream_t s, uint64_t first = 0, uint64_t* state = nullptr, int resume = 0);
// first-seen ranks of key clusters (sort_kernels.hip): DISTINCT aggregates under max_unique_keys_in_result
hipError_t ssgpu_launch_seg_first(const uint32_t* seg_id, const uint64_t* rowid, uint64_t n, uint32_t* first, hipStream_t s);
hipError_t ssgpu_launch_rank_scatter(const uint32_t* sorted, uint64_t n_seg, uint32_t* rank, hipStream_t s);
hipError_t ssgpu_launch_rank_rows(const uint32_t* seg_id, const uint32_t* rank, uint64_t n, uint32_t limit, uint32_t* out_rank, uint8_t* out_own, hipStream_t s);
hipError_t ssgpu_launch_dense_extract(const uint64_t* acc, const uint32_t* cnt, uint32_t n_gaggs, uint64_t n_rows,
                                      const GroupAggOut* outs, uint32_t n_out, hipStream_t s);

// result images (exchange_kernels.hip): one piece = one column's data or NULL mask
struct ImagePiece { const void* src; unsigned long long image_off; unsigned long long unpacked_off; unsigned int width; unsigned int pad; };
#define SSGPU_IMAGE_MAX_PIECES 92   /* kernel arguments stay under 4 KiB */
#define SSGPU_IMAGE_HEADER 64
struct ImagePackParams {
  void* image;
  const unsigned long long* rows_dev;   // device row count of the result, or null: rows_host
  unsigned long long rows_host, capacity;
  unsigned int n_pieces, n_flags;
  const unsigned int* error_flags[8];   // evaluation-error words of the plan's stages (header word 4 = their OR)
  const unsigned int* retry_flags[4];   // table / segment overflow words not yet seen by the host (header word 2 |= any set)
  unsigned int n_retry, pad2;
  ImagePiece pieces[SSGPU_IMAGE_MAX_PIECES];
};
struct ImageUnpackParams {
  const void* images; void* unpacked;
  unsigned long long image_bytes, capacity, valid_off, trailer_off;
  unsigned int n_pieces, n_images;
  ImagePiece pieces[SSGPU_IMAGE_MAX_PIECES];
};
hipError_t ssgpu_launch_pack_image(const ImagePackParams& P, hipStream_t s);
// Key-range exchange: the result's rows routed into n_dest images by a hash of their first n_keys columns (the group keys),
// image d at images + d * image_bytes.  counters: n_dest device words, zero at launch (rows appended to each image).
struct ImageRoutePieces { unsigned int n_keys, n_dest; unsigned long long image_bytes; unsigned int* counters; unsigned int key_piece[16]; int key_null_piece[16]; };
hipError_t ssgpu_launch_route_images(const ImagePackParams& P, const ImageRoutePieces& R, hipStream_t s);
hipError_t ssgpu_launch_unpack_images(const ImageUnpackParams& P, hipStream_t s);

#endif  // SSGPU_LAUNCH_H_
