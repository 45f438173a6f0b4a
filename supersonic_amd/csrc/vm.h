// vm.h -- the column-tile VM shared by the host lowering pass and the HIP kernels.
//
// Execution model (MI355X-first restatement of Supersonic's block-at-a-time
// evaluation, supersonic/expression/base/expression.cc:57-76 and
// supersonic/expression/vector/vector_primitives.h:99-105): where the reference
// keeps a 1024-row block hot in the CPU's L1 and walks a tree of virtual
// DoEvaluate() calls, we keep a TILE of rows resident in a workgroup's LDS and
// run a flat, wave-uniform instruction list over it.  Every VM "register" is a
// typed array of TILE rows in LDS; input columns reach their registers through
// a per-lane register prefetch file (the next tile's loads are in flight while
// the program runs over this one), intermediates never leave the CU, and only
// sink instructions (aggregate / store / group) touch HBM again.
//
// Thread <-> row mapping inside a tile (256 threads, K = tile_rows / 512):
//   thread t, step k owns the row PAIR  p = k*256 + t  ->  rows 2p, 2p+1.
//   8-byte registers: one ds_read_b128 at p*16;  4-byte: ds_read_b64 at p*8;
//   1-byte (BOOL / null masks): ds_read_u16 at p*2.  Lanes are contiguous, so
//   every LDS access is conflict-free and no barrier is needed between
//   elementwise instructions (a thread only ever touches its own pairs).
#ifndef SSGPU_VM_H_
#define SSGPU_VM_H_

#ifdef __HIPCC_RTC__   /* runtime compilation (rtc.cpp): no host headers */
typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;
typedef int int32_t; typedef unsigned int uint32_t; typedef long int64_t; typedef unsigned long uint64_t; typedef unsigned long uintptr_t;
typedef int hipError_t; typedef struct ihipStream_t* hipStream_t;
#else
#include <stdint.h>
#endif

#define VM_NONE 0xFFFFFFFFu
/* functions of MATH1_F64 / MATH2_F64 (expression/core/math_evaluators.h:92-204: the libm calls of the reference) */
enum { VM_MATH_EXP = 1, VM_MATH_LN, VM_MATH_LOG10, VM_MATH_LOG2, VM_MATH_SIN, VM_MATH_COS, VM_MATH_TAN, VM_MATH_ASIN, VM_MATH_ACOS,
       VM_MATH_ATAN, VM_MATH_SINH, VM_MATH_COSH, VM_MATH_TANH, VM_MATH_ASINH, VM_MATH_ACOSH, VM_MATH_ATANH, VM_MATH_POW, VM_MATH_ATAN2 };
#ifndef VM_THREADS
#define VM_THREADS 256      /* threads per workgroup */
#endif
#ifndef VM_PF_UNITS
#define VM_PF_UNITS 8       /* 16-byte-per-lane units of the register prefetch file */
#endif
#ifndef VM_WAVES_PER_EU
#define VM_WAVES_PER_EU 4   /* occupancy target of the pipeline kernel (sets its VGPR budget) */
#endif
#ifndef VM_STAGE_PRIO
#define VM_STAGE_PRIO 1     /* s_setprio level of a wave while it stages a tile (0 = same as the program) */
#endif
#define VM_COMPUTE_THREADS VM_THREADS
#define VM_WG_THREADS VM_THREADS
#define VM_WAVES (VM_THREADS / 64)       /* waves per workgroup */
#define VM_TILE_UNIT (2 * VM_THREADS)    /* rows of a K = 1 tile: one row pair per thread */
#define VM_FAST_SLOTS 8     /* aggregate slots with per-lane register accumulators */
#define VM_MAX_STAGED 80     /* input arrays (columns + NULL masks) one pipeline stages; ssgpu_plan_create refuses more.  VmParams stays under the 4 KiB kernel-argument block */
#define VM_MAX_OUTPUTS 64
#define VM_MAX_AGG_SLOTS 64
#define VM_ACC_STRIDE (VM_WAVES * 32) /* bytes of LDS per aggregate slot: one 32 B record per wave */

// type suffixes: I32 I64 U32 U64 F32 F64 (B8 = BOOL byte)
#define VM_OPS(X)                                                              \
  X(NOP)                                                                       \
  /* ---- arithmetic: dst = a op b (supersonic operators.h:68-86) ---------- */\
  X(ADD_I32) X(ADD_I64) X(ADD_F32) X(ADD_F64)                                  \
  X(SUB_I32) X(SUB_I64) X(SUB_F32) X(SUB_F64)                                  \
  X(MUL_I32) X(MUL_I64) X(MUL_F32) X(MUL_F64)                                  \
  X(DIV_F32) X(DIV_F64)                                                        \
  X(CDIV_I32) X(CDIV_I64) X(CDIV_U32) X(CDIV_U64)                              \
  X(MOD_I32) X(MOD_I64) X(MOD_U32) X(MOD_U64)                                  \
  X(NEG_I32) X(NEG_I64) X(NEG_F32) X(NEG_F64)                                  \
  /* ---- exact math (expression/core/math_evaluators.h:82-146,206-220) ----- */\
  X(ABS_I32) X(ABS_I64) X(ABS_F32) X(ABS_F64)                                  \
  X(ROUND_F32) X(ROUND_F64) X(CEIL_F32) X(CEIL_F64) X(FLOOR_F32) X(FLOOR_F64)  \
  X(TRUNC_F32) X(TRUNC_F64)                                                    \
  X(CEIL2I_F32) X(CEIL2I_F64) X(FLOOR2I_F32) X(FLOOR2I_F64) /* -> I64 */       \
  X(SQRT_F64)                                                                  \
  X(ISFINITE_F64) X(ISNAN_F64) X(ISINF_F64) X(ISNORMAL_F64) /* -> B8 */        \
  X(ISODD_32) X(ISODD_64) /* -> B8; IS_EVEN = NOT */                           \
  X(FAIL_TRUE_8) /* a = null mask, b = BOOL flags, c = selection: evaluation error where set */ \
  /* ---- bitwise ---------------------------------------------------------- */\
  X(BAND_32) X(BAND_64) X(BOR_32) X(BOR_64) X(BXOR_32) X(BXOR_64)              \
  X(BANDNOT_32) X(BANDNOT_64) X(BNOT_32) X(BNOT_64)                            \
  X(SHL_I32) X(SHL_I64) X(SHR_I32) X(SHR_I64) X(SHR_U32) X(SHR_U64)            \
  /* ---- comparisons: dst(B8) = a op b (operators.h:185-296) --------------- */\
  X(LT_I32) X(LT_I64) X(LT_U32) X(LT_U64) X(LT_F32) X(LT_F64) X(LT_B8)         \
  X(LE_I32) X(LE_I64) X(LE_U32) X(LE_U64) X(LE_F32) X(LE_F64) X(LE_B8)         \
  X(EQ_32) X(EQ_64) X(EQ_F32) X(EQ_F64) X(EQ_B8)                               \
  X(NE_32) X(NE_64) X(NE_F32) X(NE_F64) X(NE_B8)                               \
  /* mixed-sign 64-bit compares (value-correct, operators.h:189-214) */        \
  X(LT_I64_U64) X(LT_U64_I64) X(LE_I64_U64) X(LE_U64_I64)                      \
  X(EQ_I64_U64) X(NE_I64_U64)                                                  \
  /* ---- casts (cast_bound_expression.cc) ---------------------------------- */\
  X(CAST_I32_I64) X(CAST_U32_I64) X(CAST_I64_I32) /* I64->32 = truncation */   \
  X(CAST_I32_F32) X(CAST_I32_F64) X(CAST_U32_F32) X(CAST_U32_F64)              \
  X(CAST_I64_F32) X(CAST_I64_F64) X(CAST_U64_F32) X(CAST_U64_F64)              \
  X(CAST_F32_F64) X(CAST_F64_F32)                                              \
  X(CAST_F32_I32) X(CAST_F32_I64) X(CAST_F32_U32) X(CAST_F32_U64)              \
  X(CAST_F64_I32) X(CAST_F64_I64) X(CAST_F64_U32) X(CAST_F64_U64)              \
  X(CAST_B8_I32) X(CAST_B8_I64) X(CAST_B8_F32) X(CAST_B8_F64)                  \
  X(CAST_32_B8) X(CAST_64_B8) X(CAST_F32_B8) X(CAST_F64_B8)                    \
  X(COPY_8) X(COPY_32) X(COPY_64)                                              \
  /* ---- boolean logic on byte vectors (vector_logic.h:30-50) -------------- */\
  X(AND_B8) X(OR_B8) X(XOR_B8) X(ANDNOT_B8) X(NOT_B8)                          \
  /* 3-valued: dst,c = value,null out; a,b values; imm packs null offsets    */\
  X(AND3) X(OR3)                                                               \
  X(NULL_OR)   /* dst = a | b  on null masks */                                \
  X(NULL_DIVZERO_32) X(NULL_DIVZERO_64) X(NULL_DIVZERO_F32) X(NULL_DIVZERO_F64)\
  X(FAIL_DIVZERO_32) X(FAIL_DIVZERO_64) X(FAIL_DIVZERO_F32) X(FAIL_DIVZERO_F64)\
  X(FILL_8) X(FILL_32) X(FILL_64) X(ROWID_64)                                  \
  /* ---- libm family: dst = fn(a) / fn(a, b), fn = imm (VM_MATH_*); only in the MATH kernel variant */\
  X(MATH1_F64) X(MATH2_F64)                                                    \
  X(SELECT_8) X(SELECT_32) X(SELECT_64) /* dst = c ? a : b  (IF / IFNULL) */   \
  X(SEL_FROM_PRED) /* dst = a(value) & !b(null)        filter.cc:180-196 */    \
  /* ---- scalar-aggregate sinks: dst = slot, a = value, b = null, c = sel -- */\
  X(AGG_COUNT)                                                                 \
  X(AGG_SUM_I32) X(AGG_SUM_U32) X(AGG_SUM_I64) X(AGG_SUM_F32) X(AGG_SUM_F64)   \
  X(AGG_MIN_I32) X(AGG_MIN_U32) X(AGG_MIN_I64) X(AGG_MIN_U64)                  \
  X(AGG_MIN_F32) X(AGG_MIN_F64) X(AGG_MIN_B8)                                  \
  X(AGG_MAX_I32) X(AGG_MAX_U32) X(AGG_MAX_I64) X(AGG_MAX_U64)                  \
  X(AGG_MAX_F32) X(AGG_MAX_F64) X(AGG_MAX_B8)                                  \
  X(AGG_FIRST_8) X(AGG_FIRST_32) X(AGG_FIRST_64)                               \
  X(AGG_LAST_8) X(AGG_LAST_32) X(AGG_LAST_64)                                  \
  /* fused SUM(a op d) for register-resident slots: a, d operands (reg_ops bit 1 describes d) */\
  X(AGG_SUM_I64_ADD) X(AGG_SUM_I64_SUB) X(AGG_SUM_I64_MUL)                     \
  X(AGG_SUM_F64_ADD) X(AGG_SUM_F64_SUB) X(AGG_SUM_F64_MUL)                     \
  /* ---- materialising sinks ---------------------------------------------- */\
  X(SEL_COUNT)   /* a = sel: tile_counts[tile] = #selected        */           \
  X(SEL_RANK) X(PART_RANK) X(PART_REC_8) X(PART_REC_32) X(PART_REC_64) X(PART_REC_128) X(PART_FLUSH) X(JOIN_PROBE) X(JOIN_PROBE_WIDE) X(IDX_VALID) X(GATHER_64) X(GATHER_32) X(GATHER_8) X(GATHER_NULL)    /* a = sel, dst = u32 rank reg (absolute out row) */          \
  X(STORE_8) X(STORE_32) X(STORE_64)     /* dst = out col, a = reg, rows 1:1 */\
  X(STOREC_8) X(STOREC_32) X(STOREC_64)  /* + b = rank reg, c = sel (compact)*/\
  X(STORE_ROWID) /* dst = out col: int64 global row id of survivors, b,c */    \
  /* single-pass compaction (decoupled look-back over the strided tiles, VmParams.lb_*):                         */\
  /* SEL_RANK_LB: a = sel -> dst = u32 reg: tile row of the tile's i-th survivor; publishes the tile's count,    */\
  /* resolves the rows kept by all earlier tiles (scratch words 32/33 = first output row / survivors)            */\
  /* STOREG_*: dst = out col, a = value, b = the SEL_RANK_LB register: survivors gathered from LDS, lanes on     */\
  /* consecutive output rows (coalesced, line-aligned stores)                                                    */\
  X(SEL_RANK_LB) X(STOREG_8) X(STOREG_32) X(STOREG_64)                         \
  X(BARRIER)     /* workgroup barrier: registers written so far may be read by other threads (STOREG) */ \
  /* PART_RANK: a = key(64), c = sel -> dst = u32 position of the row in the tile's partition-sorted order,    */\
  /* VM_NONE for unselected rows; PART_REC_*: a (, d) = value reg(s), b = position reg, imm = byte offset      */\
  /* inside the record | record bytes << 16 (AoS records in the LDS staging area); PART_FLUSH: imm = record    */\
  /* bytes: staging area -> the rows' (hash partition, workgroup) segments of outputs[0]                       */\
  /* ---- group-aggregate sinks --------------------------------------------- */\
  X(KEY_ZERO)    /* dst(64) = 0 */                                             \
  X(KEY_APPEND_8) X(KEY_APPEND_32) X(KEY_APPEND_64)                            \
                 /* dst |= (a & mask) << shift, null b sets flag bit; imm =   */\
                 /* shift | bits<<8 | nullbit<<16 (nullbit 0xFF = none)        */\
  X(GRP_INSERT)  /* a = key(64), c = sel -> dst = u32 slot reg */              \
  X(GAGG_COUNT)                                                                \
  X(GAGG_SUM_I32) X(GAGG_SUM_U32) X(GAGG_SUM_I64) X(GAGG_SUM_F32) X(GAGG_SUM_F64)\
  X(GAGG_MIN_I32) X(GAGG_MIN_U32) X(GAGG_MIN_I64) X(GAGG_MIN_U64)              \
  X(GAGG_MIN_F32) X(GAGG_MIN_F64) X(GAGG_MIN_B8)                               \
  X(GAGG_MAX_I32) X(GAGG_MAX_U32) X(GAGG_MAX_I64) X(GAGG_MAX_U64)              \
  X(GAGG_MAX_F32) X(GAGG_MAX_F64) X(GAGG_MAX_B8)                               \
  X(GAGG_FIRST) X(GAGG_LAST)                                                   \
  X(OP_COUNT_)

enum VmOp : uint16_t {
#define X(n) VM_##n,
  VM_OPS(X)
#undef X
};

struct VmInstr {
  uint16_t op;
  uint8_t reg_ops;   /* bit 0: operand a is a row register, bit 1: operand b (fused aggregates: d) is;
                        a clear bit = the immediate, whose operand field points at the LDS constant pool */
  uint8_t imm_width; /* width in bytes (1/4/8) of the immediate, 0 = none */
  uint32_t dst;  /* LDS byte offset | aggregate slot | output column index */
  uint32_t a, b, c, d; /* LDS byte offsets, VM_NONE when absent */
  uint64_t imm;
};
static_assert(sizeof(VmInstr) == 32, "VmInstr must be 32 bytes");

struct VmStagedCol {
  const void* src;   /* device pointer of the column (data or null mask) */
  uint32_t lds_off;  /* destination register */
  uint32_t width;    /* 1, 4 or 8 bytes per row */
};

struct VmOutCol {
  void* dst;         /* device pointer of the output column */
  uint32_t width;
  uint32_t pad;
};

/* One partial-aggregate record (per workgroup per slot, then per slot). */
struct VmAccRec {
  uint64_t v0;  /* value bits (sum hi / min / max / first|last value) */
  uint64_t v1;  /* double-double low part, or row id for FIRST/LAST   */
  uint64_t cnt; /* number of contributing (non-NULL, selected) rows    */
  uint64_t pad;
};

/* Group table (open addressing, 64-bit packed keys). */
struct VmGroupTable {
  unsigned long long* keys;      /* capacity + 1 entries; [capacity] != EMPTY marks the EMPTY-valued key present */
  unsigned long long* acc;       /* [capacity+1][n_gaggs] value bits              */
  unsigned int* cnt;             /* [capacity+1][n_gaggs] contribution counts      */
  unsigned int* overflow;        /* set to 1 when probing exhausted the table      */
  uint32_t capacity_mask;        /* capacity - 1                                   */
  /* workgroup-private pre-aggregation table in LDS (local_capacity == 0: disabled).  Rows
   * whose key does not fit go straight to the global table; the local entries are merged
   * into the global table once, at the end of the kernel. */
  uint32_t local_capacity;       /* entries (a multiple of local_sub)              */
  uint32_t local_sub;            /* independent sub-tables, lane % local_sub picks one (power of two) */
  uint32_t local_sub_capacity;   /* local_capacity / local_sub                     */
  uint32_t local_keys_off;       /* LDS offsets: u64 keys[C], u64 acc[C][n], u32 cnt[C][n] */
  uint32_t local_acc_off;
  uint32_t local_cnt_off;        /* VM_NONE: no aggregate needs a contribution count */
  uint32_t local_stride;         /* accumulator words between two LDS entries: n_gaggs made odd (an even stride puts word k of
                                    every entry into a few LDS banks: 8- to 32-way conflicts on every atomic) */
  uint32_t local_pad;
  uint32_t n_gaggs;
  const unsigned long long* acc_init;  /* [n_gaggs] identities                      */
  const unsigned int* merge_op;        /* [n_gaggs] VM_MERGE_*                       */
  unsigned int* stats;           /* [0] rows that bypassed the local table, [1] max local occupancy */
};
#define VM_MERGE_ADD_U64 0u
#define VM_MERGE_MIN_U64 1u
#define VM_MERGE_MAX_U64 2u
#define VM_MERGE_ADD_F64 3u
#define VM_MERGE_ADD_F64_HI 4u /* high word of a compensated DOUBLE sum: the rounding error of the merge goes to word s + 1 */
#define VM_SLOT_LOCAL 0x80000000u /* GRP_INSERT result: index into the LDS table */
#define VM_KEY_EMPTY 0xFFFFFFFFFFFFFFFFull

/* HashJoin (UNIQUE rhs keys) fused into the pipeline: JOIN_PROBE looks a packed 64-bit lhs key up
 * in an open-addressing index over the rhs table (imm = join number) and leaves the matched rhs
 * row (or VM_NONE) in a u32 register; GATHER_* (imm = slot of join_cols) fetch rhs values by it. */
struct VmJoin {
  const unsigned long long* keys;   /* JOIN_PROBE: capacity {key, answer} pairs (16 bytes each), key == VM_KEY_EMPTY = free;
                                       JOIN_PROBE_WIDE: the first key word of every entry */
  const unsigned int* rows;         /* JOIN_PROBE_WIDE: rhs row of every entry, VM_NONE = free */
  const unsigned int* special;      /* [0]: rhs row whose packed key equals VM_KEY_EMPTY, or VM_NONE */
  uint32_t capacity_mask;
  uint32_t answer_slot;             /* JOIN_PROBE_WIDE: 1 = answer with the key's slot (NOT_UNIQUE index), 0 = with rows[slot] */
  const unsigned long long* keys_hi; /* JOIN_PROBE_WIDE: second word of every entry's key (rows[slot] == VM_NONE = free) */
};
struct VmJoinCol { const void* data; const unsigned char* is_null; };
#define VM_MAX_JOINS 2
#define VM_MAX_JOIN_COLS 24

struct VmParams {
  const VmInstr* prog;
  int32_t n_instr;
  int32_t n_staged;
  int32_t n_outputs;
  int32_t n_slots;        /* scalar aggregate slots */
  int64_t n_rows;
  int64_t row_id_base;    /* global row id of row 0 (multi-GPU shards) */
  int32_t tile_rows;      /* 512 * K */
  int32_t n_tiles;
  uint32_t acc_lds_off;   /* LDS offset of the aggregate accumulators */
  uint32_t scratch_lds_off; /* LDS offset of 256 B of scan scratch */
  uint32_t imm_pool_lds_off; /* LDS offset of the constant pool: 16 B per instruction */
  uint32_t const_lds_off;    /* LDS offset of 2 x tile_rows bytes: all 0x01, then all 0x00 */
  uint32_t lds_bytes;
  int32_t count_sub_k;    /* SEL_COUNT: 512-row units per counted tile (0 = this launch's K): the store pass's K */
  uint32_t flags;         /* VM_FLAG_*: bit 0 = tiles handed out in XCD-contiguous chunks (see vm_first_tile) */
  uint64_t slot_init0[VM_FAST_SLOTS]; /* per-lane identities of the fast slots */
  uint64_t slot_init1[VM_FAST_SLOTS];
  int32_t slot_kind[VM_FAST_SLOTS];   /* SlotKind of the fast slots */
  VmAccRec* wg_partials;        /* [grid][n_slots] */
  unsigned int* tile_counts;    /* SEL_COUNT output / scanned offsets input; PART_RANK: rows of segment [partition][workgroup] */
  uint32_t part_n;              /* PART_RANK: number of hash partitions */
  uint32_t part_lds_off;        /* LDS offset of part_n u32 counters */
  uint32_t part_seg_cap;        /* PART_RANK: records a (partition, workgroup) segment holds */
  uint32_t part_pad;
  unsigned int* part_overflow;  /* PART_RANK: set to 1 when a segment was full (the host reruns with larger segments) */
  const unsigned int* tile_offsets;
  unsigned long long* lb_status;  /* SEL_RANK_LB: one word per tile: state << 62 | epoch << 32 | rows (0 = not yet) */
  unsigned int* lb_ctrl;          /* [0..1] u64 total survivors, [3] look-back gave up                              */
  unsigned long long lb_epoch;    /* run stamp of lb_status words (stale words of earlier runs read as "not yet")    */
  const unsigned long long* n_rows_dev;  /* optional: the input's row count lives on the DEVICE (a stage fed by a stage whose row count no host has read yet); n_rows / n_tiles are then upper bounds */
  unsigned int* error_flag;     /* low byte != 0: evaluation error (signaling ops); SSGPU_FLAG_NAN_IN_MINMAX: see below */
  unsigned long long* debug;    /* optional [grid][4]: total cycles, barrier-wait cycles, tiles */
  unsigned long long* debug_pc; /* optional [n_instr + 1]: cycles per instruction (wave 0 of every workgroup); last = staging */
  uint32_t debug_pc_lds_off;    /* LDS scratch of the same shape (accumulated there, flushed once) */
  uint32_t uses_math;           /* the program has MATH1_F64 / MATH2_F64 instructions: launch the MATH kernel variant */
  VmGroupTable group;
  VmJoin join[VM_MAX_JOINS];
  VmJoinCol join_cols[VM_MAX_JOIN_COLS];
  VmStagedCol staged[VM_MAX_STAGED];
  VmOutCol outputs[VM_MAX_OUTPUTS];
};

#define VM_FLAG_XCD_CHUNKS 1u

#ifndef __HIPCC__
static inline const char* vm_op_name(uint16_t op) {
  static const char* names[] = {
#define X(n) #n,
      VM_OPS(X)
#undef X
  };
  return op < VM_OP_COUNT_ ? names[op] : "?";
}
#endif

/* Bit of a stage's error word: a NaN reached a floating MIN / MAX.  The kernels skip NaNs (order-independent); the reference keeps
 * a NaN that is the group's FIRST non-NULL value (aggregation_operators.h:189-228), so the host repeats such a run with the
 * plan lowered in its NaN-exact form (lower.cpp: PlanDesc::nan_exact) -- data without NaNs never pays for it. */
#define SSGPU_FLAG_NAN_IN_MINMAX 0x100u

#endif  // SSGPU_VM_H_
