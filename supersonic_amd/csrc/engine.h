// engine.h -- host-side plan model: schema, bound expressions, lowered stages.
//
// The binder restates Supersonic's bind-time behaviour (type promotion, result
// names, nullability, bind errors); the lowering pass turns a chain of
// Scan/Compute/Project/Filter operations ending in a blocking operator into one
// tile-VM program (vm.h).  Host code only: nothing here touches row data.
#ifndef SSGPU_ENGINE_H_
#define SSGPU_ENGINE_H_

#include <stdint.h>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ssgpu.h"
#include "vm.h"
#include "launch.h"

namespace ssgpu {

struct Status {
  int code = SSGPU_OK;
  std::string msg;
  bool ok() const { return code == SSGPU_OK; }
  static Status OK() { return Status(); }
  static Status Error(int c, const std::string& m) { Status s; s.code = c; s.msg = m; return s; }
};
#define SS_RETURN_IF_ERROR(expr) do { ::ssgpu::Status _s = (expr); if (!_s.ok()) return _s; } while (0)

struct Attr {
  std::string name;
  int dtype = SSGPU_INT64;
  bool nullable = false;
};
typedef std::vector<Attr> Schema;

const char* dtype_name(int dtype);
int dtype_width(int dtype);          // bytes per row on device; 0 = unsupported (STRING/BINARY)
bool dtype_is_integer(int dtype);
bool dtype_is_numeric(int dtype);
bool dtype_is_float(int dtype);
bool dtype_is_signed_int(int dtype);
std::string schema_to_string(const Schema& s);

// ---- bound expression (single attribute) ------------------------------------
struct BExpr;
typedef std::shared_ptr<BExpr> BExprP;
struct BExpr {
  enum Kind { INPUT, CONST, NULLCONST, OP, CAST, JOINCOL, JOINMATCH, JOINSTART, JOINCNT, ROWID /* the row's index in the plan's input (UINT64) */ } kind = INPUT;
  int op = 0;          // reference OperatorId for OP
  int dtype = SSGPU_INT64;
  bool nullable = false;
  std::string name;
  int input_col = -1;  // INPUT: column of the stage input; JOINCOL: column of the join's rhs table
  int join_id = -1;    // JOINCOL / JOINMATCH / JOINSTART / JOINCNT: index into the stage's joins
  uint64_t bits = 0;   // CONST: raw value bits in the column's device width
  int filter_depth = 0;  // number of Filter operations below this expression
  std::vector<BExprP> args;
};

// ---- lowered pipeline stage ---------------------------------------------------
enum StageKind {
  STAGE_SCALAR_AGG = 1,   // pipeline -> scalar aggregate slots -> 1 row
  STAGE_MATERIALIZE = 2,  // pipeline -> output columns (1:1 or compacted by a filter)
  STAGE_GROUP_AGG = 3,    // pipeline -> device hash table -> group rows
  STAGE_SORT = 4,         // radix sort of the stage input by key columns
  STAGE_CLUSTERS = 5,     // segmented aggregate over pre-clustered keys
  STAGE_JOIN_EXPAND = 6,  // NOT_UNIQUE hash join: (lhs row, matching rhs row) pairs -> gathered columns
  STAGE_FOLD_TAIL = 7     // GroupAggregateOptions::max_unique_keys_in_result: rows beyond the limit folded into the last kept row
};

struct AggOut {        // one aggregate result column
  int slot;            // accumulator slot (scalar) / index in the per-group record (group)
  int slot_kind;       // SlotKind
  int emit_kind;       // EmitKind
  bool has_cnt;        // group: contribution count tracked (nullable input)
  bool result_nullable;
  int gather_col = -1; // group FIRST/LAST: the slot holds a row id; the result is this stage-input column at that row
  bool gather_low32 = false;   // ... the row id is the slot's low half (its high half ordered the rows: AggPlan::order_pos)
};

struct GroupKeyField { int out_col; uint32_t shift, bits, nullbit, width; uint32_t word = 0; /* hash-join keys of 65..128 bits: key word 0 / 1 */ };

struct SortKey { int col; int order; };

// HashJoin fused into a pipeline (hash_join.h:37-56, UNIQUE rhs keys): lhs key expressions are
// packed like group keys, probed against an index built over the rhs table (the plan's auxiliary
// input) and the referenced rhs columns are gathered by the matched row.
struct JoinSpec {
  int type = 0;                          // SSGPU_JOIN_INNER / SSGPU_JOIN_LEFT_OUTER
  std::vector<BExprP> lhs_keys;          // over the stage input
  std::vector<int> rhs_key_cols;         // columns of the auxiliary input
  std::vector<GroupKeyField> fields;     // packing of the key: one 64-bit word, or two (`wide`, GroupKeyField::word)
  bool multi = false;                    // NOT_UNIQUE rhs keys: the index maps a key to a run of rhs rows
  bool wide = false;                     // the packed key takes two 64-bit words (JOIN_PROBE_WIDE)
  int depth = 0;                         // number of Filters below the join: rows their selection dropped are not probed
};
struct JoinGather { int join_id; int rhs_col; bool is_null_mask; };  // slot i of VmParams.join_cols
enum { JOIN_GATHER_RUN_START = -1, JOIN_GATHER_RUN_COUNT = -2 };   // rhs_col of a multi join's per-key run arrays

// Lowered instruction over virtual registers.  A register is an LDS array of
// tile_rows elements; its LDS byte offset is row_off * tile_rows, fixed when the
// tile size is chosen (finalize_program).
struct LReg { uint32_t width; uint32_t row_off; };
struct LInstr {
  uint16_t op = 0;
  bool a_imm = false, b_imm = false;
  uint8_t imm_width = 8;    // device width of the immediate operand (1 / 4 / 8)
  bool dst_is_reg = true;   // false: dst is a slot / output-column index
  int dst = -1;
  int a = -1, b = -1, c = -1, d = -1, e = -1;  // register ids, -1 = none
  uint64_t imm = 0;
};
struct StagedInput { int col; bool is_null_mask; int reg; };

struct Program {
  std::vector<LInstr> code;
  std::vector<LReg> regs;
  std::vector<StagedInput> staged;
  std::vector<JoinGather> gathers;   // rhs columns (and NULL masks) read by GATHER_* instructions
  bool uses_math = false;            // has MATH1_F64 / MATH2_F64: runs in the MATH kernel variant
  uint32_t bytes_per_row = 0;   // LDS bytes per tile row (peak of live registers)
  uint32_t in_bytes_per_row = 0;  // of which: the staged input registers
  int n_slots = 0;
  int n_outputs = 0;
  bool empty() const { return code.empty(); }
};

struct Stage {
  StageKind kind = STAGE_MATERIALIZE;
  Schema in_schema;     // schema of the stage input (plan input or previous stage's result)
  Schema out_schema;
  Program main;         // the pipeline program
  Program count_pass;   // STAGE_MATERIALIZE with a filter, two-pass form: predicate + SEL_COUNT
  bool single_pass = false;   // STAGE_MATERIALIZE with a filter: SEL_RANK_LB + STOREG (no count pass)
  bool has_filter = false;
  std::vector<AggOut> aggs;               // SCALAR_AGG / GROUP_AGG (out_schema order, after keys)
  std::vector<GroupKeyField> group_keys;  // GROUP_AGG
  std::vector<uint64_t> group_acc_init;   // per-group accumulator identities
  std::vector<uint32_t> group_merge_op;   // VM_MERGE_* of every group accumulator
  int n_gaggs = 0;
  // GROUP_AGG, partitioned execution (many groups): ONE pass scatters a record per selected row (packed key +
  // the distinct aggregate inputs, AoS) into its (hash partition, workgroup) segment, then one workgroup
  // aggregates each partition in LDS
  Program part_scatter;  // filters + key packing + PART_RANK + PART_REC_* per record field (pair)
  uint32_t part_rec_bytes = 0;            // record size (multiple of 8); the packed key is bytes [0, 8)
  struct PartAgg { int op; int val_off; int val_width; int null_off; int has_cnt; int word; };
  std::vector<PartAgg> part_aggs;         // one per aggregate: GAGG opcode, byte offsets of its value / NULL flag in the
                                          // record (-1 = none), first accumulator word in the group table
  // The same scatter as a kernel of its own (ssgpu_part_scatter_plain_kernel) when the stage is "plain": every group key and
  // every aggregate input is an input column as it stands (no cast, no expression), every Filter below the aggregate is
  // `column CMP constant`, and there is no join.  Record layout = part_scatter's, so phase 2 is the same kernel.
  struct PlainScatter {
    bool ok = false;
    struct Key { int col; uint32_t width, shift, bits, nullbit; bool nullable; int dtype = 0; };
    struct Field { int col; bool is_null_mask; uint32_t width, off; };
    struct Pred { int col; int kind /* 0 i32, 1 u32, 2 i64, 3 u64, 4 f32, 5 f64, 6 one byte (BOOL) */; int cmp /* 0 <, 1 <=, 2 ==, 3 != */; bool col_on_left; bool nullable; uint64_t bits; };
    std::vector<Key> keys; std::vector<Field> fields; std::vector<Pred> preds;
  } plain;
  std::vector<JoinSpec> joins;            // HashJoins fused into this stage's programs
  std::vector<SortKey> sort_keys;         // SORT / CLUSTERS (columns of in_schema)
  std::vector<int> sort_out_cols;         // SORT: projected input columns
  // DISTINCT aggregates (SCALAR_AGG / CLUSTERS over rows sorted by these columns: group keys + the distinct column): the
  // runtime appends a BOOL input column that is 1 on the first row of every run of equal values of these columns
  std::vector<int> distinct_cols;
  // MATERIALIZE: the stage's last (synthetic) input column holds, for every input row, the number of its cluster -- runs of equal
  // values of these input columns (the AggregateClusters boundary scan), computed in front of the stage's program
  std::vector<int> segment_cols;
  // MATERIALIZE, DISTINCT aggregates under max_unique_keys_in_result: the input rows are sorted by the group keys segment_cols (stable) and
  // carry their input row id in column rank_rowid_col; two synthetic input columns follow the stage's columns -- the row's RESULT ROW
  // (first-seen rank of its key, clamped to rank_limit: row_hash_set.cc:500-511) as UINT32, and a BOOL that says its own key keeps a row
  bool has_rank = false; int64_t rank_limit = 0; int rank_rowid_col = -1;
  bool has_segment = false;               // ... the stage HAS that synthetic column (segment_cols may be empty: AggregateClusters without a clustering column = one cluster)
  // SCALAR_AGG / CLUSTERS: SUM of a floating input column into an integer result -- the reference adds and truncates row after
  // row (aggregation_operators.h:173-185), so the stage's program only counts the column (the result's slot) and the runtime
  // folds the rows in their order afterwards (ssgpu_launch_seq_sum) into result column `out_col`
  struct SeqSum { int out_col; int in_col; };
  std::vector<SeqSum> seq_sums;
  // JOIN_EXPAND: the previous stage materialised [lhs fields..., run start, run count]; every output
  // column is lhs field `col` (from_rhs = false) or column `col` of the auxiliary input
  struct JoinOut { bool from_rhs; int col; };
  std::vector<JoinOut> join_out;
  // FOLD_TAIL (input: the group table sorted by first-seen row id, that id as the last column): rows [0, fold_limit] are
  // kept, every later row is merged into row fold_limit with the column's merge function (0 keep = key columns, 1 SUM, 2 MIN, 3 MAX,
  // 4 / 5 = FIRST / LAST: the value of the merged row whose UINT64 column fold_by[i] -- a row id -- is smallest / largest)
  int64_t fold_limit = -1;
  bool fold_cut = false;   // BestEffortGroupAggregate: rows [0, capacity) are KEPT as they are, nothing is merged; the first-seen id of row `capacity` is the next view's first input row
  std::vector<int> fold_op, fold_by;
  // CONCAT aggregates (column_aggregator.cc:496-505, aggregation_operators.h:236-283): a result of STRINGs that exist nowhere yet.
  // The device counts the contributing (non-NULL) values in the column's place -- a UINT64 in out_schema -- and the host
  // builds the strings when the column is fetched, from the stage's (materialised, for a group aggregate key-sorted) input:
  // out_col of out_schema <- values of stage-input column src_col, in input order, joined with ','.
  struct ConcatCol {
    int out_col; int src_col; int src_dtype;
    int stage = -1;          // the CLUSTERS stage whose ordered input and segment ids hold the values (-1: this, the last stage)
    bool distinct = false;   // DISTINCT CONCAT: a result row prints every value once, at its first occurrence (column_aggregator.cc:308-376)
  };
  std::vector<ConcatCol> concat;
  int64_t algorithmic_bytes_per_row = 0;  // staged input bytes per input row
  int64_t output_bytes_per_row = 0;       // materialised output bytes per output row
};

// ---- plan description (host copy of ssgpu_plan_desc with owned strings) --------
struct PlanDesc {
  Schema input_schema;
  Schema aux_schema;                 // auxiliary input (rhs of a HASH_JOIN)
  std::vector<ssgpu_op> ops;
  std::vector<ssgpu_expr> exprs;
  std::vector<int32_t> expr_args;
  std::vector<ssgpu_proj> projs;
  std::vector<ssgpu_agg> aggs;
  std::vector<ssgpu_sortkey> sortkeys;
  std::vector<std::string> strings;  // storage for all names (stable addresses)
  bool filter_single_pass = false;   // ctx option of the same name at plan creation (lower.cpp, finish_materialize)
  int part_rec_align = 0;            // ctx option: partition records padded to a multiple of this many bytes (0 = 8)
  // the reference keeps a NaN that is a group's FIRST non-NULL value as its floating MIN / MAX (aggregation_operators.h:189-228):
  // in this form every such aggregate gets a hidden FIRST of the same column and the result is IF(IS_NAN(first), first, min).
  // Set by the runtime when a run met a NaN in a floating MIN / MAX (SSGPU_FLAG_NAN_IN_MINMAX); plans start without it.
  bool nan_exact = false;
};
Status copy_plan_desc(const ssgpu_plan_desc* d, PlanDesc* out);

// bind.cpp: expression binding against a schema.  Results are expressions over
// the schema's columns (INPUT nodes index `schema`).
Status bind_expression(const PlanDesc& d, int expr_index, const Schema& schema, int filter_depth,
                       std::vector<BExprP>* out);
// projector binding: positions into `schema` + result names
Status bind_projector(const PlanDesc& d, int first, int n, const Schema& schema,
                      std::vector<int>* positions, std::vector<std::string>* names);
std::string bexpr_to_string(const BExprP& e);

// lower.cpp: operation chain -> stages
struct LowerOptions {
  int lds_target_bytes = 48 * 1024;
  int tile_rows = 0;  // 0 = choose by lds_target_bytes
};
Status lower_plan(const PlanDesc& d, std::vector<Stage>* stages, Schema* result_schema, std::string* describe);
// choose the tile size (K = tile_rows / 512) and fix LDS offsets for a program
struct ProgramLayout { int K; uint32_t lds_bytes; uint32_t acc_off; uint32_t scratch_off; uint32_t imm_pool_off; };
ProgramLayout layout_program(const Program& p, const LowerOptions& opt);
// final device instructions for a tile of `tile_rows`: two variants (one per input buffer),
// each n + 1 instructions (trailing NOP for the prefetch)
void finalize_program(const Program& p, const ProgramLayout& L, std::vector<VmInstr>* out);
std::string disassemble(const Program& p);

}  // namespace ssgpu
#endif  // SSGPU_ENGINE_H_
