// lower.cpp -- operation chain -> pipeline stages -> tile-VM programs.
//
// Restates the cursor layer's composition rules:
//   Compute   = replace the row's columns by the expression's result columns
//               supersonic/cursor/core/compute.cc:49-79
//   Project   = pointer re-mapping                       cursor/core/project.cc:49-59
//   Filter    = keep rows whose predicate is non-NULL TRUE, then project
//               cursor/core/filter.cc:84-92,170-199,287-303
//   ScalarAggregate / GroupAggregate binding of the AggregationSpecification
//               cursor/core/aggregator.cc:63-186, column_aggregator.cc:484-560
// Everything between the Scan and the first blocking operator is fused into ONE
// program: expressions compose by substitution, filters become a selection byte
// vector, no intermediate column or row-id list is ever written to HBM.
#include "engine.h"

#include <algorithm>
#include <functional>
#include <sstream>
#include <string.h>

namespace ssgpu {

enum {
  OP_ADD = 0, OP_MULTIPLY = 4, OP_SUBTRACT = 8, OP_DIVIDE_QUIET = 13, OP_DIVIDE_NULLING = 14,
  OP_DIVIDE_SIGNALING = 15, OP_CPP_DIVIDE_NULLING = 18, OP_CPP_DIVIDE_SIGNALING = 19,
  OP_MODULUS_NULLING = 26, OP_MODULUS_SIGNALING = 27, OP_NEGATE = 36,
  OP_IS_ODD = 140, OP_IS_EVEN = 144, OP_IS_FINITE = 148, OP_IS_INF = 152, OP_IS_NAN = 156, OP_IS_NORMAL = 160,
  OP_ROUND_WITH_MULTIPLIER = 364, OP_EXP = 320, OP_LN_QUIET = 325, OP_LN_NULLING = 326, OP_LOG10_QUIET = 329, OP_LOG10_NULLING = 330, OP_POW_QUIET = 353,
  OP_POW_NULLING = 354, OP_POW_SIGNALING = 355, OP_LOG2_QUIET = 357, OP_LOG2_NULLING = 358, OP_SIN = 800, OP_COS = 804, OP_TAN = 808,
  OP_ASIN = 812, OP_ACOS = 816, OP_ATAN = 820, OP_ATAN2 = 824, OP_SINH = 828, OP_COSH = 832, OP_TANH = 836, OP_ASINH = 840,
  OP_ACOSH = 844, OP_ATANH = 848,
  OP_ROUND = 300, OP_TRUNC = 304, OP_CEIL_TO_INT = 308, OP_FLOOR_TO_INT = 312, OP_SQRT_QUIET = 333, OP_SQRT_NULLING = 334,
  OP_SQRT_SIGNALING = 335, OP_CEIL = 342, OP_FLOOR = 346, OP_ABS = 360, OP_AND = 40, OP_OR = 44,
  OP_AND_NOT = 48, OP_NOT = 52, OP_XOR = 56, OP_BITWISE_AND = 60, OP_BITWISE_OR = 64,
  OP_BITWISE_NOT = 68, OP_BITWISE_XOR = 72, OP_SHIFT_LEFT = 76, OP_SHIFT_RIGHT = 80,
  OP_BITWISE_ANDNOT = 84, OP_EQUAL = 100, OP_NOT_EQUAL = 104, OP_LESS = 116, OP_LESS_OR_EQUAL = 120,
  OP_IF = 204, OP_IF_NULL = 220, OP_IS_NULL = 224, OP_CAST_QUIET = 265
};

// machine type classes
enum MT { M_I32, M_U32, M_I64, M_U64, M_F32, M_F64, M_B8, M_BAD };
static MT mtype(int dtype) {
  switch (dtype) {
    case SSGPU_INT32: case SSGPU_DATE: case SSGPU_STRING: return M_I32;   // STRING = order-preserving dictionary code
    case SSGPU_UINT32: return M_U32;
    case SSGPU_INT64: case SSGPU_DATETIME: return M_I64;
    case SSGPU_UINT64: return M_U64;
    case SSGPU_FLOAT: return M_F32;
    case SSGPU_DOUBLE: return M_F64;
    case SSGPU_BOOL: return M_B8;
  }
  return M_BAD;
}
static uint32_t mwidth(MT m) { return (m == M_I32 || m == M_U32 || m == M_F32) ? 4 : (m == M_B8 ? 1 : 8); }

struct Val {
  int reg = -1;       // value register (when !imm)
  bool imm = false;
  uint64_t bits = 0;
  int null = -1;      // null-mask register, -1 = never NULL
  uint32_t width = 8;
};

class Emitter {
 public:
  explicit Emitter(Program* p, const std::vector<JoinSpec>* joins = nullptr) : P(p), joins_(joins) {}

  int new_reg(uint32_t width) { LReg r; r.width = width; r.row_off = 0; P->regs.push_back(r); return (int)P->regs.size() - 1; }
  LInstr& emit(uint16_t op) { LInstr i; i.op = op; P->code.push_back(i); return P->code.back(); }

  int staged(int col, bool is_null, uint32_t width) {
    auto key = std::make_pair(col, is_null);
    auto it = staged_.find(key);
    if (it != staged_.end()) return it->second;
    int r = new_reg(width);
    StagedInput s; s.col = col; s.is_null_mask = is_null; s.reg = r;
    P->staged.push_back(s);
    staged_[key] = r;
    return r;
  }

  int materialize(const Val& v) {
    if (!v.imm) return v.reg;
    int r = new_reg(v.width);
    LInstr& i = emit(v.width == 8 ? VM_FILL_64 : v.width == 4 ? VM_FILL_32 : VM_FILL_8);
    i.dst = r; i.a_imm = true; i.imm = v.bits; i.imm_width = (uint8_t)v.width;
    return r;
  }
  int const_null_reg() {
    if (all_null_ < 0) {
      all_null_ = new_reg(1);
      LInstr& i = emit(VM_FILL_8); i.dst = all_null_; i.a_imm = true; i.imm = 1; i.imm_width = 1;
    }
    return all_null_;
  }
  int or_null(int a, int b) {
    if (a < 0) return b;
    if (b < 0) return a;
    if (a == b) return a;
    int r = new_reg(1);
    LInstr& i = emit(VM_NULL_OR); i.dst = r; i.a = a; i.b = b;
    return r;
  }
  // dst = op(a, b) with immediates folded into the instruction
  int binop(uint16_t op, Val a, Val b, uint32_t out_width) {
    if (a.imm && b.imm) { a.reg = materialize(a); a.imm = false; }
    int r = new_reg(out_width);
    LInstr& i = emit(op);
    i.dst = r;
    if (a.imm) { i.a_imm = true; i.imm = a.bits; i.imm_width = (uint8_t)a.width; } else i.a = a.reg;
    if (b.imm) { i.b_imm = true; i.imm = b.bits; i.imm_width = (uint8_t)b.width; } else i.b = b.reg;
    return r;
  }
  int unop(uint16_t op, Val a, uint32_t out_width) {
    int r = new_reg(out_width);
    LInstr& i = emit(op);
    i.dst = r;
    if (a.imm) { i.a_imm = true; i.imm = a.bits; i.imm_width = (uint8_t)a.width; } else i.a = a.reg;
    return r;
  }

  Status value(const BExprP& e, Val* out);
  bool has_value(const BExprP& e) { return memo_.count(memo_key(e)) != 0; }
  Status cast_val(const Val& v, MT from, MT to, Val* out);

  // selection registers: sel_by_depth[d] = rows passing the first d filters (-1 = all)
  std::vector<int> sel_by_depth{-1};
  int sel_at(int depth) const { return sel_by_depth[std::min<size_t>(depth, sel_by_depth.size() - 1)]; }

  Program* P;

  // HashJoin: matched rhs row of every lhs row (u32 register, VM_NONE = no match); emitted once
  Status join_index(int join_id, int* reg);
  int gather_slot(int join_id, int rhs_col, bool is_null) {
    for (size_t i = 0; i < P->gathers.size(); ++i)
      if (P->gathers[i].join_id == join_id && P->gathers[i].rhs_col == rhs_col && P->gathers[i].is_null_mask == is_null) return (int)i;
    JoinGather g; g.join_id = join_id; g.rhs_col = rhs_col; g.is_null_mask = is_null;
    P->gathers.push_back(g);
    return (int)P->gathers.size() - 1;
  }

  // join_index: the stage's filter chain and whether later filters may be evaluated ahead of a probe (emit_filters sets both)
  const std::vector<BExprP>* filters_ = nullptr;
  bool hoist_ok_ = false;
  bool may_fail(const BExprP& e) { return can_fail(e); }

 private:
  const std::vector<JoinSpec>* joins_;
  std::map<int, int> join_idx_;
  std::string key_of(const BExprP& e);
  // Guarded evaluation.  The reference evaluates a child under a skip vector: the THEN / OTHERWISE branches of IF
  // only where they are chosen (elementary_bound_expressions.cc:935-955), the right side of AND / OR only where the
  // left side has not decided (:279-318), a later argument of any operator only where the earlier ones are not NULL
  // (children share one skip vector, abstract_bound_expressions.h:129-147) -- and a skipped row never raises an
  // evaluation error (the failers take the skip vector as is_null).  Values may be computed everywhere; what has to
  // respect the skip vector is the FAIL_* instructions.  guard_ = BOOL register, 1 = row evaluated (-1 = all).
  int guard_ = -1;
  std::map<const BExpr*, bool> reads_join_;
  bool reads_join(const BExprP& e) {
    auto it = reads_join_.find(e.get());
    if (it != reads_join_.end()) return it->second;
    bool f = e->kind == BExpr::JOINCOL || e->kind == BExpr::JOINMATCH || e->kind == BExpr::JOINSTART || e->kind == BExpr::JOINCNT;
    for (auto& a : e->args) f = f || reads_join(a);
    reads_join_[e.get()] = f;
    return f;
  }
  std::map<const BExpr*, bool> can_fail_;
  bool can_fail(const BExprP& e) {
    auto it = can_fail_.find(e.get());
    if (it != can_fail_.end()) return it->second;
    bool f = e->kind == BExpr::OP && (e->op == OP_DIVIDE_SIGNALING || e->op == OP_CPP_DIVIDE_SIGNALING || e->op == OP_MODULUS_SIGNALING ||
                                      e->op == OP_SQRT_SIGNALING || e->op == OP_POW_SIGNALING);
    for (auto& a : e->args) f = f || can_fail(a);
    can_fail_[e.get()] = f;
    return f;
  }
  std::string memo_key(const BExprP& e) { return (guard_ >= 0 && can_fail(e)) ? key_of(e) + "@g" + std::to_string(guard_) : key_of(e); }
  // guard AND mask / guard AND NOT mask (mask: BOOL register without NULLs)
  int narrow_guard(int g, int mask, bool negate) {
    int r = new_reg(1);
    if (g < 0) {
      if (!negate) return mask;
      LInstr& i = emit(VM_NOT_B8); i.dst = r; i.a = mask;
    } else {
      LInstr& i = emit(negate ? VM_ANDNOT_B8 : VM_AND_B8); i.dst = r; i.a = mask; i.b = g;   // ANDNOT_B8(a, b) = !a && b
    }
    return r;
  }
  // the rows a FAIL_* instruction looks at: selected by the filters below AND evaluated (guard)
  int fail_mask(int sel) {
    if (guard_ < 0) return sel;
    if (sel < 0) return guard_;
    return narrow_guard(sel, guard_, false);
  }
  Status eval_args(const BExprP& e, std::vector<Val>* a);
  std::map<std::pair<int, bool>, int> staged_;
  std::map<std::string, Val> memo_;
  int all_null_ = -1;
};

std::string Emitter::key_of(const BExprP& e) {
  std::ostringstream s;
  switch (e->kind) {
    case BExpr::INPUT: s << "I" << e->input_col; break;
    case BExpr::CONST: s << "C" << e->dtype << ":" << e->bits; break;
    case BExpr::NULLCONST: s << "N" << e->dtype; break;
    case BExpr::JOINCOL: s << "J" << e->join_id << ":" << e->input_col; break;
    case BExpr::JOINMATCH: s << "M" << e->join_id; break;
    case BExpr::JOINSTART: s << "JS" << e->join_id; break;
    case BExpr::JOINCNT: s << "JC" << e->join_id; break;
    case BExpr::ROWID: s << "R"; break;
    default:
      s << (e->kind == BExpr::CAST ? "K" : "O") << e->op << ":" << e->dtype << ":" << e->filter_depth << "(";
      for (auto& a : e->args) s << key_of(a) << ",";
      s << ")";
  }
  return s.str();
}

Status Emitter::cast_val(const Val& v, MT from, MT to, Val* out) {
  *out = v;
  out->width = mwidth(to);
  if (from == to) return Status::OK();
  auto same_bits = [&](MT a, MT b) { return (a == M_I32 && b == M_U32) || (a == M_U32 && b == M_I32) || (a == M_I64 && b == M_U64) || (a == M_U64 && b == M_I64); };
  if (same_bits(from, to)) return Status::OK();
  uint16_t op = VM_NOP;
  const bool to64 = to == M_I64 || to == M_U64, to32 = to == M_I32 || to == M_U32;
  if (from == M_I32 && to64) op = VM_CAST_I32_I64;
  else if (from == M_U32 && to64) op = VM_CAST_U32_I64;
  else if ((from == M_I64 || from == M_U64) && to32) op = VM_CAST_I64_I32;
  else if (from == M_I32 && to == M_F32) op = VM_CAST_I32_F32;
  else if (from == M_I32 && to == M_F64) op = VM_CAST_I32_F64;
  else if (from == M_U32 && to == M_F32) op = VM_CAST_U32_F32;
  else if (from == M_U32 && to == M_F64) op = VM_CAST_U32_F64;
  else if (from == M_I64 && to == M_F32) op = VM_CAST_I64_F32;
  else if (from == M_I64 && to == M_F64) op = VM_CAST_I64_F64;
  else if (from == M_U64 && to == M_F32) op = VM_CAST_U64_F32;
  else if (from == M_U64 && to == M_F64) op = VM_CAST_U64_F64;
  else if (from == M_F32 && to == M_F64) op = VM_CAST_F32_F64;
  else if (from == M_F64 && to == M_F32) op = VM_CAST_F64_F32;
  else if ((from == M_F32 || from == M_F64) && (to64 || to32)) {
    // C++ static_cast<integer>(floating) = truncation toward zero (what an aggregate into an integer result type stores,
    // aggregation_operators.h:100-122): TRUNC, then the exact conversion of the integral value (x86 cvttsd2si for the
    // out-of-range cases, like FLOOR_TO_INT), then the bits of the narrower type
    Val t = v;
    t.reg = unop(from == M_F32 ? VM_TRUNC_F32 : VM_TRUNC_F64, t, mwidth(from)); t.imm = false;
    Val i64v = t;
    i64v.reg = unop(from == M_F32 ? VM_FLOOR2I_F32 : VM_FLOOR2I_F64, t, 8); i64v.width = 8;
    if (to64) { *out = i64v; out->null = v.null; return Status::OK(); }
    Val n = i64v;
    n.reg = unop(VM_CAST_I64_I32, i64v, 4); n.width = 4; n.null = v.null;
    *out = n;
    return Status::OK();
  }
  else return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "cast not available on device");
  Val src = v;
  out->reg = unop(op, src, mwidth(to));
  out->imm = false;
  return Status::OK();
}

static uint16_t pick(MT m, uint16_t i32, uint16_t u32, uint16_t i64, uint16_t u64, uint16_t f32, uint16_t f64, uint16_t b8 = VM_NOP) {
  switch (m) {
    case M_I32: return i32; case M_U32: return u32; case M_I64: return i64; case M_U64: return u64;
    case M_F32: return f32; case M_F64: return f64; case M_B8: return b8; default: return VM_NOP;
  }
}

Status Emitter::join_index(int join_id, int* reg) {
  auto it = join_idx_.find(join_id);
  if (it != join_idx_.end()) { *reg = it->second; return Status::OK(); }
  if (!joins_ || join_id < 0 || join_id >= (int)joins_->size()) return Status::Error(SSGPU_ERROR_UNKNOWN, "join reference outside its stage");
  const JoinSpec& js = (*joins_)[join_id];
  // pack the lhs key exactly as the index build packs the rhs key (JoinBuildParams)
  int keyreg = -1, keyreg_hi = -1;
  int any_null = -1;   // a NULL in any key column: the row matches nothing
  // one 64-bit key column IS the packed key: no packing instructions
  const bool direct = js.fields.size() == 1 && js.fields[0].width == 8 && js.fields[0].shift == 0 && !js.wide;
  if (direct) {
    Val v; SS_RETURN_IF_ERROR(value(js.lhs_keys[0], &v));
    keyreg = materialize(v);
    any_null = v.null;
  } else {
  keyreg = new_reg(8);
  { LInstr& i = emit(VM_FILL_64); i.dst = keyreg; i.a_imm = true; i.imm = 0; i.imm_width = 8; }
  if (js.wide) {
    keyreg_hi = new_reg(8);
    LInstr& i = emit(VM_FILL_64); i.dst = keyreg_hi; i.a_imm = true; i.imm = 0; i.imm_width = 8;
  }
  for (size_t k = 0; k < js.lhs_keys.size(); ++k) {
    Val v; SS_RETURN_IF_ERROR(value(js.lhs_keys[k], &v));
    const GroupKeyField& f = js.fields[k];
    int vr = materialize(v);
    LInstr& i = emit(f.width == 8 ? VM_KEY_APPEND_64 : f.width == 4 ? VM_KEY_APPEND_32 : VM_KEY_APPEND_8);
    i.dst = f.word ? keyreg_hi : keyreg; i.a = vr; i.b = -1;
    i.imm = (uint64_t)f.shift | ((uint64_t)f.bits << 8) | ((uint64_t)f.nullbit << 16);
    any_null = or_null(any_null, v.null);
  }
  }
  // Rows the selection has already dropped are not probed: the selection at the join's depth and, when nothing in the stage
  // can raise an evaluation error, the later filters that read no join column (a Filter written ABOVE the join on lhs
  // columns: the probe and the rhs gathers are the expensive part of the row, the predicate is not).  Those rows look
  // unmatched, which nothing observes: every consumer of the join's columns sits at or above the join's depth, and a
  // hoisted filter drops its rows again at its own place in the chain.
  int probe_sel = js.depth >= 0 && js.depth < (int)sel_by_depth.size() ? sel_by_depth[js.depth] : -1;
  if (filters_ && hoist_ok_) {
    for (size_t f = (size_t)std::max(js.depth, 0); f < filters_->size(); ++f) {
      const BExprP& g = (*filters_)[f];
      if (reads_join(g)) continue;
      Val pv; SS_RETURN_IF_ERROR(value(g, &pv));
      if (pv.imm) continue;
      if (pv.null < 0 && probe_sel < 0) { probe_sel = pv.reg; continue; }
      const int r = new_reg(1);
      LInstr& i = emit(VM_SEL_FROM_PRED); i.dst = r; i.a = pv.reg; i.b = pv.null; i.c = probe_sel;
      probe_sel = r;
    }
  }
  const int idx = new_reg(4);
  { LInstr& i = emit(js.wide ? VM_JOIN_PROBE_WIDE : VM_JOIN_PROBE); i.dst = idx; i.a = keyreg; i.b = any_null; i.c = keyreg_hi; i.d = probe_sel; i.imm = (uint64_t)join_id; }
  join_idx_[join_id] = idx;
  *reg = idx;
  return Status::OK();
}

// The arguments of an operator, evaluated in the reference's order and under the reference's skip vectors.
Status Emitter::eval_args(const BExprP& e, std::vector<Val>* out) {
  std::vector<Val>& a = *out;
  a.resize(e->args.size());
  const int saved = guard_;
  auto restore = [&](const Status& s) { guard_ = saved; return s; };
  if (a.empty()) return Status::OK();
  Status st = value(e->args[0], &a[0]);
  if (!st.ok()) return restore(st);
  bool later_can_fail = false;
  for (size_t i = 1; i < e->args.size(); ++i) later_can_fail = later_can_fail || can_fail(e->args[i]);
  if (!later_can_fail) {
    for (size_t i = 1; i < a.size(); ++i) { st = value(e->args[i], &a[i]); if (!st.ok()) return restore(st); }
    return restore(Status::OK());
  }
  // "decided / chosen" mask of the first argument: BOOL value that is TRUE and not NULL
  auto true_not_null = [&](const Val& v) -> int {
    int r = materialize(v);
    if (v.null < 0) return r;
    int ch = new_reg(1);
    LInstr& c = emit(VM_SEL_FROM_PRED); c.dst = ch; c.a = r; c.b = v.null;
    return ch;
  };
  const int op = e->kind == BExpr::OP ? e->op : -1;
  if (op == OP_IF || op == SSGPU_OP_NULLING_IF) {
    const int choice = true_not_null(a[0]);
    guard_ = narrow_guard(saved, choice, false);
    st = value(e->args[1], &a[1]); if (!st.ok()) return restore(st);
    guard_ = narrow_guard(saved, choice, true);
    if (op == SSGPU_OP_NULLING_IF && a[0].null >= 0) guard_ = narrow_guard(guard_, a[0].null, true);   // a NULL condition skips both branches
    st = value(e->args[2], &a[2]);
    return restore(st);
  }
  if (op == OP_AND || op == OP_OR || op == OP_AND_NOT) {
    // right side skipped where the left side decides: TRUE (OR, AND_NOT) / FALSE (AND) and not NULL
    Val l = a[0];
    if (op == OP_AND) { Val n; n.width = 1; n.reg = unop(VM_NOT_B8, l, 1); n.null = l.null; l = n; }
    const int decided = true_not_null(l);
    guard_ = narrow_guard(saved, decided, true);
    st = value(e->args[1], &a[1]);
    return restore(st);
  }
  if (op == OP_IF_NULL) {   // the replacement is evaluated only where the first argument IS NULL
    if (a[0].null >= 0) guard_ = narrow_guard(saved, a[0].null, false);
    else { Val never; never.imm = true; never.bits = 0; never.width = 1; guard_ = narrow_guard(saved, materialize(never), false); }
    st = value(e->args[1], &a[1]);
    return restore(st);
  }
  // every other operator: the children share one skip vector, each adds its NULLs before the next is evaluated
  // (the masks of EVERY earlier argument count: with three or more arguments one that cannot fail may sit between a
  //  NULL-producing one and a signaling one -- pending masks are folded in before the first later argument that can fail)
  size_t folded = 0;   // arguments [0, folded) have their NULL masks in guard_
  for (size_t i = 1; i < a.size(); ++i) {
    if (can_fail(e->args[i]))
      for (; folded < i; ++folded)
        if (a[folded].null >= 0) guard_ = narrow_guard(guard_, a[folded].null, true);
    st = value(e->args[i], &a[i]);
    if (!st.ok()) return restore(st);
  }
  return restore(Status::OK());
}

Status Emitter::value(const BExprP& e, Val* out) {
  const std::string key = memo_key(e);
  auto it = memo_.find(key);
  if (it != memo_.end()) { *out = it->second; return Status::OK(); }
  const MT mt = mtype(e->dtype);
  if (mt == M_BAD)
    return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED,
                         std::string("type ") + dtype_name(e->dtype) + " is outside the device hot path (SURVEY 8f)");
  Val v; v.width = mwidth(mt);
  switch (e->kind) {
    case BExpr::INPUT:
      v.reg = staged(e->input_col, false, v.width);
      if (e->nullable) v.null = staged(e->input_col, true, 1);
      break;
    case BExpr::CONST: v.imm = true; v.bits = e->bits; break;
    case BExpr::NULLCONST: v.imm = true; v.bits = 0; v.null = const_null_reg(); break;
    case BExpr::ROWID: { v.reg = new_reg(8); LInstr& i = emit(VM_ROWID_64); i.dst = v.reg; } break;
    case BExpr::JOINMATCH: {
      int idx; SS_RETURN_IF_ERROR(join_index(e->join_id, &idx));
      v.reg = new_reg(1);
      LInstr& i = emit(VM_IDX_VALID); i.dst = v.reg; i.a = idx;
    } break;
    case BExpr::JOINSTART: case BExpr::JOINCNT: {
      // multi join: the probe yields the key's slot; its run of rhs rows is [start, start + count)
      int idx; SS_RETURN_IF_ERROR(join_index(e->join_id, &idx));
      v.reg = new_reg(4);
      LInstr& i = emit(VM_GATHER_32); i.dst = v.reg; i.a = idx;
      i.imm = (uint64_t)gather_slot(e->join_id, e->kind == BExpr::JOINSTART ? JOIN_GATHER_RUN_START : JOIN_GATHER_RUN_COUNT, false);
    } break;
    case BExpr::JOINCOL: {
      int idx; SS_RETURN_IF_ERROR(join_index(e->join_id, &idx));
      v.reg = new_reg(v.width);
      { LInstr& i = emit(v.width == 8 ? VM_GATHER_64 : v.width == 4 ? VM_GATHER_32 : VM_GATHER_8);
        i.dst = v.reg; i.a = idx; i.imm = (uint64_t)gather_slot(e->join_id, e->input_col, false); }
      if (e->nullable) {
        v.null = new_reg(1);
        LInstr& i = emit(VM_GATHER_NULL); i.dst = v.null; i.a = idx; i.imm = (uint64_t)gather_slot(e->join_id, e->input_col, true);
      }
    } break;
    case BExpr::CAST: {
      Val a; SS_RETURN_IF_ERROR(value(e->args[0], &a));
      SS_RETURN_IF_ERROR(cast_val(a, mtype(e->args[0]->dtype), mt, &v));
    } break;
    case BExpr::OP: {
      std::vector<Val> a;
      SS_RETURN_IF_ERROR(eval_args(e, &a));
      const MT at = mtype(e->args[0]->dtype);
      const int sel = fail_mask(sel_at(e->filter_depth));
      switch (e->op) {
        case OP_ADD: v.reg = binop(pick(mt, VM_ADD_I32, VM_ADD_I32, VM_ADD_I64, VM_ADD_I64, VM_ADD_F32, VM_ADD_F64), a[0], a[1], v.width); v.null = or_null(a[0].null, a[1].null); break;
        case OP_SUBTRACT: v.reg = binop(pick(mt, VM_SUB_I32, VM_SUB_I32, VM_SUB_I64, VM_SUB_I64, VM_SUB_F32, VM_SUB_F64), a[0], a[1], v.width); v.null = or_null(a[0].null, a[1].null); break;
        case OP_MULTIPLY: v.reg = binop(pick(mt, VM_MUL_I32, VM_MUL_I32, VM_MUL_I64, VM_MUL_I64, VM_MUL_F32, VM_MUL_F64), a[0], a[1], v.width); v.null = or_null(a[0].null, a[1].null); break;
        case OP_DIVIDE_QUIET: case OP_DIVIDE_NULLING: case OP_DIVIDE_SIGNALING:
        case OP_CPP_DIVIDE_NULLING: case OP_CPP_DIVIDE_SIGNALING:
        case OP_MODULUS_NULLING: case OP_MODULUS_SIGNALING: {
          const bool is_mod = e->op == OP_MODULUS_NULLING || e->op == OP_MODULUS_SIGNALING;
          const bool nulling = e->op == OP_DIVIDE_NULLING || e->op == OP_CPP_DIVIDE_NULLING || e->op == OP_MODULUS_NULLING;
          const bool signaling = e->op == OP_DIVIDE_SIGNALING || e->op == OP_CPP_DIVIDE_SIGNALING || e->op == OP_MODULUS_SIGNALING;
          uint16_t op = is_mod ? pick(mt, VM_MOD_I32, VM_MOD_U32, VM_MOD_I64, VM_MOD_U64, VM_NOP, VM_NOP)
                               : pick(mt, VM_CDIV_I32, VM_CDIV_U32, VM_CDIV_I64, VM_CDIV_U64, VM_DIV_F32, VM_DIV_F64);
          if (op == VM_NOP) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "division variant not available on device");
          int base_null = or_null(a[0].null, a[1].null);
          const uint16_t zop_null = pick(mt, VM_NULL_DIVZERO_32, VM_NULL_DIVZERO_32, VM_NULL_DIVZERO_64, VM_NULL_DIVZERO_64, VM_NULL_DIVZERO_F32, VM_NULL_DIVZERO_F64);
          const uint16_t zop_fail = pick(mt, VM_FAIL_DIVZERO_32, VM_FAIL_DIVZERO_32, VM_FAIL_DIVZERO_64, VM_FAIL_DIVZERO_64, VM_FAIL_DIVZERO_F32, VM_FAIL_DIVZERO_F64);
          // DIVIDE_SIGNALING on DOUBLE: "double division cannot fail at runtime" is the
          // reference's comment, but its CheckFailure still flags a zero divisor
          // (expression_traits.h:1229-1242); mirror the failer.
          if (nulling) {
            int r = new_reg(1);
            LInstr& i = emit(zop_null); i.dst = r; i.a = base_null;
            if (a[1].imm) { i.b_imm = true; i.imm = a[1].bits; i.imm_width = (uint8_t)a[1].width; } else i.b = a[1].reg;
            base_null = r;
          } else if (signaling) {
            LInstr& i = emit(zop_fail); i.dst_is_reg = false; i.dst = 0; i.a = base_null; i.c = sel;
            if (a[1].imm) { i.b_imm = true; i.imm = a[1].bits; i.imm_width = (uint8_t)a[1].width; } else i.b = a[1].reg;
          }
          v.reg = binop(op, a[0], a[1], v.width);
          v.null = base_null;
        } break;
        // exact math family (math_bound_expressions.cc:150-170,318-456,473-486)
        case OP_ABS: v.reg = unop(pick(at, VM_ABS_I32, VM_NOP, VM_ABS_I64, VM_NOP, VM_ABS_F32, VM_ABS_F64), a[0], v.width); v.null = a[0].null; break;
        case OP_ROUND: v.reg = unop(at == M_F32 ? VM_ROUND_F32 : VM_ROUND_F64, a[0], v.width); v.null = a[0].null; break;
        case OP_CEIL: v.reg = unop(at == M_F32 ? VM_CEIL_F32 : VM_CEIL_F64, a[0], v.width); v.null = a[0].null; break;
        case OP_FLOOR: v.reg = unop(at == M_F32 ? VM_FLOOR_F32 : VM_FLOOR_F64, a[0], v.width); v.null = a[0].null; break;
        case OP_TRUNC: v.reg = unop(at == M_F32 ? VM_TRUNC_F32 : VM_TRUNC_F64, a[0], v.width); v.null = a[0].null; break;
        case OP_CEIL_TO_INT: v.reg = unop(at == M_F32 ? VM_CEIL2I_F32 : VM_CEIL2I_F64, a[0], v.width); v.null = a[0].null; break;
        case OP_FLOOR_TO_INT: v.reg = unop(at == M_F32 ? VM_FLOOR2I_F32 : VM_FLOOR2I_F64, a[0], v.width); v.null = a[0].null; break;
        case OP_IS_FINITE: v.reg = unop(VM_ISFINITE_F64, a[0], 1); v.null = a[0].null; break;
        case OP_IS_NAN: v.reg = unop(VM_ISNAN_F64, a[0], 1); v.null = a[0].null; break;
        case OP_IS_INF: v.reg = unop(VM_ISINF_F64, a[0], 1); v.null = a[0].null; break;
        case OP_IS_NORMAL: v.reg = unop(VM_ISNORMAL_F64, a[0], 1); v.null = a[0].null; break;
        case OP_IS_ODD: case OP_IS_EVEN: {
          Val odd; odd.width = 1; odd.reg = unop(mwidth(at) == 4 ? VM_ISODD_32 : VM_ISODD_64, a[0], 1);
          v.reg = e->op == OP_IS_ODD ? odd.reg : unop(VM_NOT_B8, odd, 1);
          v.null = a[0].null;
        } break;
        case OP_EXP: case OP_LN_QUIET: case OP_LN_NULLING: case OP_LOG10_QUIET: case OP_LOG10_NULLING: case OP_LOG2_QUIET: case OP_LOG2_NULLING:
        case OP_SIN: case OP_COS: case OP_TAN: case OP_ASIN: case OP_ACOS: case OP_ATAN: case OP_SINH: case OP_COSH: case OP_TANH:
        case OP_ASINH: case OP_ACOSH: case OP_ATANH: {
          uint32_t fn = 0;
          switch (e->op) {
            case OP_EXP: fn = VM_MATH_EXP; break; case OP_LN_QUIET: case OP_LN_NULLING: fn = VM_MATH_LN; break;
            case OP_LOG10_QUIET: case OP_LOG10_NULLING: fn = VM_MATH_LOG10; break; case OP_LOG2_QUIET: case OP_LOG2_NULLING: fn = VM_MATH_LOG2; break;
            case OP_SIN: fn = VM_MATH_SIN; break; case OP_COS: fn = VM_MATH_COS; break; case OP_TAN: fn = VM_MATH_TAN; break;
            case OP_ASIN: fn = VM_MATH_ASIN; break; case OP_ACOS: fn = VM_MATH_ACOS; break; case OP_ATAN: fn = VM_MATH_ATAN; break;
            case OP_SINH: fn = VM_MATH_SINH; break; case OP_COSH: fn = VM_MATH_COSH; break; case OP_TANH: fn = VM_MATH_TANH; break;
            case OP_ASINH: fn = VM_MATH_ASINH; break; case OP_ACOSH: fn = VM_MATH_ACOSH; break; default: fn = VM_MATH_ATANH; break;
          }
          int base_null = a[0].null;
          if (e->op == OP_LN_NULLING || e->op == OP_LOG10_NULLING || e->op == OP_LOG2_NULLING) {
            // IsNonPositiveNuller (expression_traits.h:849-860): x <= 0
            Val zero; zero.imm = true; zero.bits = 0; zero.width = 8;
            Val np; np.width = 1; np.reg = binop(VM_LE_F64, a[0], zero, 1);
            if (base_null >= 0) { int r = new_reg(1); LInstr& i = emit(VM_NULL_OR); i.dst = r; i.a = base_null; i.b = np.reg; base_null = r; }
            else base_null = np.reg;
          }
          const int x = materialize(a[0]);
          v.reg = new_reg(8);
          { LInstr& i = emit(VM_MATH1_F64); i.dst = v.reg; i.a = x; i.imm = fn; }
          P->uses_math = true;
          v.null = base_null;
        } break;
        case OP_ROUND_WITH_MULTIPLIER: {   // round(x * m) / m: three correctly rounded IEEE operations (math_evaluators.h:117-121)
          Val t; t.width = 8; t.reg = binop(VM_MUL_F64, a[0], a[1], 8);
          Val r; r.width = 8; r.reg = unop(VM_ROUND_F64, t, 8);
          v.reg = binop(VM_DIV_F64, r, a[1], 8);
          v.null = or_null(a[0].null, a[1].null);
        } break;
        case OP_POW_QUIET: case OP_POW_NULLING: case OP_POW_SIGNALING: case OP_ATAN2: {
          int base_null = or_null(a[0].null, a[1].null);
          if (e->op == OP_POW_NULLING || e->op == OP_POW_SIGNALING) {
            // FirstColumnNegativeAndSecondNonInteger (expression_traits.h:1329-1358): base < 0 and exponent != trunc(exponent)
            Val zero; zero.imm = true; zero.bits = 0; zero.width = 8;
            Val neg; neg.width = 1; neg.reg = binop(VM_LT_F64, a[0], zero, 1);
            Val tr; tr.width = 8; tr.reg = unop(VM_TRUNC_F64, a[1], 8);
            Val frac; frac.width = 1; frac.reg = binop(VM_NE_F64, a[1], tr, 1);
            Val bad; bad.width = 1; bad.reg = binop(VM_AND_B8, neg, frac, 1);
            if (e->op == OP_POW_NULLING) {
              if (base_null >= 0) { int r = new_reg(1); LInstr& i = emit(VM_NULL_OR); i.dst = r; i.a = base_null; i.b = bad.reg; base_null = r; }
              else base_null = bad.reg;
            } else {
              LInstr& i = emit(VM_FAIL_TRUE_8); i.dst_is_reg = false; i.dst = 0; i.a = base_null; i.b = bad.reg; i.c = sel;
            }
          }
          const int x = materialize(a[0]), y = materialize(a[1]);
          v.reg = new_reg(8);
          { LInstr& i = emit(VM_MATH2_F64); i.dst = v.reg; i.a = x; i.b = y; i.imm = e->op == OP_ATAN2 ? VM_MATH_ATAN2 : VM_MATH_POW; }
          P->uses_math = true;
          v.null = base_null;
        } break;
        case OP_SQRT_QUIET: case OP_SQRT_NULLING: case OP_SQRT_SIGNALING: {
          int base_null = a[0].null;
          if (e->op != OP_SQRT_QUIET) {
            // IsNegativeNuller / IsNegativeFailer (expression_traits.h:922-947): x < 0
            Val zero; zero.imm = true; zero.bits = 0; zero.width = 8;
            Val neg; neg.width = 1; neg.reg = binop(VM_LT_F64, a[0], zero, 1);
            if (e->op == OP_SQRT_NULLING) {
              if (base_null >= 0) { int r = new_reg(1); LInstr& i = emit(VM_NULL_OR); i.dst = r; i.a = base_null; i.b = neg.reg; base_null = r; }
              else base_null = neg.reg;
            } else {
              LInstr& i = emit(VM_FAIL_TRUE_8); i.dst_is_reg = false; i.dst = 0; i.a = base_null; i.b = neg.reg; i.c = sel;
            }
          }
          v.reg = unop(VM_SQRT_F64, a[0], v.width);
          v.null = base_null;
        } break;
        case OP_NEGATE: v.reg = unop(pick(mt, VM_NEG_I32, VM_NEG_I32, VM_NEG_I64, VM_NEG_I64, VM_NEG_F32, VM_NEG_F64), a[0], v.width); v.null = a[0].null; break;
        case OP_BITWISE_AND: v.reg = binop(v.width == 4 ? VM_BAND_32 : VM_BAND_64, a[0], a[1], v.width); v.null = or_null(a[0].null, a[1].null); break;
        case OP_BITWISE_OR: v.reg = binop(v.width == 4 ? VM_BOR_32 : VM_BOR_64, a[0], a[1], v.width); v.null = or_null(a[0].null, a[1].null); break;
        case OP_BITWISE_XOR: v.reg = binop(v.width == 4 ? VM_BXOR_32 : VM_BXOR_64, a[0], a[1], v.width); v.null = or_null(a[0].null, a[1].null); break;
        case OP_BITWISE_ANDNOT: v.reg = binop(v.width == 4 ? VM_BANDNOT_32 : VM_BANDNOT_64, a[0], a[1], v.width); v.null = or_null(a[0].null, a[1].null); break;
        case OP_BITWISE_NOT: v.reg = unop(v.width == 4 ? VM_BNOT_32 : VM_BNOT_64, a[0], v.width); v.null = a[0].null; break;
        case OP_SHIFT_LEFT: case OP_SHIFT_RIGHT: {
          // the count keeps its own integer type in the bound tree (the result inherits the left type):
          // bring it to the left operand's width; valid counts (< width) survive any such conversion
          Val cnt = a[1];
          const MT ct = mtype(e->args[1]->dtype);
          if (cnt.width != v.width) SS_RETURN_IF_ERROR(cast_val(a[1], ct, v.width == 8 ? M_I64 : M_I32, &cnt));
          const uint16_t vop = e->op == OP_SHIFT_LEFT ? (v.width == 4 ? VM_SHL_I32 : VM_SHL_I64)
                                                      : pick(mt, VM_SHR_I32, VM_SHR_U32, VM_SHR_I64, VM_SHR_U64, VM_NOP, VM_NOP);
          v.reg = binop(vop, a[0], cnt, v.width); v.null = or_null(a[0].null, a[1].null);
        } break;
        case OP_EQUAL: case OP_NOT_EQUAL: case OP_LESS: case OP_LESS_OR_EQUAL: {
          MT lt = at, rt = mtype(e->args[1]->dtype);
          Val l = a[0], r = a[1];
          uint16_t op = VM_NOP;
          const int c = e->op == OP_LESS ? 0 : e->op == OP_LESS_OR_EQUAL ? 1 : e->op == OP_EQUAL ? 2 : 3;
          if (lt == rt) {
            static const uint16_t T[4][7] = {
                {VM_LT_I32, VM_LT_U32, VM_LT_I64, VM_LT_U64, VM_LT_F32, VM_LT_F64, VM_LT_B8},
                {VM_LE_I32, VM_LE_U32, VM_LE_I64, VM_LE_U64, VM_LE_F32, VM_LE_F64, VM_LE_B8},
                {VM_EQ_32, VM_EQ_32, VM_EQ_64, VM_EQ_64, VM_EQ_F32, VM_EQ_F64, VM_EQ_B8},
                {VM_NE_32, VM_NE_32, VM_NE_64, VM_NE_64, VM_NE_F32, VM_NE_F64, VM_NE_B8}};
            op = T[c][lt];
          } else {
            // two different integer types: widen to 64 bits, pick the sign-correct compare
            auto is_int = [](MT m) { return m == M_I32 || m == M_U32 || m == M_I64 || m == M_U64; };
            if (!is_int(lt) || !is_int(rt)) return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH, "cannot compare these types");
            const bool l_u = lt == M_U64 || (lt == M_U32 && rt == M_U64);
            const bool r_u = rt == M_U64 || (rt == M_U32 && lt == M_U64);
            Val lw, rw;
            SS_RETURN_IF_ERROR(cast_val(l, lt, l_u ? M_U64 : M_I64, &lw));
            SS_RETURN_IF_ERROR(cast_val(r, rt, r_u ? M_U64 : M_I64, &rw));
            l = lw; r = rw;
            if (!l_u && !r_u) op = c == 0 ? VM_LT_I64 : c == 1 ? VM_LE_I64 : c == 2 ? VM_EQ_64 : VM_NE_64;
            else if (l_u && r_u) op = c == 0 ? VM_LT_U64 : c == 1 ? VM_LE_U64 : c == 2 ? VM_EQ_64 : VM_NE_64;
            else if (!l_u && r_u) op = c == 0 ? VM_LT_I64_U64 : c == 1 ? VM_LE_I64_U64 : c == 2 ? VM_EQ_I64_U64 : VM_NE_I64_U64;
            else {
              if (c >= 2) { std::swap(l, r); op = c == 2 ? VM_EQ_I64_U64 : VM_NE_I64_U64; }
              else op = c == 0 ? VM_LT_U64_I64 : VM_LE_U64_I64;
            }
          }
          v.width = 1;
          v.reg = binop(op, l, r, 1);
          v.null = or_null(a[0].null, a[1].null);
        } break;
        case OP_AND: case OP_OR: case OP_AND_NOT: {
          Val l = a[0], r = a[1];
          if (e->op == OP_AND_NOT) { l.reg = unop(VM_NOT_B8, l, 1); l.imm = false; }  // andnot(a,b) = (!a) && b
          const bool is_or = e->op == OP_OR;
          if (l.null < 0 && r.null < 0) {
            v.reg = binop(is_or ? VM_OR_B8 : VM_AND_B8, l, r, 1);
          } else {
            int lr = materialize(l), rr = materialize(r);
            v.reg = new_reg(1); v.null = new_reg(1);
            LInstr& i = emit(is_or ? VM_OR3 : VM_AND3);
            i.dst = v.reg; i.c = v.null; i.a = lr; i.b = rr; i.d = l.null; i.e = r.null;
          }
        } break;
        case OP_XOR: v.reg = binop(VM_XOR_B8, a[0], a[1], 1); v.null = or_null(a[0].null, a[1].null); break;
        case OP_NOT: v.reg = unop(VM_NOT_B8, a[0], 1); v.null = a[0].null; break;
        case OP_IS_NULL:
          if (a[0].null < 0) { v.imm = true; v.bits = 0; } else { v.reg = a[0].null; }
          break;
        case OP_IF_NULL: {  // a NULL ? b : a
          if (a[0].null < 0) {   // NULLABLE by schema but never NULL (e.g. a CASE whose chain collapsed): the value itself
            v = a[0];
            if (v.imm) { v.reg = materialize(v); v.imm = false; }
            break;
          }
          const uint16_t sop = v.width == 8 ? VM_SELECT_64 : v.width == 4 ? VM_SELECT_32 : VM_SELECT_8;
          Val x = a[1], y = a[0];
          if (x.imm && y.imm) { x.reg = materialize(x); x.imm = false; }
          v.reg = new_reg(v.width);
          LInstr& i = emit(sop); i.dst = v.reg; i.c = a[0].null;
          if (x.imm) { i.a_imm = true; i.imm = x.bits; i.imm_width = (uint8_t)x.width; } else i.a = x.reg;
          if (y.imm) { i.b_imm = true; i.imm = y.bits; i.imm_width = (uint8_t)y.width; } else i.b = y.reg;
          if (a[1].null >= 0) {
            v.null = new_reg(1);
            LInstr& j = emit(VM_SELECT_8); j.dst = v.null; j.a = a[1].null; j.b_imm = true; j.imm = 0; j.imm_width = 1; j.c = a[0].null;
          }
        } break;
        case OP_IF: case SSGPU_OP_NULLING_IF: {
          const uint16_t sop = v.width == 8 ? VM_SELECT_64 : v.width == 4 ? VM_SELECT_32 : VM_SELECT_8;
          // choice = condition is non-NULL TRUE; a NULL condition takes the ELSE branch
          int cond = materialize(a[0]);
          if (a[0].null >= 0) {
            int ch = new_reg(1);
            LInstr& c = emit(VM_SEL_FROM_PRED); c.dst = ch; c.a = cond; c.b = a[0].null;
            cond = ch;
          }
          Val x = a[1], y = a[2];
          if (x.imm && y.imm) { x.reg = materialize(x); x.imm = false; }
          v.reg = new_reg(v.width);
          LInstr& i = emit(sop); i.dst = v.reg; i.c = cond;
          if (x.imm) { i.a_imm = true; i.imm = x.bits; i.imm_width = (uint8_t)x.width; } else i.a = x.reg;
          if (y.imm) { i.b_imm = true; i.imm = y.bits; i.imm_width = (uint8_t)y.width; } else i.b = y.reg;
          if (a[1].null >= 0 || a[2].null >= 0) {
            v.null = new_reg(1);
            LInstr& j = emit(VM_SELECT_8); j.dst = v.null; j.c = cond;
            if (a[1].null >= 0) j.a = a[1].null; else { j.a_imm = true; j.imm = 0; j.imm_width = 1; }
            if (a[2].null >= 0) j.b = a[2].null; else { j.b_imm = true; j.imm = 0; j.imm_width = 1; }
          }
          if (e->op == SSGPU_OP_NULLING_IF) v.null = or_null(v.null, a[0].null);   // a NULL condition is a NULL result
        } break;
        default:
          return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "operator has no device lowering: " + e->name);
      }
    } break;
  }
  memo_[key] = v;
  *out = v;
  return Status::OK();
}

// ---- register allocation: linear scan over LDS row offsets ---------------------
static void for_each_use(const LInstr& i, const std::function<void(int)>& f) {
  if (!i.a_imm && i.a >= 0) f(i.a);
  const bool fused_agg = i.op >= VM_AGG_SUM_I64_ADD && i.op <= VM_AGG_SUM_F64_MUL;
  if ((!i.b_imm || fused_agg) && i.b >= 0) f(i.b);
  if (i.d >= 0) f(i.d);
  if (i.e >= 0) f(i.e);
  const bool c_is_def = i.op == VM_AND3 || i.op == VM_OR3;
  if (i.c >= 0 && !c_is_def) f(i.c);
  const bool dst_is_use = i.op == VM_KEY_APPEND_8 || i.op == VM_KEY_APPEND_32 || i.op == VM_KEY_APPEND_64;
  if (i.dst_is_reg && dst_is_use && i.dst >= 0) f(i.dst);
}
static void for_each_def(const LInstr& i, const std::function<void(int)>& f) {
  if (i.dst_is_reg && i.dst >= 0) f(i.dst);
  if ((i.op == VM_AND3 || i.op == VM_OR3) && i.c >= 0) f(i.c);
}

static void allocate_registers(Program* p) {
  const int n = (int)p->regs.size();
  std::vector<int> first(n, 1 << 30), last(n, -1);
  for (auto& s : p->staged) first[s.reg] = -1;
  for (int pc = 0; pc < (int)p->code.size(); ++pc) {
    for_each_def(p->code[pc], [&](int r) { first[r] = std::min(first[r], pc); last[r] = std::max(last[r], pc); });
    for_each_use(p->code[pc], [&](int r) { last[r] = std::max(last[r], pc); first[r] = std::min(first[r], pc); });
  }
  // Free lists PER WIDTH CLASS.  Elementwise instructions need no barrier because a thread
  // only ever touches the bytes of its own row pairs -- which bytes those are depends on the
  // register width, so a slot may only be recycled by a register of the SAME width (waves are
  // not in lockstep between instructions: a recycled slot of another width would let one wave
  // overwrite bytes another wave still has to read).
  std::map<uint32_t, std::vector<uint32_t>> free_slots;
  uint32_t top = 0, peak = 0, in_bpr = 0;
  std::vector<bool> placed(n, false);
  auto place = [&](int r) {
    if (placed[r]) return;
    placed[r] = true;
    const uint32_t w = p->regs[r].width;
    auto& fl = free_slots[w];
    if (!fl.empty()) { p->regs[r].row_off = fl.back(); fl.pop_back(); return; }
    p->regs[r].row_off = top; top += w; peak = std::max(peak, top);
  };
  auto release = [&](int r) { free_slots[p->regs[r].width].push_back(p->regs[r].row_off); };
  // staged registers live from the start (wide ones first)
  std::vector<int> st;
  for (auto& s : p->staged) st.push_back(s.reg);
  std::stable_sort(st.begin(), st.end(), [&](int a, int b) { return p->regs[a].width > p->regs[b].width; });
  for (int r : st) place(r);
  in_bpr = top;
  for (int pc = 0; pc < (int)p->code.size(); ++pc) {
    for_each_def(p->code[pc], [&](int r) { place(r); });
    // a register whose last use is this instruction is released AFTER the definition was placed
    std::vector<int> dead;
    for_each_use(p->code[pc], [&](int r) { if (last[r] == pc) dead.push_back(r); });
    for_each_def(p->code[pc], [&](int r) { if (last[r] == pc) dead.push_back(r); });
    std::sort(dead.begin(), dead.end()); dead.erase(std::unique(dead.begin(), dead.end()), dead.end());
    for (int r : dead) release(r);
  }
  p->bytes_per_row = std::max(peak, in_bpr);
  p->in_bytes_per_row = in_bpr;
}

ProgramLayout layout_program(const Program& p, const LowerOptions& opt) {
  ProgramLayout L;
  auto lds_for = [&](int K, uint32_t* acc, uint32_t* scr) {
    // input registers + temporaries
    uint32_t regs = p.bytes_per_row * (uint32_t)VM_TILE_UNIT * (uint32_t)K;
    uint32_t a = (regs + 15u) & ~15u;
    uint32_t s = a + (uint32_t)p.n_slots * VM_ACC_STRIDE;
    if (acc) *acc = a;
    if (scr) *scr = s;
    return s + 256u + 16u * (uint32_t)p.code.size()   // + the immediates' constant pool
           + 2u * (uint32_t)VM_TILE_UNIT * (uint32_t)K;                   // + all-ones / all-zeros mask arrays of one tile
  };
  int K = 1;
  if (opt.tile_rows > 0) {
    K = std::max(1, opt.tile_rows / VM_TILE_UNIT);
    if (K >= 4) K = 4; else if (K >= 2) K = 2; else K = 1;
    while (K > 1 && lds_for(K, nullptr, nullptr) > 160u * 1024u) K /= 2;  // must fit one CU
  } else {
    // the largest tile that fits the LDS target AND whose staged units fit the kernel's
    // register prefetch file (units beyond it are fetched with their latency exposed)
    for (int cand : {4, 2, 1}) {
      K = cand;
      if ((int)lds_for(cand, nullptr, nullptr) <= opt.lds_target_bytes && (int)p.staged.size() * cand <= VM_PF_UNITS) break;
    }
  }
  L.K = K;
  L.lds_bytes = lds_for(K, &L.acc_off, &L.scratch_off);
  L.imm_pool_off = L.scratch_off + 256u;
  return L;
}

void finalize_program(const Program& p, const ProgramLayout& L, std::vector<VmInstr>* out) {
  out->clear();
  const uint32_t T = (uint32_t)VM_TILE_UNIT * (uint32_t)L.K;
  // VM register -> LDS byte offset: a register is `row_off` bytes per row, i.e. an array at row_off * T
  auto off = [&](int r) -> uint32_t { return r < 0 ? VM_NONE : p.regs[r].row_off * T; };
  for (size_t pc = 0; pc < p.code.size(); ++pc) {
    const LInstr& i = p.code[pc];
    VmInstr v; memset(&v, 0, sizeof(v));
    v.op = i.op;
    v.reg_ops = (uint8_t)((i.a_imm ? 0 : 1) | (i.b_imm ? 0 : 2));
    v.imm_width = (i.a_imm || i.b_imm) ? i.imm_width : 0;
    v.dst = i.dst_is_reg ? off(i.dst) : (uint32_t)i.dst;
    const uint32_t pool = L.imm_pool_off + 16u * (uint32_t)pc;  // this instruction's constant
    v.a = i.a_imm ? pool : off(i.a);
    v.b = i.b_imm ? pool : off(i.b);
    v.c = off(i.c); v.d = off(i.d);
    const bool fused_agg = i.op >= VM_AGG_SUM_I64_ADD && i.op <= VM_AGG_SUM_F64_MUL;
    if (fused_agg) {  // operands a and d; b is the null mask; b_imm flags operand d
      v.b = off(i.b);
      if (i.b_imm) v.d = pool;
    }
    v.imm = i.imm;
    if (i.op == VM_AND3 || i.op == VM_OR3) v.imm = (uint64_t)off(i.d) | ((uint64_t)off(i.e) << 32);
    out->push_back(v);
  }
  // one trailing NOP: the kernel prefetches instruction pc + 1
  VmInstr nop; memset(&nop, 0, sizeof(nop)); nop.op = VM_NOP;
  nop.dst = nop.a = nop.b = nop.c = nop.d = VM_NONE;
  out->push_back(nop);
}

std::string disassemble(const Program& p) {
  std::ostringstream s;
  s << "  staged:";
  for (auto& st : p.staged) s << " r" << st.reg << "<-col" << st.col << (st.is_null_mask ? ".null" : "") << "(w" << p.regs[st.reg].width << ")";
  s << "\n  lds bytes/row: " << p.bytes_per_row << ", slots: " << p.n_slots << ", outputs: " << p.n_outputs << "\n";
  auto R = [&](int r) { return r < 0 ? std::string("-") : "r" + std::to_string(r) + "@" + std::to_string(p.regs[r].row_off); };
  for (size_t pc = 0; pc < p.code.size(); ++pc) {
    const LInstr& i = p.code[pc];
    s << "  " << pc << ": " << vm_op_name(i.op) << " dst=" << (i.dst_is_reg ? R(i.dst) : "#" + std::to_string(i.dst))
      << " a=" << (i.a_imm ? "imm" : R(i.a)) << " b=" << (i.b_imm ? "imm" : R(i.b)) << " c=" << R(i.c) << " d=" << R(i.d);
    if (i.e >= 0) s << " e=" << R(i.e);
    if (i.a_imm || i.b_imm || i.imm) s << " imm=0x" << std::hex << i.imm << std::dec;
    s << "\n";
  }
  return s.str();
}

// ---- expression composition -------------------------------------------------------
struct VCol { BExprP expr; std::string name; };

static BExprP substitute(const BExprP& e, const std::vector<VCol>& cols, std::map<const BExpr*, BExprP>* memo) {
  auto it = memo->find(e.get());
  if (it != memo->end()) return it->second;
  BExprP r;
  if (e->kind == BExpr::INPUT) {
    r = cols[e->input_col].expr;
  } else if (e->args.empty()) {
    r = e;
  } else {
    r.reset(new BExpr(*e));
    for (auto& a : r->args) a = substitute(a, cols, memo);
  }
  (*memo)[e.get()] = r;
  return r;
}

static Schema schema_of(const std::vector<VCol>& cols) {
  Schema s;
  for (auto& c : cols) { Attr a; a.name = c.name; a.dtype = c.expr->dtype; a.nullable = c.expr->nullable; s.push_back(a); }
  return s;
}

static int lookup_pos(const Schema& s, const std::string& name) {
  for (size_t i = 0; i < s.size(); ++i) if (s[i].name == name) return (int)i;
  return -1;
}

// ---- aggregate specification binding (aggregator.cc:63-186) --------------------------
enum { AGG_FIRST_SEEN = 9001 };   // internal: the group's smallest global row id (UINT64, never NULL) -- first-seen order of the groups
struct AggPlan {
  int aggregation = 0;
  int input_pos = -1;  // -1 for COUNT(*)
  int out_type = 0;
  std::string out_name;
  bool result_nullable = true;
  bool distinct = false;   // only the first occurrence of every value of a group contributes (column_aggregator.cc:308-376)
  int flag_pos = -1;       // DISTINCT over a second, third ... column: pipe column holding that column's first-of-run flag
  int order_pos = -1;      // FIRST / LAST over rows that were re-ordered on the way (the DISTINCT shape): pipe column holding the
                           // row's id in the ORIGINAL order -- the aggregate picks by it, not by the row's position here
  bool sequential = false; // SUM of a floating input into an integer result: folded row after row in the input order (Stage::seq_sums)
  bool concat_distinct = false;  // DISTINCT CONCAT: the host prints a value once per result row (Stage::ConcatCol::distinct)
  bool rowid_only = false; // FIRST / LAST that yields the chosen row's id (UINT64) instead of its value: the arg-min / arg-max
                           // column the fold of a key limit picks values by
};

static Status bind_aggregations(const PlanDesc& d, int first, int n, const Schema& in, std::vector<AggPlan>* out) {
  for (int i = 0; i < n; ++i) {
    const ssgpu_agg& a = d.aggs[first + i];
    AggPlan p;
    p.aggregation = a.aggregation;
    p.out_name = a.output;
    if (a.aggregation == SSGPU_COUNT && !a.distinct && a.input[0] == 0) {
      p.input_pos = -1;
    } else {
      p.input_pos = lookup_pos(in, a.input);
      if (p.input_pos < 0)
        return Status::Error(SSGPU_ERROR_ATTRIBUTE_MISSING,
                             std::string("Incorrect aggregation specification. Aggregation input column does not exist: ") + a.input + ".");
    }
    if (a.output_type >= 0) p.out_type = a.output_type;
    else if (a.aggregation == SSGPU_COUNT) p.out_type = SSGPU_UINT64;
    else p.out_type = in[p.input_pos].dtype;
    p.result_nullable = a.aggregation != SSGPU_COUNT;
    for (auto& q : *out)
      if (q.out_name == p.out_name)
        return Status::Error(SSGPU_ERROR_ATTRIBUTE_EXISTS,
                             "Incorrect aggregation specification. Aggregation output column name is non-unique: '" + p.out_name + "'.");
    // DISTINCT changes SUM and COUNT only: the MIN / MAX / FIRST / LAST of the distinct values are those of all values
    p.distinct = a.distinct && (a.aggregation == SSGPU_SUM || a.aggregation == SSGPU_COUNT);
    if (a.aggregation == SSGPU_CONCAT) {
      // CONCAT -> STRING over every type with a PrintTyped form (column_aggregator.cc:496-505).  The values are ordered on the
      // device and printed on the host (Stage::ConcatCol: DATE / DATETIME in the reference's strftime forms).  DISTINCT CONCAT prints
      // a value at its first occurrence in the result row only (the DistinctAggregator in front of the CONCAT, :308-376).
      const int it = in[p.input_pos].dtype;
      if (a.output_type >= 0 && a.output_type != SSGPU_STRING)
        return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_TYPE, std::string("Aggregation not supported. Aggregation function not defined for types ") +
                                                                    dtype_name(it) + " and " + dtype_name(a.output_type) + ".");
      if (it == SSGPU_BINARY || dtype_width(it) == 0)
        return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "CONCAT of BINARY values is outside the device path");
      p.out_type = SSGPU_STRING; p.result_nullable = true; p.distinct = false; p.concat_distinct = a.distinct != 0;
      out->push_back(p);
      continue;
    }
    if (a.aggregation == SSGPU_SUM_RESIDUAL) {   // extension (ssgpu.h): the exact residual of the DOUBLE SUM of the same column
      bool found = false;
      for (auto& q : *out) found = found || (q.aggregation == SSGPU_SUM && q.input_pos == p.input_pos && q.out_type == SSGPU_DOUBLE && !q.distinct);
      if (a.distinct || in[p.input_pos].dtype != SSGPU_DOUBLE || p.out_type != SSGPU_DOUBLE || !found)
        return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_TYPE, "SUM_RESIDUAL follows the (non-DISTINCT) DOUBLE SUM of the same DOUBLE column in the same specification");
      out->push_back(p);
      continue;
    }
    // supported matrix (column_aggregator.cc:484-532)
    if (a.aggregation == SSGPU_COUNT) {
      if (!dtype_is_integer(p.out_type))
        return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_TYPE, std::string("COUNT output type must be an integer, got ") + dtype_name(p.out_type));
    } else {
      const int it = in[p.input_pos].dtype;
      bool ok = (dtype_is_numeric(it) && dtype_is_numeric(p.out_type)) ||
                (it == p.out_type && a.aggregation != SSGPU_SUM && (it == SSGPU_BOOL || it == SSGPU_DATE || it == SSGPU_DATETIME || it == SSGPU_STRING));
      if (!ok)
        return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_TYPE,
                             std::string("Aggregation not supported. Aggregation function not defined for types ") +
                                 dtype_name(it) + " and " + dtype_name(p.out_type) + ".");
      // floating input into an integer result: MIN / MAX / FIRST / LAST store the truncated value, and because truncation is
      // monotone the reference's fold (compare the floating value with the integer result, store the cast) gives
      // min / max of the truncated values whatever the order.  SUM adds a floating value to an integer result and
      // truncates after EVERY row: order-dependent -- the rows are folded one after the other in the input order (take_sequential).
      if (dtype_is_float(it) && dtype_is_integer(p.out_type) && a.aggregation == SSGPU_SUM) {
        if (a.distinct) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "SUM DISTINCT of a floating input into an integer output is not available on the device path");
        p.sequential = true;
      }
    }
    out->push_back(p);
  }
  return Status::OK();
}

// scalar-aggregate instruction selection
struct AggSel { uint16_t op; int slot_kind; int emit_kind; };
static bool select_scalar_agg(int aggregation, int out_type, AggSel* s) {
  const MT m = mtype(out_type);
  switch (aggregation) {
    case SSGPU_SUM:
      switch (m) {
        case M_I32: *s = {VM_AGG_SUM_I32, SLOT_SUM_INT, EMIT_U32}; return true;
        case M_U32: *s = {VM_AGG_SUM_U32, SLOT_SUM_INT, EMIT_U32}; return true;
        case M_I64: case M_U64: *s = {VM_AGG_SUM_I64, SLOT_SUM_INT, EMIT_U64}; return true;
        case M_F32: *s = {VM_AGG_SUM_F32, SLOT_SUM_DD, EMIT_DD_F32}; return true;
        case M_F64: *s = {VM_AGG_SUM_F64, SLOT_SUM_DD, EMIT_DD_F64}; return true;
        default: return false;
      }
    case SSGPU_MIN: case SSGPU_MAX: {
      const bool mn = aggregation == SSGPU_MIN;
      switch (m) {
        case M_I32: *s = {mn ? VM_AGG_MIN_I32 : VM_AGG_MAX_I32, mn ? SLOT_MIN_U64 : SLOT_MAX_U64, EMIT_I32KEY}; return true;
        case M_U32: *s = {mn ? VM_AGG_MIN_U32 : VM_AGG_MAX_U32, mn ? SLOT_MIN_U64 : SLOT_MAX_U64, EMIT_U32}; return true;
        case M_I64: *s = {mn ? VM_AGG_MIN_I64 : VM_AGG_MAX_I64, mn ? SLOT_MIN_U64 : SLOT_MAX_U64, EMIT_I64KEY}; return true;
        case M_U64: *s = {mn ? VM_AGG_MIN_U64 : VM_AGG_MAX_U64, mn ? SLOT_MIN_U64 : SLOT_MAX_U64, EMIT_U64}; return true;
        case M_F32: *s = {mn ? VM_AGG_MIN_F32 : VM_AGG_MAX_F32, mn ? SLOT_MIN_F64 : SLOT_MAX_F64, EMIT_F32}; return true;
        case M_F64: *s = {mn ? VM_AGG_MIN_F64 : VM_AGG_MAX_F64, mn ? SLOT_MIN_F64 : SLOT_MAX_F64, EMIT_F64}; return true;
        case M_B8: *s = {mn ? VM_AGG_MIN_B8 : VM_AGG_MAX_B8, mn ? SLOT_MIN_U64 : SLOT_MAX_U64, EMIT_U8}; return true;
        default: return false;
      }
    }
    case SSGPU_FIRST: case SSGPU_LAST: {
      const bool f = aggregation == SSGPU_FIRST;
      const uint32_t w = mwidth(m);
      *s = {(uint16_t)(w == 8 ? (f ? VM_AGG_FIRST_64 : VM_AGG_LAST_64) : w == 4 ? (f ? VM_AGG_FIRST_32 : VM_AGG_LAST_32)
                                                                              : (f ? VM_AGG_FIRST_8 : VM_AGG_LAST_8)),
            f ? SLOT_FIRST : SLOT_LAST, w == 8 ? EMIT_U64 : w == 4 ? EMIT_U32 : EMIT_U8};
      return m != M_BAD;
    }
  }
  return false;
}
// MIN / MAX from one integer type into another (AddAggregationWithDefinedOutputType): the reference compares every value IN
// ITS OWN TYPE with the running result (ThreeWayCompare<InputType, OutputType>, aggregation_operators.h:187-228;
// aggregation_operators_test.cc:200-210 is the regression test: MAX of INT32 {1, -1} into UINT32 is 1) and stores the cast.
// Here the extremum is found in the input's domain widened to 64 bits and truncated to the result type by the emit step
// (the accumulator keys of all integer MIN / MAX are 64-bit): equal to the reference whenever the result type can hold the
// values; where it cannot, the reference's result depends on the row order (include/ssgpu.h).
static bool minmax_in_own_type(int aggregation, int in_type, int out_type, int* agg_type, int* emit_kind) {
  if (aggregation != SSGPU_MIN && aggregation != SSGPU_MAX) return false;
  const MT in = mtype(in_type), out = mtype(out_type);
  auto integer = [](MT m) { return m == M_I32 || m == M_U32 || m == M_I64 || m == M_U64; };
  if (in == out || !integer(in) || !integer(out) || in_type == SSGPU_STRING || out_type == SSGPU_STRING) return false;
  const bool uns = in == M_U64;
  *agg_type = uns ? SSGPU_UINT64 : SSGPU_INT64;
  *emit_kind = mwidth(out) == 8 ? (uns ? EMIT_U64 : EMIT_I64KEY) : (uns ? EMIT_U32 : EMIT_I32KEY);
  return true;
}
static bool select_group_agg(int aggregation, int out_type, AggSel* s, uint64_t* init) {
  const MT m = mtype(out_type);
  *init = 0;
  switch (aggregation) {
    case SSGPU_SUM:
      switch (m) {
        case M_I32: *s = {VM_GAGG_SUM_I32, 0, EMIT_U32}; return true;
        case M_U32: *s = {VM_GAGG_SUM_U32, 0, EMIT_U32}; return true;
        case M_I64: case M_U64: *s = {VM_GAGG_SUM_I64, 0, EMIT_U64}; return true;
        case M_F32: *s = {VM_GAGG_SUM_F32, 0, EMIT_F32}; return true;
        case M_F64: *s = {VM_GAGG_SUM_F64, 0, EMIT_F64}; return true;
        default: return false;
      }
    case SSGPU_MIN: case SSGPU_MAX: {
      const bool mn = aggregation == SSGPU_MIN;
      *init = mn ? ~0ull : 0ull;
      switch (m) {
        case M_I32: *s = {mn ? VM_GAGG_MIN_I32 : VM_GAGG_MAX_I32, 0, EMIT_I32KEY}; return true;
        case M_U32: *s = {mn ? VM_GAGG_MIN_U32 : VM_GAGG_MAX_U32, 0, EMIT_U32}; return true;
        case M_I64: *s = {mn ? VM_GAGG_MIN_I64 : VM_GAGG_MAX_I64, 0, EMIT_I64KEY}; return true;
        case M_U64: *s = {mn ? VM_GAGG_MIN_U64 : VM_GAGG_MAX_U64, 0, EMIT_U64}; return true;
        case M_F32: *s = {mn ? VM_GAGG_MIN_F32 : VM_GAGG_MAX_F32, 0, EMIT_FKEY_F32}; return true;
        case M_F64: *s = {mn ? VM_GAGG_MIN_F64 : VM_GAGG_MAX_F64, 0, EMIT_FKEY_F64}; return true;
        case M_B8: *s = {mn ? VM_GAGG_MIN_B8 : VM_GAGG_MAX_B8, 0, EMIT_U8}; return true;
        default: return false;
      }
    }
  }
  return false;
}

// ---- one pipeline being assembled -----------------------------------------------------
struct Pipe {
  Schema in_schema;                // stage input
  std::vector<VCol> cols;          // current virtual schema over the stage input
  std::vector<BExprP> filters;     // predicates in order; filter i was bound at depth i
  std::vector<JoinSpec> joins;     // HashJoins fused into this pipeline
  int depth() const { return (int)filters.size(); }
};

static void reset_pipe(Pipe* p, const Schema& in) {
  p->in_schema = in; p->cols.clear(); p->filters.clear(); p->joins.clear();
  for (size_t i = 0; i < in.size(); ++i) {
    BExprP e(new BExpr);
    e->kind = BExpr::INPUT; e->input_col = (int)i; e->dtype = in[i].dtype; e->nullable = in[i].nullable; e->name = in[i].name;
    VCol c; c.expr = e; c.name = in[i].name;
    p->cols.push_back(c);
  }
}

// emit the filter chain; fills em.sel_by_depth
static Status emit_filters(Emitter& em, const Pipe& pipe) {
  em.filters_ = &pipe.filters;
  em.hoist_ok_ = !pipe.joins.empty();
  for (auto& f : pipe.filters) em.hoist_ok_ = em.hoist_ok_ && !em.may_fail(f);
  for (auto& c : pipe.cols) em.hoist_ok_ = em.hoist_ok_ && !em.may_fail(c.expr);
  for (auto& j : pipe.joins) for (auto& k : j.lhs_keys) em.hoist_ok_ = em.hoist_ok_ && !em.may_fail(k);
  for (size_t f = 0; f < pipe.filters.size(); ++f) {
    Val pv;
    SS_RETURN_IF_ERROR(em.value(pipe.filters[f], &pv));
    if (!pv.imm && pv.null < 0 && em.sel_by_depth.back() < 0) {
      // first filter, never-NULL predicate: its 0/1 byte vector IS the selection
      em.sel_by_depth.push_back(pv.reg);
      continue;
    }
    int r = em.new_reg(1);
    LInstr& i = em.emit(VM_SEL_FROM_PRED);
    i.dst = r;
    if (pv.imm) { i.a_imm = true; i.imm = pv.bits; i.imm_width = 1; } else i.a = pv.reg;
    i.b = pv.null;
    i.c = em.sel_by_depth.back();
    em.sel_by_depth.push_back(r);
  }
  return Status::OK();
}

static int64_t staged_bytes(const Program& p) {
  int64_t b = 0;
  for (auto& s : p.staged) b += p.regs[s.reg].width;
  return b;
}

// distinct_flag_input >= 0: index of a synthetic BOOL input column (appended by the runtime) that is 1 on the first row of
// every run of equal (group keys, distinct column) in the -- sorted -- stage input; a DISTINCT aggregate treats every other
// row like a NULL input
static bool has_sequential(const std::vector<AggPlan>& plans) { for (auto& ap : plans) if (ap.sequential) return true; return false; }
// Sequential sums (AggPlan::sequential) leave the stage's program: each becomes COUNT(column) -- its slot, its result type -- and
// an entry of Stage::seq_sums; the runtime overwrites the count with the row-after-row fold.  `pipe` must be the identity
// over stored columns (the callers materialise first), `first_out` = the result column of plans[0].
static Status take_sequential(std::vector<AggPlan>* plans, const Pipe& pipe, size_t first_out, std::vector<Stage::SeqSum>* out) {
  for (size_t j = 0; j < plans->size(); ++j) {
    AggPlan& ap = (*plans)[j];
    if (!ap.sequential) continue;
    const BExprP& src = pipe.cols[ap.input_pos].expr;
    if (src->kind != BExpr::INPUT) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "SUM of a floating value into an integer output needs a stored input column");
    out->push_back(Stage::SeqSum{(int)(first_out + j), src->input_col});
    ap.aggregation = SSGPU_COUNT; ap.sequential = false;
  }
  return Status::OK();
}

static Status finish_scalar_agg_bound(const std::vector<AggPlan>& plans, const Pipe& pipe, Stage* st, int distinct_flag_input = -1) {
  for (auto& ap : plans)
    if (ap.aggregation == SSGPU_SUM_RESIDUAL)
      return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "SUM_RESIDUAL belongs to group aggregates (a sharded ScalarAggregate carries its double-double words in the partial state)");
  st->kind = STAGE_SCALAR_AGG;
  st->in_schema = pipe.in_schema;
  if ((int)plans.size() > VM_MAX_AGG_SLOTS) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "too many aggregations for one pipeline");
  st->joins = pipe.joins;
  Emitter em(&st->main, &pipe.joins);
  SS_RETURN_IF_ERROR(emit_filters(em, pipe));
  const int sel = em.sel_by_depth.back();
  int notfirst = -1;
  if (distinct_flag_input >= 0) { Val f; f.width = 1; f.reg = em.staged(distinct_flag_input, false, 1); notfirst = em.unop(VM_NOT_B8, f, 1); }
  std::map<int, int> notfirst_of_col;   // materialised flag columns of further DISTINCT inputs (AggPlan::flag_pos)
  auto notfirst_for = [&](const AggPlan& ap, int* out) -> Status {
    *out = notfirst;
    if (ap.flag_pos < 0) return Status::OK();
    auto it = notfirst_of_col.find(ap.flag_pos);
    if (it == notfirst_of_col.end()) {
      Val f; SS_RETURN_IF_ERROR(em.value(pipe.cols[ap.flag_pos].expr, &f));
      Val fm; fm.width = 1; fm.reg = em.materialize(f);
      it = notfirst_of_col.emplace(ap.flag_pos, em.unop(VM_NOT_B8, fm, 1)).first;
    }
    *out = it->second;
    return Status::OK();
  };
  // COUNT(*) (or COUNT of a never-NULL column) under the same selection equals the contribution
  // count every non-COUNT aggregate of a never-NULL input already keeps: share it, no instruction
  int count_donor = -1;
  for (size_t j = 0; j < plans.size(); ++j)
    if (plans[j].aggregation != SSGPU_COUNT && !plans[j].distinct && !pipe.cols[plans[j].input_pos].expr->nullable) { count_donor = (int)j; break; }
  for (size_t j = 0; j < plans.size(); ++j) {
    const AggPlan& ap = plans[j];
    AggOut ao; ao.slot = (int)j; ao.has_cnt = true; ao.result_nullable = ap.result_nullable;
    if (ap.aggregation == SSGPU_COUNT) {
      const bool never_null = !ap.distinct && (ap.input_pos < 0 || !pipe.cols[ap.input_pos].expr->nullable);
      if (never_null && count_donor >= 0) {
        ao.slot = count_donor; ao.slot_kind = SLOT_COUNT;
        ao.emit_kind = dtype_width(ap.out_type) == 4 ? EMIT_CNT_U32 : EMIT_CNT_U64;
        st->aggs.push_back(ao);
        Attr a; a.name = ap.out_name; a.dtype = ap.out_type; a.nullable = ap.result_nullable;
        st->out_schema.push_back(a);
        continue;
      }
      int nullreg = -1;
      if (ap.input_pos >= 0) { Val v; SS_RETURN_IF_ERROR(em.value(pipe.cols[ap.input_pos].expr, &v)); nullreg = v.null; }
      if (ap.distinct) { int nf; SS_RETURN_IF_ERROR(notfirst_for(ap, &nf)); nullreg = em.or_null(nullreg, nf); }
      LInstr& i = em.emit(VM_AGG_COUNT); i.dst_is_reg = false; i.dst = (int)j; i.b = nullreg; i.c = sel;
      ao.slot_kind = SLOT_COUNT;
      ao.emit_kind = dtype_width(ap.out_type) == 4 ? EMIT_U32 : EMIT_U64;
    } else {
      const BExprP& src = pipe.cols[ap.input_pos].expr;
      AggSel s;
      if (!select_scalar_agg(ap.aggregation, ap.out_type, &s))
        return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_TYPE, "aggregation not supported for this type");
      // fused SUM(x op y): the binary node feeds only this aggregate -> no LDS round trip
      const MT smt = mtype(src->dtype);
      const bool fusable = !ap.distinct && ap.aggregation == SSGPU_SUM && (int)j < VM_FAST_SLOTS && src->kind == BExpr::OP &&
                           src->dtype == ap.out_type && (smt == M_I64 || smt == M_U64 || smt == M_F64) &&
                           (src->op == OP_ADD || src->op == OP_SUBTRACT || src->op == OP_MULTIPLY) && !em.has_value(src);
      if (fusable) {
        Val x, y;
        SS_RETURN_IF_ERROR(em.value(src->args[0], &x));
        SS_RETURN_IF_ERROR(em.value(src->args[1], &y));
        if (x.imm && y.imm) { x.reg = em.materialize(x); x.imm = false; }
        const int nullreg = em.or_null(x.null, y.null);
        const bool f = smt == M_F64;
        const uint16_t op = src->op == OP_ADD ? (f ? VM_AGG_SUM_F64_ADD : VM_AGG_SUM_I64_ADD)
                          : src->op == OP_SUBTRACT ? (f ? VM_AGG_SUM_F64_SUB : VM_AGG_SUM_I64_SUB)
                                                   : (f ? VM_AGG_SUM_F64_MUL : VM_AGG_SUM_I64_MUL);
        LInstr& i = em.emit(op); i.dst_is_reg = false; i.dst = (int)j; i.b = nullreg; i.c = sel;
        if (x.imm) { i.a_imm = true; i.imm = x.bits; i.imm_width = 8; } else i.a = x.reg;
        if (y.imm) { i.b_imm = true; i.imm = y.bits; i.imm_width = 8; } else i.d = y.reg;
      } else {
        Val v; SS_RETURN_IF_ERROR(em.value(src, &v));
        Val c;
        int wide_type = 0, wide_emit = 0;
        if (minmax_in_own_type(ap.aggregation, src->dtype, ap.out_type, &wide_type, &wide_emit)) {
          select_scalar_agg(ap.aggregation, wide_type, &s); s.emit_kind = wide_emit;
          SS_RETURN_IF_ERROR(em.cast_val(v, mtype(src->dtype), mtype(wide_type), &c));
        } else {
          SS_RETURN_IF_ERROR(em.cast_val(v, mtype(src->dtype), mtype(ap.out_type), &c));
        }
        int vr = em.materialize(c);
        int nullreg = v.null;
        if (ap.distinct) { int nf; SS_RETURN_IF_ERROR(notfirst_for(ap, &nf)); nullreg = em.or_null(v.null, nf); }
        int order_reg = -1;   // FIRST / LAST over re-ordered rows: the column that holds the original order (AggPlan::order_pos)
        if (ap.order_pos >= 0 && (ap.aggregation == SSGPU_FIRST || ap.aggregation == SSGPU_LAST)) {
          Val ov; SS_RETURN_IF_ERROR(em.value(pipe.cols[ap.order_pos].expr, &ov));
          order_reg = em.materialize(ov);
        }
        LInstr& i = em.emit(s.op); i.dst_is_reg = false; i.dst = (int)j; i.a = vr; i.b = nullreg; i.c = sel; i.d = order_reg;
      }
      ao.slot_kind = s.slot_kind; ao.emit_kind = s.emit_kind;
    }
    st->aggs.push_back(ao);
    Attr a; a.name = ap.out_name; a.dtype = ap.out_type; a.nullable = ap.result_nullable;
    st->out_schema.push_back(a);
  }
  st->main.n_slots = (int)plans.size();
  allocate_registers(&st->main);
  st->algorithmic_bytes_per_row = staged_bytes(st->main);
  st->has_filter = !pipe.filters.empty();
  return Status::OK();
}

static Status finish_scalar_agg(const PlanDesc& d, const ssgpu_op& op, const Pipe& pipe, Stage* st) {
  std::vector<AggPlan> plans;
  SS_RETURN_IF_ERROR(bind_aggregations(d, op.agg_first, op.agg_n, schema_of(pipe.cols), &plans));
  return finish_scalar_agg_bound(plans, pipe, st);
}

struct AggPlan;
static thread_local int g_part_rec_align = 0;   // PlanDesc::part_rec_align of the plan being lowered
static Status build_partition_programs(const struct Pipe& pipe, const std::vector<int>& kpos, const std::vector<AggPlan>& plans, Stage* st);

// clustered = AggregateClusters (aggregate_clusters.cc:338-520): the group id of a row is the
// number of key changes before it, computed by a flag + scan pre-pass over the materialised
// input (segment ids arrive as an extra staged UINT32 column) -- no hash table.
struct GroupBinding {   // keys and aggregations of a Group/Clusters aggregate, bound against a pipe's virtual schema
  std::vector<int> kpos; std::vector<std::string> knames;
  std::vector<AggPlan> plans;
};

static Status bind_group_agg(const PlanDesc& d, const ssgpu_op& op, const Pipe& pipe, GroupBinding* g) {
  const Schema vs = schema_of(pipe.cols);
  SS_RETURN_IF_ERROR(bind_projector(d, op.proj_first, op.proj_n, vs, &g->kpos, &g->knames));
  SS_RETURN_IF_ERROR(bind_aggregations(d, op.agg_first, op.agg_n, vs, &g->plans));
  for (auto& ap : g->plans)
    for (auto& kn : g->knames)
      if (kn == ap.out_name)
        return Status::Error(SSGPU_ERROR_ATTRIBUTE_EXISTS, "Duplicate attribute name \"" + kn + "\" in result schema");
  if ((int)g->plans.size() > VM_MAX_AGG_SLOTS || g->kpos.size() > 16)
    return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "too many keys/aggregations for one pipeline");
  return Status::OK();
}

// ---- CONCAT (Stage::ConcatCol) -----------------------------------------------------------------------------------------
// In the aggregate's place the device counts the contributing values (COUNT(x) into UINT64); `pending` remembers which
// result column becomes a STRING built on the host from which stage-input column.
struct ConcatPlan { size_t agg; int input_pos; int dtype; bool distinct; };
static void take_concat_plans(const Schema& vs, std::vector<AggPlan>* plans, std::vector<ConcatPlan>* pending) {
  for (size_t i = 0; i < plans->size(); ++i) {
    AggPlan& ap = (*plans)[i];
    if (ap.aggregation != SSGPU_CONCAT) continue;
    pending->push_back(ConcatPlan{i, ap.input_pos, vs[ap.input_pos].dtype, ap.concat_distinct});
    ap.aggregation = SSGPU_COUNT; ap.out_type = SSGPU_UINT64; ap.result_nullable = false;
  }
}
static bool has_concat(const std::vector<AggPlan>& plans) {
  for (auto& ap : plans) if (ap.aggregation == SSGPU_CONCAT) return true;
  return false;
}

// ---- NaN-exact floating MIN / MAX (PlanDesc::nan_exact) ---------------------------------------------------------------
// The reference's MIN / MAX assign a group's first non-NULL value and then replace it only when `val < result`
// (aggregation_operators.h:189-228): a NaN that comes FIRST stays, later NaNs are skipped.  The kernels skip every NaN,
// so in this form each floating MIN / MAX gets a hidden FIRST of the same column (the FIRST machinery already follows
// input order) and the visible result becomes IF(IS_NAN(first), first, min).  NanFix: visible aggregate -> its hidden FIRST.
struct NanFix { size_t agg; size_t first; };
static void add_nan_exact_plans(bool enabled, const Schema& vs, std::vector<AggPlan>* plans, std::vector<NanFix>* fixes) {
  if (!enabled) return;
  const size_t n_user = plans->size();
  for (size_t i = 0; i < n_user; ++i) {
    const AggPlan ap = (*plans)[i];
    if ((ap.aggregation != SSGPU_MIN && ap.aggregation != SSGPU_MAX) || ap.input_pos < 0) continue;
    if (!dtype_is_float(vs[ap.input_pos].dtype) || !dtype_is_float(ap.out_type)) continue;
    size_t first = 0;
    for (size_t q = n_user; q < plans->size() && !first; ++q)
      if ((*plans)[q].input_pos == ap.input_pos && (*plans)[q].out_type == ap.out_type) first = q;
    if (!first) {
      if ((int)plans->size() + 1 > VM_MAX_AGG_SLOTS) return;   // no slot left: the remaining aggregates keep the order-independent answer
      AggPlan h; h.aggregation = SSGPU_FIRST; h.input_pos = ap.input_pos; h.out_type = ap.out_type;
      h.out_name = "$first$" + std::to_string(plans->size()); h.result_nullable = true;
      plans->push_back(h);
      first = plans->size() - 1;
    }
    fixes->push_back(NanFix{i, first});
  }
}
// the pipe over the aggregate's result: visible columns only, the fixed ones as IF(IS_NAN(first), first, min)
static void apply_nan_fixes(const Schema& out_schema, size_t n_keys, size_t n_user, const std::vector<NanFix>& fixes, Pipe* pipe) {
  std::vector<VCol> cols;
  for (size_t c = 0; c < n_keys + n_user; ++c) {
    VCol v = pipe->cols[c];
    for (auto& f : fixes) {
      if (n_keys + f.agg != c) continue;
      const BExprP mn = pipe->cols[c].expr, fi = pipe->cols[n_keys + f.first].expr;
      BExprP arg = fi;
      if (fi->dtype != SSGPU_DOUBLE) { arg.reset(new BExpr); arg->kind = BExpr::CAST; arg->dtype = SSGPU_DOUBLE; arg->nullable = fi->nullable; arg->name = "CAST_TO_DOUBLE(" + fi->name + ")"; arg->args = {fi}; }
      BExprP isnan(new BExpr); isnan->kind = BExpr::OP; isnan->op = OP_IS_NAN; isnan->dtype = SSGPU_BOOL; isnan->nullable = fi->nullable; isnan->name = "IS_NAN(" + fi->name + ")"; isnan->args = {arg};
      BExprP sel(new BExpr); sel->kind = BExpr::OP; sel->op = OP_IF; sel->dtype = out_schema[c].dtype; sel->nullable = true;
      sel->name = "IF(" + isnan->name + ", " + fi->name + ", " + mn->name + ")"; sel->args = {isnan, fi, mn};
      v.expr = sel;
    }
    cols.push_back(v);
  }
  pipe->cols = cols;
}

// *too_wide is set (with an error status) when the hash aggregate cannot run as one fused pipeline
// -- the packed key needs more than 64 bits, or FIRST/LAST reads a computed expression --
// and lower_plan falls back to materialise + sort + clustered aggregation.
static Status finish_group_agg(const GroupBinding& g, const Pipe& pipe, Stage* st, bool clustered = false, bool* too_wide = nullptr,
                               int distinct_flag_input = -1) {
  st->kind = clustered ? STAGE_CLUSTERS : STAGE_GROUP_AGG;
  st->in_schema = pipe.in_schema;
  const std::vector<int>& kpos = g.kpos; const std::vector<std::string>& knames = g.knames;
  const std::vector<AggPlan>& plans = g.plans;
  st->joins = pipe.joins;
  Emitter em(&st->main, &pipe.joins);
  SS_RETURN_IF_ERROR(emit_filters(em, pipe));
  const int sel = em.sel_by_depth.back();
  int slotreg = -1;
  int notfirst = -1;   // DISTINCT aggregates (see finish_scalar_agg_bound)
  if (distinct_flag_input >= 0) { Val f; f.width = 1; f.reg = em.staged(distinct_flag_input, false, 1); notfirst = em.unop(VM_NOT_B8, f, 1); }
  std::map<int, int> notfirst_of_col;
  auto notfirst_for = [&](const AggPlan& ap, int* out) -> Status {
    *out = notfirst;
    if (ap.flag_pos < 0) return Status::OK();
    auto it = notfirst_of_col.find(ap.flag_pos);
    if (it == notfirst_of_col.end()) {
      Val f; SS_RETURN_IF_ERROR(em.value(pipe.cols[ap.flag_pos].expr, &f));
      Val fm; fm.width = 1; fm.reg = em.materialize(f);
      it = notfirst_of_col.emplace(ap.flag_pos, em.unop(VM_NOT_B8, fm, 1)).first;
    }
    *out = it->second;
    return Status::OK();
  };
  if (clustered) {
    for (size_t k = 0; k < kpos.size(); ++k) {
      const BExprP& ke = pipe.cols[kpos[k]].expr;
      if (ke->kind != BExpr::INPUT) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "clustered keys must be plain input columns");
      if (mtype(ke->dtype) == M_BAD) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "cluster key type is outside the device hot path");
      SortKey sk; sk.col = ke->input_col; sk.order = 0; st->sort_keys.push_back(sk);
      Attr a; a.name = knames[k]; a.dtype = ke->dtype; a.nullable = ke->nullable;
      st->out_schema.push_back(a);
    }
    slotreg = em.staged((int)pipe.in_schema.size(), false, 4);   // segment ids: synthetic last input column
  } else {
  // pack the key columns into one 64-bit word (value bits + one NULL flag bit per nullable key)
  int keyreg = em.new_reg(8);
  { LInstr& i = em.emit(VM_FILL_64); i.dst = keyreg; i.a_imm = true; i.imm = 0; i.imm_width = 8; }
  uint32_t shift = 0;
  for (size_t k = 0; k < kpos.size(); ++k) {
    const BExprP& ke = pipe.cols[kpos[k]].expr;
    const MT m = mtype(ke->dtype);
    if (m == M_BAD) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, std::string("group key type ") + dtype_name(ke->dtype) + " is outside the device hot path (SURVEY 8f.2)");
    Val v; SS_RETURN_IF_ERROR(em.value(ke, &v));
    const uint32_t w = mwidth(m), bits = w * 8;
    GroupKeyField f; f.out_col = (int)k; f.shift = shift; f.bits = bits; f.width = w;
    f.nullbit = v.null >= 0 ? shift + bits : 0xFF;
    const uint32_t used = bits + (v.null >= 0 ? 1 : 0);
    if (shift + used > 64) {
      if (too_wide) *too_wide = true;
      return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "group keys wider than 64 packed bits");
    }
    int vr = em.materialize(v);
    LInstr& i = em.emit(w == 8 ? VM_KEY_APPEND_64 : w == 4 ? VM_KEY_APPEND_32 : VM_KEY_APPEND_8);
    i.dst = keyreg; i.a = vr; i.b = v.null;
    i.imm = (uint64_t)f.shift | ((uint64_t)f.bits << 8) | ((uint64_t)f.nullbit << 16);
    st->group_keys.push_back(f);
    shift += used;
    Attr a; a.name = knames[k]; a.dtype = ke->dtype; a.nullable = ke->nullable;
    st->out_schema.push_back(a);
  }
  slotreg = em.new_reg(4);
  { LInstr& i = em.emit(VM_GRP_INSERT); i.dst = slotreg; i.a = keyreg; i.c = sel; }
  }
  // accumulator words of a group: one per aggregate, two (sum, compensation) for a DOUBLE sum -- every add's
  // rounding error is captured exactly (TwoSum on the value the atomic returns) and accumulated next to it
  auto is_dd = [](const AggPlan& ap) { return ap.aggregation == SSGPU_SUM && mtype(ap.out_type) == M_F64; };
  uint64_t ng = 0;
  for (auto& ap : plans) ng += ap.aggregation == SSGPU_SUM_RESIDUAL ? 0 : is_dd(ap) ? 2 : 1;
  int rowid_reg = -1;
  int word = 0;
  for (size_t jj = 0; jj < plans.size(); ++jj) {
    const AggPlan& ap = plans[jj];
    const uint64_t j = (uint64_t)word;
    AggOut ao; ao.slot = word; ao.slot_kind = 0; ao.result_nullable = ap.result_nullable; ao.has_cnt = false;
    uint64_t init = 0;
    if (ap.aggregation == SSGPU_SUM_RESIDUAL) {
      // no accumulator of its own: the second view of the SUM's (sum, compensation) words
      int src_j = -1;
      for (size_t q = 0; q < jj; ++q)
        if (plans[q].aggregation == SSGPU_SUM && plans[q].input_pos == ap.input_pos && plans[q].out_type == SSGPU_DOUBLE && !plans[q].distinct) src_j = (int)q;
      if (src_j < 0) return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_TYPE, "SUM_RESIDUAL without its SUM");
      ao = st->aggs[src_j]; ao.emit_kind = EMIT_DDRES_F64; ao.result_nullable = ap.result_nullable;
      st->aggs.push_back(ao);
      Attr a; a.name = ap.out_name; a.dtype = ap.out_type; a.nullable = ap.result_nullable;
      st->out_schema.push_back(a);
      continue;
    }
    if (ap.aggregation == AGG_FIRST_SEEN) {
      if (clustered) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "first-seen group order needs the hash aggregate");
      AggSel s; select_group_agg(SSGPU_MIN, SSGPU_UINT64, &s, &init);
      if (rowid_reg < 0) { rowid_reg = em.new_reg(8); LInstr& r = em.emit(VM_ROWID_64); r.dst = rowid_reg; }
      LInstr& i = em.emit(s.op); i.dst_is_reg = false; i.dst = (int)j; i.a = rowid_reg; i.b = -1; i.c = slotreg;
      i.imm = (ng << 32) | j;
      ao.emit_kind = s.emit_kind;
    } else if (ap.aggregation == SSGPU_COUNT) {
      int nullreg = -1;
      if (ap.input_pos >= 0) { Val v; SS_RETURN_IF_ERROR(em.value(pipe.cols[ap.input_pos].expr, &v)); nullreg = v.null; }
      if (ap.distinct) { int nf; SS_RETURN_IF_ERROR(notfirst_for(ap, &nf)); nullreg = em.or_null(nullreg, nf); }
      LInstr& i = em.emit(VM_GAGG_COUNT); i.dst_is_reg = false; i.dst = (int)j; i.b = nullreg; i.c = slotreg;
      i.imm = (ng << 32) | j;
      ao.emit_kind = dtype_width(ap.out_type) == 4 ? EMIT_U32 : EMIT_U64;
    } else {
      const BExprP& src = pipe.cols[ap.input_pos].expr;
      Val v; SS_RETURN_IF_ERROR(em.value(src, &v));
      int agg_type = ap.out_type, wide_emit = -1;
      minmax_in_own_type(ap.aggregation, src->dtype, ap.out_type, &agg_type, &wide_emit);
      Val c = v;
      if (!ap.rowid_only) SS_RETURN_IF_ERROR(em.cast_val(v, mtype(src->dtype), mtype(agg_type), &c));
      AggSel s;
      int vr;
      if (ap.aggregation == SSGPU_FIRST || ap.aggregation == SSGPU_LAST) {
        // FIRST / LAST (aggregation_operators.h:290-320): MIN / MAX of the contributing row ids;
        // the value is fetched from the input column after extraction (runtime: gather_rowid)
        if (!ap.rowid_only && (src->kind != BExpr::INPUT || dtype_width(src->dtype) == 0)) {
          if (too_wide && dtype_width(src->dtype) != 0) *too_wide = true;   // computed input: materialise it first
          return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "FIRST/LAST inside GroupAggregate need a plain input column on device");
        }
        select_group_agg(ap.aggregation == SSGPU_FIRST ? SSGPU_MIN : SSGPU_MAX, SSGPU_UINT64, &s, &init);
        if (rowid_reg < 0) { rowid_reg = em.new_reg(8); LInstr& r = em.emit(VM_ROWID_64); r.dst = rowid_reg; }
        vr = rowid_reg;
        if (ap.order_pos >= 0) {
          // MIN / MAX of (original row id << 32 | position here): the extremum's low half is where the value is fetched
          // (both are below 2^32: the Sort stage in front refuses more rows)
          Val ov; SS_RETURN_IF_ERROR(em.value(pipe.cols[ap.order_pos].expr, &ov));
          const int packed = em.new_reg(8);
          { LInstr& z = em.emit(VM_FILL_64); z.dst = packed; z.a_imm = true; z.imm = 0; z.imm_width = 8; }
          { LInstr& k = em.emit(VM_KEY_APPEND_64); k.dst = packed; k.a = em.materialize(ov); k.b = -1; k.imm = 32ull | (32ull << 8); }
          { LInstr& k = em.emit(VM_KEY_APPEND_64); k.dst = packed; k.a = rowid_reg; k.b = -1; k.imm = 0ull | (32ull << 8); }
          vr = packed;
          ao.gather_low32 = true;
        }
        if (!ap.rowid_only) ao.gather_col = src->input_col;
      } else {
        if (!select_group_agg(ap.aggregation, agg_type, &s, &init))
          return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_TYPE, "aggregation not supported for this type");
        if (wide_emit >= 0) s.emit_kind = wide_emit;
        vr = em.materialize(c);
      }
      int nullreg = v.null;
      if (ap.distinct) { int nf; SS_RETURN_IF_ERROR(notfirst_for(ap, &nf)); nullreg = em.or_null(v.null, nf); }
      ao.has_cnt = nullreg >= 0;
      LInstr& i = em.emit(s.op); i.dst_is_reg = false; i.dst = (int)j; i.a = vr; i.b = nullreg; i.c = slotreg;
      i.imm = ((uint64_t)(ao.has_cnt ? 1 : 0) << 63) | (ng << 32) | j;
      ao.emit_kind = s.emit_kind;
    }
    if (is_dd(ap)) {
      st->group_acc_init.push_back(0x8000000000000000ull);   // -0.0: the identity of IEEE + (an all -0.0 group sums to -0.0)
      st->group_acc_init.push_back(0);
      st->group_merge_op.push_back(VM_MERGE_ADD_F64_HI);
      st->group_merge_op.push_back(VM_MERGE_ADD_F64);
      ao.emit_kind = EMIT_DD_F64;
      word += 2;
    } else {
      st->group_acc_init.push_back(init);
      uint32_t mop = VM_MERGE_ADD_U64;
      if (ap.aggregation == SSGPU_MIN || ap.aggregation == SSGPU_FIRST || ap.aggregation == AGG_FIRST_SEEN) mop = VM_MERGE_MIN_U64;
      else if (ap.aggregation == SSGPU_MAX || ap.aggregation == SSGPU_LAST) mop = VM_MERGE_MAX_U64;
      else if (ap.aggregation == SSGPU_SUM && mtype(ap.out_type) == M_F32) mop = VM_MERGE_ADD_F64;
      st->group_merge_op.push_back(mop);
      word += 1;
    }
    st->aggs.push_back(ao);
    Attr a; a.name = ap.out_name; a.dtype = ap.out_type; a.nullable = ap.result_nullable;
    st->out_schema.push_back(a);
  }
  st->n_gaggs = (int)ng;
  allocate_registers(&st->main);
  st->algorithmic_bytes_per_row = staged_bytes(st->main);
  st->has_filter = !pipe.filters.empty();
  if (!clustered) SS_RETURN_IF_ERROR(build_partition_programs(pipe, kpos, plans, st));
  return Status::OK();
}

// Partitioned execution of a hash GroupAggregate (many groups): one more program over the same pipe.
// It packs the key exactly like the direct program and writes ONE record per selected row -- the packed key
// followed by every distinct aggregate input (after its cast) and NULL flag -- into the row's (hash
// partition, workgroup) segment of the partition buffer.  Record layout: 8-byte fields (key first), then
// 4-byte fields, then 1-byte fields, padded to a multiple of 8 bytes; adjacent 8-byte fields leave as one
// 16-byte store.
static Status build_partition_programs(const Pipe& pipe, const std::vector<int>& kpos, const std::vector<AggPlan>& plans, Stage* st) {
  Emitter em(&st->part_scatter, &pipe.joins);
  SS_RETURN_IF_ERROR(emit_filters(em, pipe));
  const int sel = em.sel_by_depth.back();
  int keyreg = em.new_reg(8);
  { LInstr& i = em.emit(VM_FILL_64); i.dst = keyreg; i.a_imm = true; i.imm = 0; i.imm_width = 8; }
  for (size_t k = 0; k < kpos.size(); ++k) {
    const BExprP& ke = pipe.cols[kpos[k]].expr;
    Val v; SS_RETURN_IF_ERROR(em.value(ke, &v));
    const GroupKeyField& f = st->group_keys[k];
    int vr = em.materialize(v);
    LInstr& i = em.emit(f.width == 8 ? VM_KEY_APPEND_64 : f.width == 4 ? VM_KEY_APPEND_32 : VM_KEY_APPEND_8);
    i.dst = keyreg; i.a = vr; i.b = v.null;
    i.imm = (uint64_t)f.shift | ((uint64_t)f.bits << 8) | ((uint64_t)f.nullbit << 16);
  }
  const int rank = em.new_reg(4);
  { LInstr& i = em.emit(VM_PART_RANK); i.dst = rank; i.a = keyreg; i.c = sel; }
  struct Field { int reg; uint32_t width; int off; };
  std::vector<Field> fields;
  std::map<int, int> field_of_reg;   // value / mask register -> record field
  auto add_field = [&](int reg, uint32_t w) -> int {
    auto it = field_of_reg.find(reg);
    if (it != field_of_reg.end()) return it->second;
    fields.push_back(Field{reg, w, -1});
    field_of_reg[reg] = (int)fields.size() - 1;
    return (int)fields.size() - 1;
  };
  add_field(keyreg, 8);
  st->part_aggs.clear();
  struct Ref { int val_field, null_field; };
  std::vector<Ref> refs;
  int rowid_reg = -1;
  for (size_t j = 0; j < plans.size(); ++j) {
    const AggPlan& ap = plans[j];
    Stage::PartAgg pa; pa.op = VM_GAGG_COUNT; pa.val_off = -1; pa.val_width = 0; pa.null_off = -1; pa.has_cnt = 0; pa.word = st->aggs[j].slot;
    Ref rf{-1, -1};
    if (ap.aggregation == SSGPU_SUM_RESIDUAL) continue;   // a second view of its SUM's accumulator words: nothing to accumulate
    if (ap.aggregation == AGG_FIRST_SEEN) {
      AggSel sl; uint64_t init = 0;
      select_group_agg(SSGPU_MIN, SSGPU_UINT64, &sl, &init);
      if (rowid_reg < 0) { rowid_reg = em.new_reg(8); LInstr& r = em.emit(VM_ROWID_64); r.dst = rowid_reg; }
      pa.op = sl.op;
      rf.val_field = add_field(rowid_reg, 8); pa.val_width = 8;
    } else if (ap.aggregation == SSGPU_COUNT) {
      if (ap.input_pos >= 0) {
        Val v; SS_RETURN_IF_ERROR(em.value(pipe.cols[ap.input_pos].expr, &v));
        if (v.null >= 0) rf.null_field = add_field(v.null, 1);
      }
    } else {
      const BExprP& src = pipe.cols[ap.input_pos].expr;
      Val v; SS_RETURN_IF_ERROR(em.value(src, &v));
      int agg_type = ap.out_type, wide_emit = -1;
      minmax_in_own_type(ap.aggregation, src->dtype, ap.out_type, &agg_type, &wide_emit);   // (the emit kind is the direct path's: same AggOut)
      Val c = v;
      if (!ap.rowid_only) SS_RETURN_IF_ERROR(em.cast_val(v, mtype(src->dtype), mtype(agg_type), &c));
      AggSel sl; uint64_t init = 0;
      if (ap.aggregation == SSGPU_FIRST || ap.aggregation == SSGPU_LAST) {
        select_group_agg(ap.aggregation == SSGPU_FIRST ? SSGPU_MIN : SSGPU_MAX, SSGPU_UINT64, &sl, &init);
        if (rowid_reg < 0) { rowid_reg = em.new_reg(8); LInstr& r = em.emit(VM_ROWID_64); r.dst = rowid_reg; }
        pa.op = sl.op;
        rf.val_field = add_field(rowid_reg, 8); pa.val_width = 8;
      } else {
        if (!select_group_agg(ap.aggregation, agg_type, &sl, &init))
          return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_TYPE, "aggregation not supported for this type");
        pa.op = sl.op;
        rf.val_field = add_field(em.materialize(c), c.width); pa.val_width = (int)c.width;
      }
      if (v.null >= 0) { rf.null_field = add_field(v.null, 1); pa.has_cnt = 1; }
    }
    st->part_aggs.push_back(pa);
    refs.push_back(rf);
  }
  uint32_t off = 0;
  for (uint32_t w : {8u, 4u, 1u})
    for (auto& f : fields) if (f.width == w) { f.off = (int)off; off += w; }
  st->part_rec_bytes = (off + 7u) & ~7u;
  if (g_part_rec_align > 8) st->part_rec_bytes = (st->part_rec_bytes + (uint32_t)g_part_rec_align - 1u) / (uint32_t)g_part_rec_align * (uint32_t)g_part_rec_align;
  if (st->part_rec_bytes > SSGPU_PART_MAX_WORDS * 8u) { st->part_scatter = Program(); st->part_aggs.clear(); return Status::OK(); }   // wider records: direct path only
  for (size_t j = 0; j < refs.size(); ++j) {
    if (refs[j].val_field >= 0) st->part_aggs[j].val_off = fields[refs[j].val_field].off;
    if (refs[j].null_field >= 0) st->part_aggs[j].null_off = fields[refs[j].null_field].off;
  }
  const uint64_t rb = (uint64_t)st->part_rec_bytes << 16;
  std::vector<const Field*> wide;
  for (auto& f : fields) if (f.width == 8) wide.push_back(&f);
  for (size_t q = 0; q < wide.size(); q += 2) {
    if (q + 1 < wide.size()) {
      LInstr& i = em.emit(VM_PART_REC_128); i.dst_is_reg = false; i.dst = 0; i.a = wide[q]->reg; i.d = wide[q + 1]->reg; i.b = rank; i.imm = (uint64_t)wide[q]->off | rb;
    } else {
      LInstr& i = em.emit(VM_PART_REC_64); i.dst_is_reg = false; i.dst = 0; i.a = wide[q]->reg; i.b = rank; i.imm = (uint64_t)wide[q]->off | rb;
    }
  }
  for (auto& f : fields) {
    if (f.width == 8) continue;
    LInstr& i = em.emit(f.width == 4 ? VM_PART_REC_32 : VM_PART_REC_8); i.dst_is_reg = false; i.dst = 0; i.a = f.reg; i.b = rank; i.imm = (uint64_t)f.off | rb;
  }
  { const uint32_t wpr = st->part_rec_bytes / 8u;   // imm = words per record | (floor(2^32 / wpr) + 1) << 32
    LInstr& i = em.emit(VM_PART_FLUSH); i.dst_is_reg = false; i.dst = 0; i.imm = (uint64_t)wpr | ((uint64_t)(uint32_t)(0x100000000ull / wpr + 1ull) << 32); }
  st->part_scatter.n_outputs = 1;
  // ---- the plain form of the same pass (Stage::PlainScatter): sources of the record's fields, key columns, predicates ----
  {
    Stage::PlainScatter pl;
    pl.ok = pipe.joins.empty() && kpos.size() <= 8 && fields.size() <= 24 && pipe.filters.size() <= 4;
    std::map<int, std::pair<int, bool>> staged_of_reg;   // register -> (input column, is NULL mask)
    for (auto& sgd : st->part_scatter.staged) staged_of_reg[sgd.reg] = std::make_pair(sgd.col, sgd.is_null_mask);
    for (size_t k = 0; pl.ok && k < kpos.size(); ++k) {
      const BExprP& ke = pipe.cols[kpos[k]].expr;
      const GroupKeyField& f = st->group_keys[k];
      if (ke->kind != BExpr::INPUT || (uint32_t)dtype_width(ke->dtype) != f.width) { pl.ok = false; break; }
      pl.keys.push_back(Stage::PlainScatter::Key{ke->input_col, f.width, f.shift, f.bits, f.nullbit, f.nullbit != 0xFF, (int)ke->dtype});
    }
    for (size_t q = 1; pl.ok && q < fields.size(); ++q) {   // field 0 is the packed key
      auto it = staged_of_reg.find(fields[q].reg);
      if (it == staged_of_reg.end() || fields[q].off < 0) { pl.ok = false; break; }
      const int col = it->second.first;
      if (!it->second.second && (uint32_t)dtype_width(pipe.in_schema[col].dtype) != fields[q].width) { pl.ok = false; break; }
      pl.fields.push_back(Stage::PlainScatter::Field{col, it->second.second, fields[q].width, (uint32_t)fields[q].off});
    }
    for (size_t q = 0; pl.ok && q < pipe.filters.size(); ++q) {
      const BExprP& e = pipe.filters[q];
      if (e->kind == BExpr::INPUT && mtype(e->dtype) == M_B8) {   // a BOOL column as the predicate itself (e.g. the validity column of unpacked result images): column != FALSE
        pl.preds.push_back(Stage::PlainScatter::Pred{e->input_col, (int)M_B8, 3, true, e->nullable, 0});
        continue;
      }
      if (e->kind != BExpr::OP || e->args.size() != 2 || !(e->op == OP_LESS || e->op == OP_LESS_OR_EQUAL || e->op == OP_EQUAL || e->op == OP_NOT_EQUAL)) { pl.ok = false; break; }
      const bool col_left = e->args[0]->kind == BExpr::INPUT && e->args[1]->kind == BExpr::CONST;
      const bool col_right = e->args[1]->kind == BExpr::INPUT && e->args[0]->kind == BExpr::CONST;
      if (!col_left && !col_right) { pl.ok = false; break; }
      const BExprP& ce = e->args[col_left ? 0 : 1]; const BExprP& ke = e->args[col_left ? 1 : 0];
      const MT m = mtype(ce->dtype);
      if (m != mtype(ke->dtype) || !(m == M_I32 || m == M_U32 || m == M_I64 || m == M_U64 || m == M_F32 || m == M_F64)) { pl.ok = false; break; }
      pl.preds.push_back(Stage::PlainScatter::Pred{ce->input_col, (int)m, e->op == OP_LESS ? 0 : e->op == OP_LESS_OR_EQUAL ? 1 : e->op == OP_EQUAL ? 2 : 3,
                                                   col_left, ce->nullable, ke->bits});
    }
    if (!pl.ok) { pl.keys.clear(); pl.fields.clear(); pl.preds.clear(); }
    st->plain = pl;
  }
  // records beyond 16 words exist in the plain form only (the scatter as its own kernel; the VM's record tile would not fit the LDS)
  if (st->part_rec_bytes > 128u && !st->plain.ok) { st->part_scatter = Program(); st->part_aggs.clear(); return Status::OK(); }
  allocate_registers(&st->part_scatter);
  return Status::OK();
}

// A materialising stage.  With a filter, two forms exist:
//  * two passes (default): a count pass (predicate + SEL_COUNT) and a device scan give every tile its first output
//    row; the store pass ranks the survivors inside the tile (SEL_RANK) and each lane stores its own rows (STOREC);
//  * one pass (ctx option filter_single_pass): SEL_RANK_LB counts the tile's survivors, learns the rows kept by all
//    earlier tiles by decoupled look-back and the columns are gathered from LDS into consecutive output rows (STOREG).
//    The gathers read rows other threads computed, so a batch of column values is evaluated first and a workgroup
//    barrier (SEL_RANK_LB's own, or BARRIER for later batches) separates it from its stores; a second barrier keeps
//    the next batch's temporaries from reusing registers that are still being gathered from.
// Measured on MI355X (100 M rows x 8 columns, 50 %): 2.37 ms for the two passes against 3.2 ms for the single pass --
// the look-back is a chain of n_tiles / grid cross-XCD round trips ("materialising Filter").
static thread_local bool g_filter_single_pass = false;   // PlanDesc::filter_single_pass of the plan being lowered
static Status finish_materialize(const Pipe& pipe, Stage* st) {
  const bool single = g_filter_single_pass;
  st->kind = STAGE_MATERIALIZE;
  st->in_schema = pipe.in_schema;
  st->out_schema = schema_of(pipe.cols);
  st->has_filter = !pipe.filters.empty();
  st->single_pass = st->has_filter && single;
  if (st->has_filter && !single) {
    Emitter ec(&st->count_pass, &pipe.joins);
    SS_RETURN_IF_ERROR(emit_filters(ec, pipe));
    LInstr& i = ec.emit(VM_SEL_COUNT); i.dst_is_reg = false; i.dst = 0; i.a = ec.sel_by_depth.back();
    allocate_registers(&st->count_pass);
  }
  st->joins = pipe.joins;
  Emitter em(&st->main, &pipe.joins);
  SS_RETURN_IF_ERROR(emit_filters(em, pipe));
  const int sel = em.sel_by_depth.back();
  int rank = -1;
  if (st->has_filter && !single) {
    rank = em.new_reg(4);
    LInstr& i = em.emit(VM_SEL_RANK); i.dst = rank; i.a = sel;
  }
  int out_index = 0;
  struct Pending { Val v; bool nullable; };
  std::vector<Pending> batch;
  uint32_t batch_bytes = 0;
  auto store = [&](const Val& x, uint32_t w) {
    const uint16_t op = !st->has_filter ? (w == 8 ? VM_STORE_64 : w == 4 ? VM_STORE_32 : VM_STORE_8)
                        : single        ? (w == 8 ? VM_STOREG_64 : w == 4 ? VM_STOREG_32 : VM_STOREG_8)
                                        : (w == 8 ? VM_STOREC_64 : w == 4 ? VM_STOREC_32 : VM_STOREC_8);
    LInstr& i = em.emit(op); i.dst_is_reg = false; i.dst = out_index++;
    if (x.imm) { i.a_imm = true; i.imm = x.bits; i.imm_width = (uint8_t)x.width; } else i.a = x.reg;
    if (st->has_filter) { i.b = rank; if (!single) i.c = sel; }
  };
  auto flush = [&]() {
    if (st->single_pass) {
      if (rank < 0) { rank = em.new_reg(4); LInstr& i = em.emit(VM_SEL_RANK_LB); i.dst = rank; i.a = sel; }
      else { LInstr& i = em.emit(VM_BARRIER); i.dst_is_reg = false; i.dst = 0; }
    }
    for (auto& pd : batch) {
      store(pd.v, pd.v.width);
      if (pd.nullable) {
        Val nv; nv.width = 1;
        if (pd.v.null >= 0) nv.reg = pd.v.null; else { nv.imm = true; nv.bits = 0; }
        store(nv, 1);
      }
    }
    batch.clear(); batch_bytes = 0;
  };
  const uint32_t kBatchBytes = 128;   // bytes per row of computed values kept live for one batch of gathers
  for (size_t c = 0; c < pipe.cols.size(); ++c) {
    const BExprP& e = pipe.cols[c].expr;
    if (st->single_pass && batch_bytes >= kBatchBytes) {
      flush();
      LInstr& i = em.emit(VM_BARRIER); i.dst_is_reg = false; i.dst = 0;
    }
    const size_t before = st->main.code.size();
    Val v; SS_RETURN_IF_ERROR(em.value(e, &v));
    if (st->main.code.size() != before) batch_bytes += v.width + (e->nullable ? 1u : 0u);
    batch.push_back(Pending{v, e->nullable});
    if (!st->single_pass) flush();
    st->output_bytes_per_row += v.width + (e->nullable ? 1 : 0);
  }
  flush();
  if (st->single_pass) {   // the next tile's staging overwrites the input registers: not before the last gather is done
    LInstr& i = em.emit(VM_BARRIER); i.dst_is_reg = false; i.dst = 0;
  }
  if (out_index > VM_MAX_OUTPUTS) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "too many output columns for one pipeline");
  st->main.n_outputs = out_index;
  allocate_registers(&st->main);
  st->algorithmic_bytes_per_row = staged_bytes(st->main);
  return Status::OK();
}

// The columns an aggregate reads -- keys first, then the aggregated inputs, each once, in first-use order -- as a pipe of their own
// (what the composed shapes materialise before they sort): *kpos and the plans' input_pos are rewritten to positions in it.
static Pipe prune_to_used(const Pipe& pipe, std::vector<int>* kpos, std::vector<AggPlan>* plans) {
  std::vector<int> used;
  auto slot_of = [&](int pos) { for (size_t i = 0; i < used.size(); ++i) if (used[i] == pos) return (int)i; used.push_back(pos); return (int)used.size() - 1; };
  if (kpos) for (auto& k : *kpos) k = slot_of(k);
  for (auto& ap : *plans) if (ap.input_pos >= 0) ap.input_pos = slot_of(ap.input_pos);
  Pipe pruned = pipe; pruned.cols.clear();
  for (size_t i = 0; i < used.size(); ++i) { VCol c = pipe.cols[used[i]]; c.name = "c" + std::to_string(i); pruned.cols.push_back(c); }
  return pruned;
}
// One more column of such a pipe: the row's index in the plan's input ("$row", UINT64, NOT NULL).  -> its position
static int add_row_id_column(Pipe* pruned) {
  VCol rc; rc.name = "$row"; rc.expr = std::make_shared<BExpr>();
  rc.expr->kind = BExpr::ROWID; rc.expr->dtype = SSGPU_UINT64; rc.expr->nullable = false; rc.expr->name = rc.name;
  pruned->cols.push_back(rc);
  return (int)pruned->cols.size() - 1;
}

// The DISTINCT shape behind its first materialise (`pipe` = the identity over that stage's rows).  Every DISTINCT column but the
// first gets its first-of-run flags as a stored BOOL column: the rows are sorted by (sort_keys, that column), flagged (the
// synthetic input behind the stage's columns) and written back with the flag; the last sort -- by (sort_keys, first DISTINCT
// column) -- carries those flags along as payload; then the aggregation over the sorted rows: scalar, or clustered by g->kpos.
// restore_row_pos >= 0 (the pipe column holding the input row id): the specification also holds a sum folded row after row, which
// needs every group's rows in INPUT order -- then EVERY DISTINCT column gets stored flags, a last sort by (sort_keys, row id)
// puts the rows back, and the aggregation reads the flags as columns; the row-after-row sums leave the program (take_sequential).
// CONCAT aggregates ride the same order: they become COUNTs here and *concats says which (the caller attaches Stage::ConcatCol).
static Status lower_distinct_sorts(std::vector<Stage>* stages, Pipe* pipe_io, GroupBinding* g_io, const std::vector<int>& sort_keys,
                                   const std::vector<int>& dcols, bool scalar, Stage* st_out, int restore_row_pos = -1,
                                   std::vector<ConcatPlan>* concats = nullptr) {
  Pipe& pipe = *pipe_io; GroupBinding& g = *g_io; Stage& st = *st_out;
  // sorts the pipe's rows by (sort_keys, dcol); *run_cols = those columns
  auto sort_by_run = [&](int dcol, std::vector<int>* run_cols) -> Status {
    Stage so; so.kind = STAGE_SORT; so.in_schema = pipe.in_schema; so.out_schema = pipe.in_schema;
    *run_cols = sort_keys; run_cols->push_back(dcol);
    for (int k : *run_cols) {
      if (dtype_width(so.in_schema[k].dtype) == 0) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "variable-length keys are outside the device hot path");
      SortKey sk; sk.col = k; sk.order = SSGPU_ASCENDING; so.sort_keys.push_back(sk);
    }
    for (size_t i = 0; i < so.in_schema.size(); ++i) so.sort_out_cols.push_back((int)i);
    stages->push_back(so);
    reset_pipe(&pipe, so.out_schema);
    return Status::OK();
  };
  for (size_t e = restore_row_pos >= 0 ? 0 : 1; e < dcols.size(); ++e) {
    std::vector<int> run_cols;
    SS_RETURN_IF_ERROR(sort_by_run(dcols[e], &run_cols));
    const int n_cols = (int)pipe.in_schema.size();
    Pipe with_flag = pipe;
    VCol fc; fc.name = "f" + std::to_string(e);
    auto fe = std::make_shared<BExpr>();
    fe->kind = BExpr::INPUT; fe->input_col = n_cols; fe->dtype = SSGPU_BOOL; fe->nullable = false; fe->name = fc.name;
    fc.expr = fe;
    with_flag.cols.push_back(fc);
    Stage mf; SS_RETURN_IF_ERROR(finish_materialize(with_flag, &mf));
    mf.distinct_cols = run_cols;
    stages->push_back(mf);
    reset_pipe(&pipe, mf.out_schema);
    for (auto& ap : g.plans) if (ap.distinct && ap.input_pos == dcols[e]) ap.flag_pos = n_cols;
  }
  std::vector<int> run_cols;
  if (restore_row_pos >= 0) {
    SS_RETURN_IF_ERROR(sort_by_run(restore_row_pos, &run_cols));
    if (concats) take_concat_plans(schema_of(pipe.cols), &g.plans, concats);
    std::vector<Stage::SeqSum> seqs;
    SS_RETURN_IF_ERROR(take_sequential(&g.plans, pipe, scalar ? 0 : g.kpos.size(), &seqs));
    if (scalar) SS_RETURN_IF_ERROR(finish_scalar_agg_bound(g.plans, pipe, &st));
    else SS_RETURN_IF_ERROR(finish_group_agg(g, pipe, &st, true));
    st.seq_sums = seqs;
    return Status::OK();
  }
  SS_RETURN_IF_ERROR(sort_by_run(dcols[0], &run_cols));
  const int n_in = (int)pipe.in_schema.size();
  if (scalar) SS_RETURN_IF_ERROR(finish_scalar_agg_bound(g.plans, pipe, &st, n_in));
  else SS_RETURN_IF_ERROR(finish_group_agg(g, pipe, &st, true, nullptr, n_in + 1));   // clustered: segment ids at n_in, the flag behind
  st.distinct_cols = run_cols;
  return Status::OK();
}

Status lower_plan(const PlanDesc& d, std::vector<Stage>* stages, Schema* result_schema, std::string* describe) {
  g_filter_single_pass = d.filter_single_pass;
  g_part_rec_align = d.part_rec_align;
  if (d.ops.empty()) return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_VALUE, "empty plan");
  // chain from the root down to the scan
  std::vector<int> chain;
  for (int i = (int)d.ops.size() - 1; i >= 0;) {
    chain.push_back(i);
    const int c = d.ops[i].child;
    if (d.ops[i].kind == SSGPU_OP_SCAN) break;
    if (c < 0 || c >= i) return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_VALUE, "operation child must precede its parent");
    i = c;
  }
  std::reverse(chain.begin(), chain.end());
  if (d.ops[chain[0]].kind != SSGPU_OP_SCAN) return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_VALUE, "plan must start with a scan");
  for (auto& a : d.input_schema)
    if (dtype_width(a.dtype) == 0)
      ;  // STRING/BINARY columns may exist in the input as long as no expression touches them
  std::ostringstream desc;
  Pipe pipe;
  reset_pipe(&pipe, d.input_schema);
  bool pending = true;  // pipe has operations not yet flushed into a stage
  // A clustered aggregation keyed by a hidden first column ($rank, $segment): the stage, then the projection that drops the column.
  // CONCAT columns belong to the LAST stage (the host prints them when the result is fetched), which is that projection: they point
  // back at the aggregation's stage (Stage::ConcatCol::stage), whose ordered input and segment ids hold the values.
  auto drop_hidden_key = [&](const Stage& agg, size_t n_stage_keys, const std::vector<ConcatPlan>& concats) -> Status {
    stages->push_back(agg);
    const int agg_stage = (int)stages->size() - 1;
    reset_pipe(&pipe, agg.out_schema);
    pipe.cols.erase(pipe.cols.begin());
    pending = true;
    if (concats.empty()) return Status::OK();
    Stage fm; SS_RETURN_IF_ERROR(finish_materialize(pipe, &fm));
    for (auto& cp : concats) {
      Stage::ConcatCol cc; cc.out_col = (int)(n_stage_keys - 1 + cp.agg); cc.src_col = cp.input_pos; cc.src_dtype = cp.dtype; cc.stage = agg_stage; cc.distinct = cp.distinct;
      fm.concat.push_back(cc);
    }
    stages->push_back(fm);
    reset_pipe(&pipe, fm.out_schema);
    pending = false;
    return Status::OK();
  };
  for (size_t ci = 1; ci < chain.size(); ++ci) {
    const ssgpu_op& op = d.ops[chain[ci]];
    const Schema vs = schema_of(pipe.cols);
    switch (op.kind) {
      case SSGPU_OP_COMPUTE: {
        std::vector<BExprP> bound;
        SS_RETURN_IF_ERROR(bind_expression(d, op.expr, vs, pipe.depth(), &bound));
        std::map<const BExpr*, BExprP> memo;
        std::vector<VCol> nc;
        for (auto& b : bound) { VCol c; c.name = b->name; c.expr = substitute(b, pipe.cols, &memo); nc.push_back(c); }
        // substitution keeps the bound node's name/type; INPUT leaves adopt the child's name, restore it
        for (size_t i = 0; i < nc.size(); ++i) if (nc[i].expr->name != nc[i].name) { BExprP r(new BExpr(*nc[i].expr)); r->name = nc[i].name; nc[i].expr = r; }
        pipe.cols = nc; pending = true;
        desc << "Compute -> [" << schema_to_string(schema_of(pipe.cols)) << "]\n";
      } break;
      case SSGPU_OP_PROJECT: {
        std::vector<int> pos; std::vector<std::string> names;
        SS_RETURN_IF_ERROR(bind_projector(d, op.proj_first, op.proj_n, vs, &pos, &names));
        std::vector<VCol> nc;
        for (size_t i = 0; i < pos.size(); ++i) { VCol c = pipe.cols[pos[i]]; c.name = names[i]; nc.push_back(c); }
        pipe.cols = nc; pending = true;
        desc << "Project -> [" << schema_to_string(schema_of(pipe.cols)) << "]\n";
      } break;
      case SSGPU_OP_FILTER: {
        std::vector<BExprP> bound;
        SS_RETURN_IF_ERROR(bind_expression(d, op.expr, vs, pipe.depth(), &bound));
        if (bound.size() != 1)
          return Status::Error(SSGPU_ERROR_ATTRIBUTE_COUNT_MISMATCH, "Predicate has to return exactly one column of type BOOL");
        if (bound[0]->dtype != SSGPU_BOOL)
          return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH, "Predicate has to return exactly one column of type BOOL");
        std::vector<int> pos; std::vector<std::string> names;
        SS_RETURN_IF_ERROR(bind_projector(d, op.proj_first, op.proj_n, vs, &pos, &names));
        std::map<const BExpr*, BExprP> memo;
        pipe.filters.push_back(substitute(bound[0], pipe.cols, &memo));
        std::vector<VCol> nc;
        for (size_t i = 0; i < pos.size(); ++i) { VCol c = pipe.cols[pos[i]]; c.name = names[i]; nc.push_back(c); }
        pipe.cols = nc; pending = true;
        desc << "Filter " << bound[0]->name << " -> [" << schema_to_string(schema_of(pipe.cols)) << "]\n";
      } break;
      case SSGPU_OP_HASH_JOIN: {
        // HashJoinOperation (hash_join.h:37-56): the probe and the rhs gathers join the lhs pipeline
        const int jtype = (int)(op.option0 & 0xFF), uniq = (int)((op.option0 >> 8) & 0xFF);
        if (jtype != SSGPU_JOIN_INNER && jtype != SSGPU_JOIN_LEFT_OUTER)
          return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "only INNER and LEFT_OUTER hash joins are on device");
        const bool multi = uniq != SSGPU_KEYS_UNIQUE;
        if (op.child2 < 0 || op.child2 >= (int)d.ops.size() || d.ops[op.child2].kind != SSGPU_OP_SCAN || d.ops[op.child2].option0 != 1)
          return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "the rhs of a device hash join must be a scan of the auxiliary input (a resident table)");
        if ((int)pipe.joins.size() >= VM_MAX_JOINS)
          return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "too many hash joins in one pipeline");
        const Schema& rs = d.aux_schema;
        std::vector<int> lpos, rpos; std::vector<std::string> lnames, rnames;
        SS_RETURN_IF_ERROR(bind_projector(d, op.proj_first, op.proj_n, vs, &lpos, &lnames));
        SS_RETURN_IF_ERROR(bind_projector(d, op.proj2_first, op.proj2_n, rs, &rpos, &rnames));
        if (lpos.size() != rpos.size() || lpos.empty())
          return Status::Error(SSGPU_ERROR_ATTRIBUTE_COUNT_MISMATCH, "hash join key selectors must pick the same, non-zero number of columns");
        if (lpos.size() > 8) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "hash joins on more than 8 key columns are not on device");
        JoinSpec js; js.type = jtype; js.multi = multi; js.depth = pipe.depth();
        uint32_t fill[2] = {0, 0};      // a key of 65..128 bits takes a second word; a field never straddles the two (first fit)
        for (size_t k = 0; k < lpos.size(); ++k) {
          const BExprP& le = pipe.cols[lpos[k]].expr;
          if (le->dtype != rs[rpos[k]].dtype)
            return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH, std::string("hash join key types differ: ") + dtype_name(le->dtype) + " vs " + dtype_name(rs[rpos[k]].dtype));
          const uint32_t w = (uint32_t)dtype_width(le->dtype);
          if (w == 0) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "hash join key type is outside the device hot path");
          const uint32_t word = fill[0] + w * 8 <= 64 ? 0u : 1u;
          if (fill[word] + w * 8 > 64) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "hash join keys that do not pack into two 64-bit words are not on device yet");
          if (word) js.wide = true;
          GroupKeyField f; f.out_col = (int)k; f.shift = fill[word]; f.bits = w * 8; f.width = w; f.word = word;
          f.nullbit = 0xFF;   // NULL keys never match: lhs NULLs are masked after the probe, rhs NULL rows are not indexed
          fill[word] += f.bits;
          js.lhs_keys.push_back(le); js.rhs_key_cols.push_back(rpos[k]); js.fields.push_back(f);
        }
        const int join_id = (int)pipe.joins.size();
        std::vector<VCol> nc;
        std::vector<Stage::JoinOut> jout;   // multi join: where every result column comes from
        std::vector<VCol> lhs_fields;       // multi join: the lhs columns the result keeps
        Schema joined_schema;
        for (int q = 0; q < op.proj3_n; ++q) {
          const ssgpu_proj& pr = d.projs[op.proj3_first + q];
          std::vector<int> pos; std::vector<std::string> names;
          SS_RETURN_IF_ERROR(bind_projector(d, op.proj3_first + q, 1, pr.source == 0 ? vs : rs, &pos, &names));
          for (size_t i = 0; i < pos.size(); ++i) {
            VCol c;
            if (pr.source == 0) { c = pipe.cols[pos[i]]; }
            else if (pr.source == 1) {
              BExprP e(new BExpr);
              e->kind = BExpr::JOINCOL; e->join_id = join_id; e->input_col = pos[i]; e->dtype = rs[pos[i]].dtype;
              e->nullable = rs[pos[i]].nullable || jtype == SSGPU_JOIN_LEFT_OUTER;   // hash_join.h:39-40
              e->name = names[i]; e->filter_depth = pipe.depth();
              if (dtype_width(e->dtype) == 0) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "rhs column type is outside the device hot path");
              c.expr = e;
            } else return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_VALUE, "hash join result projector: source index must be 0 or 1");
            c.name = names[i];
            for (auto& o : nc) if (o.name == c.name) return Status::Error(SSGPU_ERROR_ATTRIBUTE_EXISTS, "Duplicate attribute name \"" + c.name + "\" in result schema");
            nc.push_back(c);
            Attr a; a.name = c.name; a.dtype = c.expr->dtype; a.nullable = c.expr->nullable;
            joined_schema.push_back(a);
            if (pr.source == 0) { jout.push_back({false, (int)lhs_fields.size()}); VCol f = c; f.name = "l" + std::to_string(lhs_fields.size()); lhs_fields.push_back(f); }
            else jout.push_back({true, pos[i]});
          }
        }
        if (multi) {
          // NOT_UNIQUE rhs keys multiply rows, which no per-row pipeline can do: the lhs pipeline ends here.
          // It materialises the lhs columns the result keeps plus, per row, the run [start, start + count) of
          // matching rhs rows (LEFT_OUTER: a run of one "no row" for an unmatched lhs row); the expand stage
          // turns the runs into (lhs row, rhs row) pairs and gathers every result column.  Output order is
          // the reference's: lhs order, matches in rhs order (hash_join.cc:361-470, row_hash_set.cc:581-600).
          auto join_expr = [&](BExpr::Kind k, int dtype, const char* name) {
            BExprP e(new BExpr);
            e->kind = k; e->join_id = join_id; e->dtype = dtype; e->nullable = false; e->name = name; e->filter_depth = pipe.depth();
            return e;
          };
          BExprP start = join_expr(BExpr::JOINSTART, SSGPU_UINT32, "JOIN_RUN_START");
          BExprP cnt = join_expr(BExpr::JOINCNT, SSGPU_UINT32, "JOIN_RUN_COUNT");
          BExprP match = join_expr(BExpr::JOINMATCH, SSGPU_BOOL, "JOIN_MATCH");
          Pipe pp = pipe;
          pp.joins.push_back(js);
          if (jtype == SSGPU_JOIN_INNER) pp.filters.push_back(match);
          else {
            auto pick = [&](const BExprP& matched, uint32_t otherwise, const char* name) {
              BExprP k(new BExpr); k->kind = BExpr::CONST; k->dtype = SSGPU_UINT32; k->bits = otherwise; k->name = "CONST_UINT32"; k->filter_depth = pipe.depth();
              BExprP e(new BExpr);
              e->kind = BExpr::OP; e->op = OP_IF; e->dtype = SSGPU_UINT32; e->nullable = false; e->name = name; e->filter_depth = pipe.depth();
              e->args = {match, matched, k};
              return e;
            };
            start = pick(start, VM_NONE, "JOIN_RUN_START_OR_NONE");
            cnt = pick(cnt, 1u, "JOIN_RUN_COUNT_OR_ONE");
          }
          pp.cols = lhs_fields;
          VCol cs; cs.expr = start; cs.name = "__join_run_start"; pp.cols.push_back(cs);
          VCol cc; cc.expr = cnt; cc.name = "__join_run_count"; pp.cols.push_back(cc);
          Stage m; SS_RETURN_IF_ERROR(finish_materialize(pp, &m));
          stages->push_back(m);
          Stage x; x.kind = STAGE_JOIN_EXPAND; x.in_schema = m.out_schema; x.out_schema = joined_schema; x.join_out = jout;
          stages->push_back(x);
          reset_pipe(&pipe, x.out_schema);
          pending = false;
          desc << (jtype == SSGPU_JOIN_INNER ? "HashJoin INNER" : "HashJoin LEFT_OUTER") << " (NOT_UNIQUE: materialise + expand) -> [" << schema_to_string(x.out_schema) << "]\n";
          break;
        }
        if (jtype == SSGPU_JOIN_INNER) {   // rows without a match are dropped: one more (never-NULL) filter
          BExprP m(new BExpr);
          m->kind = BExpr::JOINMATCH; m->join_id = join_id; m->dtype = SSGPU_BOOL; m->nullable = false;
          m->name = "JOIN_MATCH"; m->filter_depth = pipe.depth();
          pipe.filters.push_back(m);
        }
        pipe.joins.push_back(js);
        pipe.cols = nc; pending = true;
        desc << (jtype == SSGPU_JOIN_INNER ? "HashJoin INNER" : "HashJoin LEFT_OUTER") << " -> [" << schema_to_string(schema_of(pipe.cols)) << "]\n";
      } break;
      case SSGPU_OP_SCALAR_AGGREGATE: case SSGPU_OP_GROUP_AGGREGATE: case SSGPU_OP_BEST_EFFORT_GROUP_AGGREGATE: {
        // BestEffortGroupAggregate (aggregate.h:230-250, aggregate_groups.cc:332-433): a GroupAggregate whose table holds a bounded
        // number of groups.  Lowered as the hash aggregate with the hidden first-seen row id, the table sorted by it (= first-seen
        // order) and a CUT: the first `capacity` rows are the view, and the first-seen id of row `capacity` -- the first input row
        // whose key found no room -- is where ssgpu_plan_run_best_effort aggregates again from (runtime.cpp).
        const bool best_effort = op.kind == SSGPU_OP_BEST_EFFORT_GROUP_AGGREGATE;
        Stage st;
        std::vector<NanFix> nan_fixes; size_t n_user_aggs = 0, n_group_keys = 0;   // NaN-exact form (PlanDesc::nan_exact)
        // DISTINCT aggregates (SUM / COUNT of the distinct values of a group, column_aggregator.cc:308-376): materialise the
        // keys and the aggregated columns, sort by (keys, distinct column), flag the first row of every (keys, value) run
        // and aggregate with the flag standing in for "not NULL": a scalar aggregate over the sorted rows, or the
        // clustered aggregation over the key runs.  One distinct column per specification.
        bool any_distinct = false, any_concat = false, any_seq = false;
        {
          std::vector<AggPlan> probe;
          SS_RETURN_IF_ERROR(bind_aggregations(d, op.agg_first, op.agg_n, schema_of(pipe.cols), &probe));
          for (auto& ap : probe) any_distinct = any_distinct || ap.distinct;
          any_concat = has_concat(probe);
          // (a sum folded row after row needs a result row's rows in input order: the CONCAT shape and the key limit's result-row
          // shape keep it, the DISTINCT shape sorts by the values)
          any_seq = has_sequential(probe);
        }
        const bool limited_group = op.kind == SSGPU_OP_GROUP_AGGREGATE && op.option0 != 0;
        if (best_effort && ci + 1 != chain.size())
          return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "a BestEffortGroupAggregate below another operation is not available on the device path (its views are handed out one run at a time)");
        if (best_effort && (any_distinct || any_concat || any_seq))
          return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "BestEffortGroupAggregate with DISTINCT / CONCAT aggregates or a floating SUM into an integer is not available on the device path");
        if (any_concat && !limited_group && !any_distinct) {
          // CONCAT (Stage::ConcatCol): the values have to reach the host in input order, group by group -- materialise the keys and
          // the aggregated columns, (stable) sort by the keys, aggregate the key runs with the clustered kernel (CONCAT counted as
          // COUNT(x)); the host prints the strings from the sorted rows and their segment ids when the column is fetched.
          if (any_distinct) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "CONCAT next to a DISTINCT aggregate is not available on the device path");
          if (ci + 1 != chain.size()) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "a CONCAT result cannot feed another operation on the device path (its strings are built on the host)");
          GroupBinding g;
          if (op.kind == SSGPU_OP_GROUP_AGGREGATE) SS_RETURN_IF_ERROR(bind_group_agg(d, op, pipe, &g));
          else SS_RETURN_IF_ERROR(bind_aggregations(d, op.agg_first, op.agg_n, schema_of(pipe.cols), &g.plans));
          Pipe pruned = prune_to_used(pipe, &g.kpos, &g.plans);
          Stage m; SS_RETURN_IF_ERROR(finish_materialize(pruned, &m));
          stages->push_back(m);
          reset_pipe(&pipe, m.out_schema);
          if (op.kind == SSGPU_OP_GROUP_AGGREGATE) {
            Stage so; so.kind = STAGE_SORT; so.in_schema = m.out_schema; so.out_schema = m.out_schema;
            for (int k : g.kpos) {
              if (dtype_width(so.in_schema[k].dtype) == 0) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "variable-length group keys are outside the device hot path");
              SortKey sk; sk.col = k; sk.order = SSGPU_ASCENDING; so.sort_keys.push_back(sk);
            }
            for (size_t i = 0; i < so.in_schema.size(); ++i) so.sort_out_cols.push_back((int)i);
            stages->push_back(so);
            reset_pipe(&pipe, so.out_schema);
          }
          std::vector<ConcatPlan> concats;
          take_concat_plans(schema_of(pipe.cols), &g.plans, &concats);
          std::vector<Stage::SeqSum> seqs;   // (a group's rows are adjacent and in input order here: row-after-row sums fold them as they lie)
          SS_RETURN_IF_ERROR(take_sequential(&g.plans, pipe, g.kpos.size(), &seqs));
          if (op.kind == SSGPU_OP_SCALAR_AGGREGATE) SS_RETURN_IF_ERROR(finish_scalar_agg_bound(g.plans, pipe, &st));
          else SS_RETURN_IF_ERROR(finish_group_agg(g, pipe, &st, true));
          st.seq_sums = seqs;
          for (auto& cp : concats) { Stage::ConcatCol cc; cc.out_col = (int)(g.kpos.size() + cp.agg); cc.src_col = cp.input_pos; cc.src_dtype = cp.dtype; cc.distinct = cp.distinct; st.concat.push_back(cc); }
          desc << "(materialise" << (op.kind == SSGPU_OP_GROUP_AGGREGATE ? " + sort + clustered aggregation" : "") << "; CONCAT printed on the host) ";
        } else if ((any_distinct || any_concat || any_seq) && limited_group) {
          // DISTINCT aggregates under GroupAggregateOptions::max_unique_keys_in_result (aggregate.h:160-205).  The reference keeps one
          // set of seen values per RESULT ROW (column_aggregator.cc:308-376 indexes its sets by the row the RowHashSet answered),
          // and under the limit that row is min(first-seen rank of the key, limit) (row_hash_set.cc:500-511) -- so the rows beyond the
          // limit share ONE set, and no merge of per-key results can give its answer.  The result row is computed for every INPUT
          // row instead and stands in for the key: materialise (keys, aggregated columns, row id) -> stable sort by the keys ->
          // store the rows once more with `$rank` (Stage::has_rank: cluster numbers, every cluster's first row id, the clusters'
          // order by it, clamped) and with every key masked to NULL outside the groups that keep a row of their own -> the
          // DISTINCT shape with `$rank` as its one key; the visible keys are FIRST(masked key) by row id; `$rank` is dropped behind.
          // CONCAT under the limit takes the same road: a result row's string joins the values of ALL its rows in input order
          // (column_aggregator.cc:108-124 over the same result index), so the stored rows are sorted by ($rank, row id) and aggregated
          // as clusters of `$rank`; the host prints from that stage's input (Stage::ConcatCol::stage) behind the projection.
          if (any_concat && ci + 1 != chain.size()) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "a CONCAT result cannot feed another operation on the device path (its strings are built on the host)");
          const int64_t limit = op.option0 < 0 ? 0 : op.option0;     // (ranks are below 2^32 - 1: the stage clamps a larger limit to "never folds")
          GroupBinding g; SS_RETURN_IF_ERROR(bind_group_agg(d, op, pipe, &g));
          const size_t n_keys = g.kpos.size();
          Pipe pruned = prune_to_used(pipe, &g.kpos, &g.plans);
          const int row_pos = add_row_id_column(&pruned);
          Stage m; SS_RETURN_IF_ERROR(finish_materialize(pruned, &m));
          stages->push_back(m);
          Stage so; so.kind = STAGE_SORT; so.in_schema = m.out_schema; so.out_schema = m.out_schema;
          for (int k : g.kpos) {
            if (dtype_width(so.in_schema[k].dtype) == 0) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "variable-length group keys are outside the device hot path");
            SortKey sk; sk.col = k; sk.order = SSGPU_ASCENDING; so.sort_keys.push_back(sk);
          }
          for (size_t i = 0; i < so.in_schema.size(); ++i) so.sort_out_cols.push_back((int)i);
          stages->push_back(so);
          reset_pipe(&pipe, so.out_schema);
          const int n_stored = (int)pipe.cols.size();
          auto synthetic = [&](const char* name, int input_col, int dtype) {
            VCol sc; sc.name = name; sc.expr = std::make_shared<BExpr>();
            sc.expr->kind = BExpr::INPUT; sc.expr->input_col = input_col; sc.expr->dtype = dtype; sc.expr->nullable = false; sc.expr->name = name;
            pipe.cols.push_back(sc);
            return (int)pipe.cols.size() - 1;
          };
          const int rank_pos = synthetic("$rank", n_stored, SSGPU_UINT32);
          const int own_pos = synthetic("$own", n_stored + 1, SSGPU_BOOL);
          std::vector<int> masked_pos;
          for (size_t k = 0; k < n_keys; ++k) {
            const BExprP key = pipe.cols[g.kpos[k]].expr;
            BExprP none(new BExpr); none->kind = BExpr::NULLCONST; none->dtype = key->dtype; none->nullable = true; none->name = "NULL";
            VCol v; v.name = "$key" + std::to_string(k); v.expr = std::make_shared<BExpr>();
            v.expr->kind = BExpr::OP; v.expr->op = OP_IF; v.expr->dtype = key->dtype; v.expr->nullable = true;
            v.expr->name = "IF($own, " + key->name + ", NULL)"; v.expr->args = {pipe.cols[own_pos].expr, key, none};
            pipe.cols.push_back(v);
            masked_pos.push_back((int)pipe.cols.size() - 1);
          }
          Stage mr; SS_RETURN_IF_ERROR(finish_materialize(pipe, &mr));
          mr.has_rank = true; mr.rank_limit = limit; mr.rank_rowid_col = row_pos; mr.segment_cols = g.kpos;
          stages->push_back(mr);
          reset_pipe(&pipe, mr.out_schema);
          GroupBinding gr;
          gr.kpos = {rank_pos}; gr.knames = {"$rank"};
          for (size_t k = 0; k < n_keys; ++k) {
            AggPlan kp; kp.aggregation = SSGPU_FIRST; kp.input_pos = masked_pos[k]; kp.out_type = mr.out_schema[masked_pos[k]].dtype;
            kp.out_name = g.knames[k]; kp.result_nullable = so.out_schema[g.kpos[k]].nullable; kp.order_pos = row_pos;
            gr.plans.push_back(kp);
          }
          std::vector<int> dcols;  // the DISTINCT input columns, in first-use order
          for (auto& ap : g.plans) {
            if (ap.distinct && std::find(dcols.begin(), dcols.end(), ap.input_pos) == dcols.end()) dcols.push_back(ap.input_pos);
            if (ap.aggregation == SSGPU_FIRST || ap.aggregation == SSGPU_LAST) ap.order_pos = row_pos;
            gr.plans.push_back(ap);
          }
          if ((any_concat || any_seq) && !any_distinct) {   // (a result row's rows in input order: CONCAT prints them, a row-after-row SUM folds them)
            Stage s2; s2.kind = STAGE_SORT; s2.in_schema = mr.out_schema; s2.out_schema = mr.out_schema;
            for (int k : {rank_pos, row_pos}) { SortKey sk; sk.col = k; sk.order = SSGPU_ASCENDING; s2.sort_keys.push_back(sk); }
            for (size_t i = 0; i < s2.in_schema.size(); ++i) s2.sort_out_cols.push_back((int)i);
            stages->push_back(s2);
            reset_pipe(&pipe, s2.out_schema);
            std::vector<ConcatPlan> concats;
            take_concat_plans(schema_of(pipe.cols), &gr.plans, &concats);
            std::vector<Stage::SeqSum> seqs;
            SS_RETURN_IF_ERROR(take_sequential(&gr.plans, pipe, 1, &seqs));
            SS_RETURN_IF_ERROR(finish_group_agg(gr, pipe, &st, true));
            st.seq_sums = seqs;
            stages->push_back(st);
            const int cluster_stage = (int)stages->size() - 1;
            reset_pipe(&pipe, st.out_schema);
            pipe.cols.erase(pipe.cols.begin());   // ($rank)
            Stage fm; SS_RETURN_IF_ERROR(finish_materialize(pipe, &fm));
            for (auto& cp : concats) { Stage::ConcatCol cc; cc.out_col = (int)cp.agg; cc.src_col = cp.input_pos; cc.src_dtype = cp.dtype; cc.stage = cluster_stage; cc.distinct = cp.distinct; fm.concat.push_back(cc); }
            desc << "(materialise + sort by the keys + result row of every input row under the limit " << limit
                 << " + sort by (result row, row id) + clustered aggregation" << (concats.empty() ? "" : "; CONCAT printed on the host") << ") GroupAggregate -> [" << schema_to_string(fm.out_schema) << "]\n";
            stages->push_back(fm);
            reset_pipe(&pipe, fm.out_schema);
            pending = false;
            break;
          }
          std::vector<ConcatPlan> concats;
          SS_RETURN_IF_ERROR(lower_distinct_sorts(stages, &pipe, &gr, std::vector<int>{rank_pos}, dcols, false, &st, any_seq || any_concat ? row_pos : -1, &concats));
          desc << "(materialise + sort by the keys + result row of every input row under the limit " << limit << " + " << dcols.size()
               << " x (sort + first-of-run flags)" << (any_seq || any_concat ? " + sort back into input order" : "") << ") GroupAggregate -> ["
               << schema_to_string(st.out_schema) << "] minus its first column\n";
          SS_RETURN_IF_ERROR(drop_hidden_key(st, gr.kpos.size(), concats));
          break;
        } else if (any_distinct) {
          GroupBinding g;
          if (op.kind == SSGPU_OP_GROUP_AGGREGATE) SS_RETURN_IF_ERROR(bind_group_agg(d, op, pipe, &g));
          else SS_RETURN_IF_ERROR(bind_aggregations(d, op.agg_first, op.agg_n, schema_of(pipe.cols), &g.plans));
          // FIRST / LAST follow the INPUT order (aggregation_operators.h:290-320); the DISTINCT shape aggregates rows that were
          // sorted by (keys, distinct column), where "first" would mean "smallest distinct value": refused, not answered wrongly
          // They pick by the row's ORIGINAL id instead, stored as one more column and carried through the sorts (AggPlan::order_pos).
          bool first_last = false;
          for (auto& ap : g.plans) first_last = first_last || ap.aggregation == SSGPU_FIRST || ap.aggregation == SSGPU_LAST;
          Pipe pruned = prune_to_used(pipe, &g.kpos, &g.plans);
          std::vector<int> dcols;  // the DISTINCT input columns, in first-use order
          for (auto& ap : g.plans) if (ap.distinct && std::find(dcols.begin(), dcols.end(), ap.input_pos) == dcols.end()) dcols.push_back(ap.input_pos);
          // CONCAT next to DISTINCT: the values have to reach the host group by group in input order -- the restored order of the
          // row-after-row sums serves it (every DISTINCT column's flags stored, a last sort by (keys, row id))
          if (any_concat && ci + 1 != chain.size()) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "a CONCAT result cannot feed another operation on the device path (its strings are built on the host)");
          const bool restore = any_seq || any_concat;
          int row_pos = -1;
          if (first_last || restore) {
            row_pos = add_row_id_column(&pruned);
            for (auto& ap : g.plans) if (ap.aggregation == SSGPU_FIRST || ap.aggregation == SSGPU_LAST) ap.order_pos = row_pos;
          }
          Stage m; SS_RETURN_IF_ERROR(finish_materialize(pruned, &m));
          stages->push_back(m);
          reset_pipe(&pipe, m.out_schema);
          std::vector<ConcatPlan> concats;
          SS_RETURN_IF_ERROR(lower_distinct_sorts(stages, &pipe, &g, g.kpos, dcols, op.kind == SSGPU_OP_SCALAR_AGGREGATE, &st, restore ? row_pos : -1, &concats));
          for (auto& cp : concats) { Stage::ConcatCol cc; cc.out_col = (int)(g.kpos.size() + cp.agg); cc.src_col = cp.input_pos; cc.src_dtype = cp.dtype; cc.distinct = cp.distinct; st.concat.push_back(cc); }
          desc << "(materialise + " << dcols.size() << " x (sort + first-of-run flags)" << (restore ? " + sort back into input order" : "")
               << (concats.empty() ? "" : "; CONCAT printed on the host") << ") ";
        } else if (op.kind == SSGPU_OP_SCALAR_AGGREGATE) {
          std::vector<AggPlan> plans;
          SS_RETURN_IF_ERROR(bind_aggregations(d, op.agg_first, op.agg_n, schema_of(pipe.cols), &plans));
          n_user_aggs = plans.size();
          std::vector<Stage::SeqSum> seqs;
          if (has_sequential(plans)) {
            // the row-after-row sums read stored columns in input order: the aggregated columns are materialised first
            Pipe pruned = prune_to_used(pipe, nullptr, &plans);
            Stage m; SS_RETURN_IF_ERROR(finish_materialize(pruned, &m));
            stages->push_back(m);
            reset_pipe(&pipe, m.out_schema);
            SS_RETURN_IF_ERROR(take_sequential(&plans, pipe, 0, &seqs));
            desc << "(materialise; floating sums into integers folded row after row) ";
          }
          add_nan_exact_plans(d.nan_exact, schema_of(pipe.cols), &plans, &nan_fixes);
          SS_RETURN_IF_ERROR(finish_scalar_agg_bound(plans, pipe, &st));
          st.seq_sums = seqs;
        } else {
          // GroupAggregateOptions::max_unique_keys_in_result folds every key beyond the limit into one extra last row
          // (aggregate_groups.cc:326): depends on first-seen key order, which no device shape has -- refuse loudly
          GroupBinding g; SS_RETURN_IF_ERROR(bind_group_agg(d, op, pipe, &g));
          // GroupAggregateOptions::max_unique_keys_in_result (aggregate.h:160-205, row_hash_set.cc:500-511): the result keeps the
          // first limit + 1 distinct keys in first-seen order and every row with another key is aggregated into the last of
          // them.  Composed from existing stages: the hash aggregate with one hidden aggregate -- the group's smallest row id
          // --, a sort of the group table by it (= first-seen order), and a fold of the rows beyond the limit into row `limit`
          // with the aggregates' merge functions (SUM of sums, MIN of mins, MAX of maxes, SUM of counts).
          const bool limited = best_effort || op.option0 != 0;     // (both forms need the first-seen row id and the table sorted by it)
          const int64_t limit = op.option0 < 0 ? 0 : op.option0;
          n_user_aggs = g.plans.size(); n_group_keys = g.kpos.size();
          if (!limited || best_effort) add_nan_exact_plans(d.nan_exact, schema_of(pipe.cols), &g.plans, &nan_fixes);   // (under a key limit FIRST has no merge function; a cut merges nothing)
          std::vector<int> fold_ops, fold_by;
          if (best_effort) {
            for (size_t k = 0; k < g.kpos.size() + g.plans.size(); ++k) { fold_ops.push_back(0); fold_by.push_back(-1); }   // every column is kept as it is
            if ((int)g.plans.size() + 1 > VM_MAX_AGG_SLOTS) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "too many aggregations for one pipeline");
            AggPlan hidden; hidden.aggregation = AGG_FIRST_SEEN; hidden.input_pos = -1; hidden.out_type = SSGPU_UINT64;
            hidden.out_name = "$first_seen"; hidden.result_nullable = false;
            g.plans.push_back(hidden);
          } else if (limited) {
            // FIRST / LAST under the limit: the folded row's value is the one at the smallest / largest contributing row id
            // over all the groups it absorbs (NULL inputs never contribute, aggregation_operators.h:290-320) -- every
            // FIRST / LAST gets a hidden twin that yields that row id, and the fold picks the value by it.
            const size_t n_user = g.plans.size();
            std::vector<AggPlan> twins;
            for (size_t k = 0; k < g.kpos.size(); ++k) { fold_ops.push_back(0); fold_by.push_back(-1); }
            for (auto& ap : g.plans) {
              int by = -1;
              if (ap.aggregation == SSGPU_SUM || ap.aggregation == SSGPU_COUNT) fold_ops.push_back(1);
              else if (ap.aggregation == SSGPU_MIN) fold_ops.push_back(2);
              else if (ap.aggregation == SSGPU_MAX) fold_ops.push_back(3);
              else if (ap.aggregation == SSGPU_FIRST || ap.aggregation == SSGPU_LAST) {
                fold_ops.push_back(ap.aggregation == SSGPU_FIRST ? 4 : 5);
                AggPlan t = ap; t.rowid_only = true; t.out_type = SSGPU_UINT64; t.out_name = "$row_of_" + std::to_string(twins.size());
                by = (int)(g.kpos.size() + n_user + twins.size());
                twins.push_back(t);
              }
              else return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "max_unique_keys_in_result with CONCAT aggregates is not available on the device path");
              fold_by.push_back(by);
            }
            for (auto& t : twins) g.plans.push_back(t);
            if ((int)g.plans.size() + 1 > VM_MAX_AGG_SLOTS) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "too many aggregations for one pipeline");
            AggPlan hidden; hidden.aggregation = AGG_FIRST_SEEN; hidden.input_pos = -1; hidden.out_type = SSGPU_UINT64;
            hidden.out_name = "$first_seen"; hidden.result_nullable = false;
            g.plans.push_back(hidden);
          }
          bool too_wide = has_sequential(g.plans);   // (row-after-row sums: the sorted shape keeps the input order inside a group)
          Status s = too_wide ? Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "sorted shape") : finish_group_agg(g, pipe, &st, false, &too_wide);
          if (!s.ok() && !too_wide) return s;
          if (too_wide && limited) {
            // keys wider than one 64-bit word (or FIRST / LAST of a computed value) under a limit: the sorted shape below, its
            // first-seen id = MIN of the input row id stored as one more materialised column; the FIRST / LAST twins order by
            // that column too (AggPlan::order_pos: `row id << 32 | position`, which folds like the plain row id)
            g.plans.pop_back();   // (AGG_FIRST_SEEN: the hash aggregate's form)
          }
          auto append_limit_tail = [&]() -> Status {
            stages->push_back(st);
            Stage so; so.kind = STAGE_SORT; so.in_schema = st.out_schema; so.out_schema = st.out_schema;
            SortKey sk; sk.col = (int)st.out_schema.size() - 1; sk.order = SSGPU_ASCENDING; so.sort_keys.push_back(sk);
            for (size_t i = 0; i < so.in_schema.size(); ++i) so.sort_out_cols.push_back((int)i);
            stages->push_back(so);
            Stage ft; ft.kind = STAGE_FOLD_TAIL; ft.in_schema = so.out_schema; ft.fold_limit = limit; ft.fold_op = fold_ops; ft.fold_by = fold_by;
            ft.fold_cut = best_effort;
            if (best_effort) {
              // GroupAggregateOptions::memory_quota in bytes (option0; 0 = none) -> the rows the result block holds: every visible column's
              // width + its byte of is_null where NULLABLE (block.cc:20-36); aggregate_groups_test.cc:601-626: 20 bytes / (4+1 + 4+1) = 2
              int64_t row_bytes = 0;
              for (size_t i = 0; i < fold_ops.size(); ++i) row_bytes += dtype_width(so.out_schema[i].dtype) + (so.out_schema[i].nullable ? 1 : 0);
              ft.fold_limit = op.option0 > 0 ? std::max<int64_t>(1, op.option0 / std::max<int64_t>(row_bytes, 1)) : 0;   // (0 = no bound; nothing is merged)
            }
            ft.out_schema.assign(so.out_schema.begin(), so.out_schema.begin() + (long)fold_ops.size());   // (the row-id twins and the first-seen id end here)
            for (auto& a : ft.out_schema) if (dtype_width(a.dtype) == 0) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "variable-length columns are outside the device hot path");
            st = ft;
            if (best_effort) desc << "(" << (too_wide ? "" : "hash aggregate + ") << "first-seen order + cut at " << ft.fold_limit << " groups) ";
            else desc << "(" << (too_wide ? "" : "hash aggregate + ") << "first-seen order + fold beyond " << limit << " keys) ";
            return Status::OK();
          };
          if (limited && !too_wide) SS_RETURN_IF_ERROR(append_limit_tail());
          if (too_wide) {
            // Keys that do not pack into one 64-bit word (e.g. a NULLABLE INT64 key, three INT32
            // keys), or FIRST/LAST of a computed expression (the value is fetched by row id from a
            // stored column): materialise the keys and the aggregated columns, radix-sort the rows by the
            // keys and aggregate the now contiguous groups with the clustered kernel.  Group order
            // is unspecified in the reference (hash order, aggregate_groups.cc:332-433).
            GroupBinding gm = g;
            Pipe pruned = prune_to_used(pipe, &gm.kpos, &gm.plans);
            if (limited) {
              const int row_pos = add_row_id_column(&pruned);
              for (auto& ap : gm.plans) if (ap.rowid_only) ap.order_pos = row_pos;
              AggPlan hidden; hidden.aggregation = SSGPU_MIN; hidden.input_pos = row_pos; hidden.out_type = SSGPU_UINT64;
              hidden.out_name = "$first_seen"; hidden.result_nullable = false;
              gm.plans.push_back(hidden);
            }
            Stage m; SS_RETURN_IF_ERROR(finish_materialize(pruned, &m));
            stages->push_back(m);
            Stage so; so.kind = STAGE_SORT; so.in_schema = m.out_schema; so.out_schema = m.out_schema;
            for (int k : gm.kpos) {
              if (dtype_width(so.in_schema[k].dtype) == 0) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "variable-length group keys are outside the device hot path");
              SortKey sk; sk.col = k; sk.order = SSGPU_ASCENDING; so.sort_keys.push_back(sk);
            }
            for (size_t i = 0; i < so.in_schema.size(); ++i) so.sort_out_cols.push_back((int)i);
            stages->push_back(so);
            reset_pipe(&pipe, so.out_schema);
            st = Stage();
            std::vector<Stage::SeqSum> seqs;
            SS_RETURN_IF_ERROR(take_sequential(&gm.plans, pipe, gm.kpos.size(), &seqs));
            SS_RETURN_IF_ERROR(finish_group_agg(gm, pipe, &st, true));
            st.seq_sums = seqs;
            desc << "(materialise + sort + clustered aggregation) ";
            if (limited) SS_RETURN_IF_ERROR(append_limit_tail());
          }
        }
        desc << (op.kind == SSGPU_OP_SCALAR_AGGREGATE ? "ScalarAggregate" : best_effort ? "BestEffortGroupAggregate" : "GroupAggregate") << " -> [" << schema_to_string(st.out_schema) << "]\n";
        stages->push_back(st);
        reset_pipe(&pipe, st.out_schema);
        pending = false;
        if (!nan_fixes.empty()) {   // the visible result is computed from (min, hidden first): one more (tiny) pipeline over the aggregate's rows
          apply_nan_fixes(st.out_schema, n_group_keys, n_user_aggs, nan_fixes, &pipe);
          pending = true;
          desc << "  (NaN-exact MIN / MAX: IF(IS_NAN(first), first, min))\n";
        }
      } break;
      case SSGPU_OP_SORT: case SSGPU_OP_AGGREGATE_CLUSTERS: {
        // blocking operators over materialised rows: flush the pipeline first (a pipe that is
        // still the identity over its input needs no copy)
        bool identity = pipe.filters.empty() && pipe.cols.size() == pipe.in_schema.size();
        for (size_t i = 0; identity && i < pipe.cols.size(); ++i)
          identity = pipe.cols[i].expr->kind == BExpr::INPUT && pipe.cols[i].expr->input_col == (int)i &&
                     pipe.cols[i].name == pipe.in_schema[i].name;
        if (!identity) {
          Stage m; SS_RETURN_IF_ERROR(finish_materialize(pipe, &m));
          stages->push_back(m);
          reset_pipe(&pipe, m.out_schema);
        }
        Stage st;
        std::vector<NanFix> c_fixes; size_t c_user_aggs = 0, c_keys = 0;
        st.in_schema = pipe.in_schema;
        if (op.kind == SSGPU_OP_SORT) {
          st.kind = STAGE_SORT;
          for (int k = 0; k < op.sort_n; ++k) {
            const ssgpu_sortkey& sk = d.sortkeys[op.sort_first + k];
            int pos = lookup_pos(st.in_schema, sk.name);
            if (pos < 0) return Status::Error(SSGPU_ERROR_ATTRIBUTE_MISSING, std::string("No attribute '") + sk.name + "' in the schema:\n '" + schema_to_string(st.in_schema) + "'");
            if (dtype_width(st.in_schema[pos].dtype) == 0) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "variable-length sort keys are outside the device hot path");
            SortKey s; s.col = pos; s.order = sk.order; st.sort_keys.push_back(s);
          }
          std::vector<int> pos; std::vector<std::string> names;
          SS_RETURN_IF_ERROR(bind_projector(d, op.proj_first, op.proj_n, st.in_schema, &pos, &names));
          st.sort_out_cols = pos;
          for (size_t i = 0; i < pos.size(); ++i) { Attr a = st.in_schema[pos[i]]; a.name = names[i]; st.out_schema.push_back(a); }
          desc << "Sort -> [" << schema_to_string(st.out_schema) << "]\n";
        } else {
          GroupBinding g; SS_RETURN_IF_ERROR(bind_group_agg(d, op, pipe, &g));
          c_user_aggs = g.plans.size(); c_keys = g.kpos.size();
          std::vector<ConcatPlan> concats;
          if (has_concat(g.plans)) {
            if (ci + 1 != chain.size()) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "a CONCAT result cannot feed another operation on the device path (its strings are built on the host)");
            for (auto& ap : g.plans)
              if (ap.aggregation == SSGPU_CONCAT && pipe.cols[ap.input_pos].expr->kind != BExpr::INPUT)
                return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "CONCAT inside AggregateClusters needs a plain input column");
            for (auto& ap : g.plans) if (ap.aggregation == SSGPU_CONCAT) ap.input_pos = ap.input_pos;   // (the stage input IS the pipe here: materialised above, or the plan input)
            take_concat_plans(schema_of(pipe.cols), &g.plans, &concats);
          }
          bool any_distinct = false, first_last = false;
          for (auto& ap : g.plans) {
            any_distinct = any_distinct || ap.distinct;
            first_last = first_last || ap.aggregation == SSGPU_FIRST || ap.aggregation == SSGPU_LAST;
          }
          if (any_distinct) {
            // DISTINCT aggregates of clusters (Aggregator::Create serves AggregateClusters too, aggregator.cc:88-101): the DISTINCT
            // shape of the GroupAggregate with the cluster's NUMBER as the sort key.  The rows are stored once more together with
            // their segment id (the boundary scan over the key columns, run in front of this stage: Stage::segment_cols), sorted
            // by (segment id, DISTINCT column), flagged and aggregated as clusters of (segment id, keys...) -- equal keys of
            // different clusters stay apart and the clusters keep their input order; the segment id is projected away behind.
            const bool c_seq = has_sequential(g.plans) || !concats.empty();   // (both need a cluster's rows back in input order)
            std::vector<int> key_inputs;
            for (int k : g.kpos) {
              if (pipe.cols[k].expr->kind != BExpr::INPUT) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "clustered keys must be plain input columns");
              key_inputs.push_back(pipe.cols[k].expr->input_col);
            }
            Pipe pruned = prune_to_used(pipe, &g.kpos, &g.plans);
            for (auto& cp : concats) cp.input_pos = g.plans[cp.agg].input_pos;   // (the CONCAT columns -- COUNTs by now -- moved with the pruning)
            std::vector<int> dcols;  // the DISTINCT input columns, in first-use order
            for (auto& ap : g.plans) if (ap.distinct && std::find(dcols.begin(), dcols.end(), ap.input_pos) == dcols.end()) dcols.push_back(ap.input_pos);
            { VCol sc; sc.name = "$segment"; sc.expr = std::make_shared<BExpr>();
              sc.expr->kind = BExpr::INPUT; sc.expr->input_col = (int)pipe.in_schema.size(); sc.expr->dtype = SSGPU_UINT32; sc.expr->nullable = false; sc.expr->name = sc.name;
              pruned.cols.push_back(sc); }
            const int seg_pos = (int)pruned.cols.size() - 1;
            int row_pos = -1;
            if (first_last || c_seq) {
              row_pos = add_row_id_column(&pruned);
              for (auto& ap : g.plans) if (ap.aggregation == SSGPU_FIRST || ap.aggregation == SSGPU_LAST) ap.order_pos = row_pos;
            }
            Stage m; SS_RETURN_IF_ERROR(finish_materialize(pruned, &m));
            m.segment_cols = key_inputs; m.has_segment = true;
            stages->push_back(m);
            reset_pipe(&pipe, m.out_schema);
            g.kpos.insert(g.kpos.begin(), seg_pos); g.knames.insert(g.knames.begin(), "$segment");
            SS_RETURN_IF_ERROR(lower_distinct_sorts(stages, &pipe, &g, std::vector<int>{seg_pos}, dcols, false, &st, c_seq ? row_pos : -1));
            desc << "(materialise with segment ids + " << dcols.size() << " x (sort + first-of-run flags)) AggregateClusters -> [" << schema_to_string(st.out_schema) << "] minus its first column\n";
            SS_RETURN_IF_ERROR(drop_hidden_key(st, g.kpos.size(), concats));
            break;
          }
          std::vector<Stage::SeqSum> seqs;
          SS_RETURN_IF_ERROR(take_sequential(&g.plans, pipe, g.kpos.size(), &seqs));
          add_nan_exact_plans(d.nan_exact && concats.empty(), schema_of(pipe.cols), &g.plans, &c_fixes);
          SS_RETURN_IF_ERROR(finish_group_agg(g, pipe, &st, true));
          st.seq_sums = seqs;
          for (auto& cp : concats) { Stage::ConcatCol cc; cc.out_col = (int)(g.kpos.size() + cp.agg); cc.src_col = pipe.cols[cp.input_pos].expr->input_col; cc.src_dtype = cp.dtype; cc.distinct = cp.distinct; st.concat.push_back(cc); }
          desc << "AggregateClusters -> [" << schema_to_string(st.out_schema) << "]\n";
        }
        stages->push_back(st);
        reset_pipe(&pipe, st.out_schema);
        pending = false;
        if (!c_fixes.empty()) { apply_nan_fixes(st.out_schema, c_keys, c_user_aggs, c_fixes, &pipe); pending = true; }
      } break;
      default:
        return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "operation kind " + std::to_string(op.kind) + " is outside the device hot path");
    }
  }
  if (pending) {
    Stage m; SS_RETURN_IF_ERROR(finish_materialize(pipe, &m));
    stages->push_back(m);
  }
  for (auto& st : *stages)
    for (const Program* pr : {&st.main, &st.count_pass, &st.part_scatter})
      if (pr->gathers.size() > VM_MAX_JOIN_COLS)
        return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "too many rhs columns gathered by the hash joins of one pipeline");
  *result_schema = stages->back().out_schema;
  for (auto& cc : stages->back().concat) { (*result_schema)[cc.out_col].dtype = SSGPU_STRING; (*result_schema)[cc.out_col].nullable = true; }   // (out_schema keeps the device's UINT64 count)
  for (size_t i = 0; i < stages->size(); ++i) {
    desc << "stage " << i << " kind=" << (*stages)[i].kind << "\n" << disassemble((*stages)[i].main);
    if (!(*stages)[i].count_pass.empty()) desc << " count pass:\n" << disassemble((*stages)[i].count_pass);
    if (!(*stages)[i].part_scatter.empty()) desc << " partition scatter pass:\n" << disassemble((*stages)[i].part_scatter);
    if ((*stages)[i].plain.ok) {
      const Stage::PlainScatter& pl = (*stages)[i].plain;
      desc << " plain partition scatter: keys";
      for (auto& k : pl.keys) desc << " col" << k.col << "(w" << k.width << " <<" << k.shift << (k.nullable ? " nullable" : "") << ")";
      desc << " | fields";
      for (auto& f : pl.fields) desc << " col" << f.col << (f.is_null_mask ? ".null" : "") << "(w" << f.width << ")@" << f.off;
      desc << " | predicates";
      for (auto& q : pl.preds) desc << " " << (q.col_on_left ? "col" : "const") << " cmp" << q.cmp << " " << (q.col_on_left ? "const" : "col") << " [col" << q.col << " kind " << q.kind << " bits " << q.bits << "]";
      desc << "\n";
    }
  }
  *describe = desc.str();
  return Status::OK();
}

}  // namespace ssgpu
