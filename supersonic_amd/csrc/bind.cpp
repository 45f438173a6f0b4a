// bind.cpp -- bind-time semantics of the expression layer, restated from:
//   type promotion      supersonic/expression/templated/bound_expression_factory.cc:44-107
//   casts               supersonic/expression/templated/cast_bound_expression.cc:121-470
//   result naming       supersonic/expression/vector/expression_traits.h:1158-1570
//   nullability         supersonic/expression/infrastructure/expression_utils.cc:130-193
//   comparisons         supersonic/expression/core/comparison_bound_expressions.cc:587-848
//   logic / IF / ISNULL supersonic/expression/core/elementary_bound_expressions.cc:1084-1430
//   constant folding    supersonic/expression/infrastructure/basic_bound_expression.cc:57-92
//   projectors          supersonic/base/infrastructure/projector.cc:95-270
#include "engine.h"

#include <math.h>
#include <string.h>

namespace ssgpu {

// OperatorId values used here (supersonic/expression/proto/operators.proto)
enum {
  OP_ADD = 0, OP_MULTIPLY = 4, OP_SUBTRACT = 8, OP_DIVIDE_QUIET = 13, OP_DIVIDE_NULLING = 14,
  OP_DIVIDE_SIGNALING = 15, OP_CPP_DIVIDE_NULLING = 18, OP_CPP_DIVIDE_SIGNALING = 19,
  OP_MODULUS_NULLING = 26, OP_MODULUS_SIGNALING = 27, OP_NEGATE = 36, OP_AND = 40, OP_OR = 44,
  OP_AND_NOT = 48, OP_NOT = 52, OP_XOR = 56, OP_BITWISE_AND = 60, OP_BITWISE_OR = 64,
  OP_BITWISE_NOT = 68, OP_BITWISE_XOR = 72, OP_SHIFT_LEFT = 76, OP_SHIFT_RIGHT = 80,
  OP_BITWISE_ANDNOT = 84, OP_EQUAL = 100, OP_NOT_EQUAL = 104, OP_LESS = 116, OP_LESS_OR_EQUAL = 120,
  OP_IS_ODD = 140, OP_IS_EVEN = 144, OP_IS_FINITE = 148, OP_IS_INF = 152, OP_IS_NAN = 156, OP_IS_NORMAL = 160,
  OP_ROUND = 300, OP_TRUNC = 304, OP_CEIL_TO_INT = 308, OP_FLOOR_TO_INT = 312, OP_ROUND_TO_INT = 316,
  OP_ROUND_WITH_MULTIPLIER = 364, OP_EXP = 320, OP_LN_QUIET = 325, OP_LN_NULLING = 326, OP_LOG10_QUIET = 329, OP_LOG10_NULLING = 330, OP_POW_QUIET = 353,
  OP_POW_NULLING = 354, OP_POW_SIGNALING = 355, OP_LOG2_QUIET = 357, OP_LOG2_NULLING = 358, OP_SIN = 800, OP_COS = 804, OP_TAN = 808,
  OP_ASIN = 812, OP_ACOS = 816, OP_ATAN = 820, OP_ATAN2 = 824, OP_SINH = 828, OP_COSH = 832, OP_TANH = 836, OP_ASINH = 840,
  OP_ACOSH = 844, OP_ATANH = 848,
  OP_SQRT_QUIET = 333, OP_SQRT_NULLING = 334, OP_SQRT_SIGNALING = 335, OP_CEIL = 342, OP_FLOOR = 346, OP_ABS = 360,
  OP_CASE = 200, OP_IF = 204, OP_IN = 208, OP_IF_NULL = 220, OP_IS_NULL = 224, OP_CAST_QUIET = 265
};

const char* dtype_name(int t) {
  switch (t) {
    case SSGPU_INT32: return "INT32"; case SSGPU_INT64: return "INT64"; case SSGPU_UINT32: return "UINT32";
    case SSGPU_UINT64: return "UINT64"; case SSGPU_FLOAT: return "FLOAT"; case SSGPU_DOUBLE: return "DOUBLE";
    case SSGPU_BOOL: return "BOOL"; case SSGPU_DATE: return "DATE"; case SSGPU_DATETIME: return "DATETIME";
    case SSGPU_STRING: return "STRING"; case SSGPU_BINARY: return "BINARY";
  }
  return "UNKNOWN";
}
int dtype_width(int t) {
  switch (t) {
    case SSGPU_INT32: case SSGPU_UINT32: case SSGPU_FLOAT: case SSGPU_DATE: return 4;
    case SSGPU_INT64: case SSGPU_UINT64: case SSGPU_DOUBLE: case SSGPU_DATETIME: return 8;
    case SSGPU_BOOL: return 1;
    case SSGPU_STRING: return 4;   // dictionary code (see ssgpu.h: STRING columns)
  }
  return 0;
}
bool dtype_is_integer(int t) { return t == SSGPU_INT32 || t == SSGPU_INT64 || t == SSGPU_UINT32 || t == SSGPU_UINT64; }
bool dtype_is_float(int t) { return t == SSGPU_FLOAT || t == SSGPU_DOUBLE; }
bool dtype_is_numeric(int t) { return dtype_is_integer(t) || dtype_is_float(t); }
bool dtype_is_signed_int(int t) { return t == SSGPU_INT32 || t == SSGPU_INT64; }

std::string schema_to_string(const Schema& s) {
  std::string r;
  for (size_t i = 0; i < s.size(); ++i) {
    if (i) r += ", ";
    r += s[i].name + ": " + dtype_name(s[i].dtype) + (s[i].nullable ? "" : " NOT NULL");
  }
  return r;
}

Status copy_plan_desc(const ssgpu_plan_desc* d, PlanDesc* out) {
  if (!d) return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_VALUE, "null plan description");
  // count strings first so that c_str() pointers stay stable
  size_t nstr = (size_t)d->n_attrs + d->n_exprs + 2 * (size_t)d->n_projs + 2 * (size_t)d->n_aggs + d->n_sortkeys;
  out->strings.reserve(nstr + 8);
  auto keep = [&](const char* s) -> const char* {
    out->strings.push_back(s ? s : "");
    return out->strings.back().c_str();
  };
  for (int i = 0; i < d->n_attrs; ++i) {
    Attr a; a.name = d->input_schema[i].name ? d->input_schema[i].name : "";
    a.dtype = d->input_schema[i].dtype; a.nullable = d->input_schema[i].nullable != 0;
    out->input_schema.push_back(a);
  }
  for (int i = 0; i < d->n_aux_attrs && d->aux_schema; ++i) {
    Attr a; a.name = d->aux_schema[i].name ? d->aux_schema[i].name : "";
    a.dtype = d->aux_schema[i].dtype; a.nullable = d->aux_schema[i].nullable != 0;
    out->aux_schema.push_back(a);
  }
  out->ops.assign(d->ops, d->ops + d->n_ops);
  out->exprs.assign(d->exprs, d->exprs + d->n_exprs);
  for (auto& e : out->exprs) e.name = keep(e.name);
  out->expr_args.assign(d->expr_args, d->expr_args + d->n_expr_args);
  out->projs.assign(d->projs, d->projs + d->n_projs);
  for (auto& p : out->projs) { p.name = keep(p.name); p.alias = keep(p.alias); }
  out->aggs.assign(d->aggs, d->aggs + d->n_aggs);
  for (auto& a : out->aggs) { a.input = keep(a.input); a.output = keep(a.output); }
  out->sortkeys.assign(d->sortkeys, d->sortkeys + d->n_sortkeys);
  for (auto& k : out->sortkeys) k.name = keep(k.name);
  return Status::OK();
}

// ---- schema lookups (tuple_schema.h LookupAttributePosition) -------------------
static int lookup_attr(const Schema& s, const std::string& name) {
  for (size_t i = 0; i < s.size(); ++i) if (s[i].name == name) return (int)i;
  return -1;
}

Status bind_projector(const PlanDesc& d, int first, int n, const Schema& schema,
                      std::vector<int>* positions, std::vector<std::string>* names) {
  for (int i = 0; i < n; ++i) {
    const ssgpu_proj& p = d.projs[first + i];
    switch (p.kind) {
      case SSGPU_PROJ_ALL:   // ProjectAllAttributes(prefix): the optional prefix travels in `alias`
        for (size_t c = 0; c < schema.size(); ++c) { positions->push_back((int)c); names->push_back(std::string(p.alias ? p.alias : "") + schema[c].name); }
        break;
      case SSGPU_PROJ_NAMED:
      case SSGPU_PROJ_NAMED_AS: {
        int pos = lookup_attr(schema, p.name);
        if (pos < 0)
          return Status::Error(SSGPU_ERROR_ATTRIBUTE_MISSING,
                               std::string("No attribute '") + p.name + "' in the schema:\n '" + schema_to_string(schema) + "'");
        positions->push_back(pos);
        names->push_back(p.kind == SSGPU_PROJ_NAMED_AS ? std::string(p.alias) : schema[pos].name);
      } break;
      case SSGPU_PROJ_AT:
        if (p.position < 0 || p.position >= (int)schema.size())
          return Status::Error(SSGPU_ERROR_ATTRIBUTE_COUNT_MISMATCH,
                               "source schema has too few attributes (" + std::to_string(schema.size()) + " vs " +
                                   std::to_string(p.position) + ")");
        positions->push_back(p.position);
        names->push_back(schema[p.position].name);
        break;
      default:
        return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_VALUE, "unknown projector kind");
    }
  }
  // compound projector: duplicate result names are an error (projector.cc:258-267)
  for (size_t i = 0; i < names->size(); ++i)
    for (size_t j = i + 1; j < names->size(); ++j)
      if ((*names)[i] == (*names)[j])
        return Status::Error(SSGPU_ERROR_ATTRIBUTE_EXISTS,
                             "Duplicate attribute name \"" + (*names)[i] + "\" in result schema");
  return Status::OK();
}

// ---- constants -------------------------------------------------------------------
static uint64_t const_bits(int dtype, int64_t i64, double f64) {
  uint64_t b = 0;
  switch (dtype) {
    case SSGPU_INT32: case SSGPU_DATE: case SSGPU_STRING: { int32_t v = (int32_t)i64; uint32_t u; memcpy(&u, &v, 4); b = u; } break;
    case SSGPU_UINT32: b = (uint32_t)i64; break;
    case SSGPU_INT64: case SSGPU_DATETIME: case SSGPU_UINT64: memcpy(&b, &i64, 8); break;
    case SSGPU_FLOAT: { float v = (float)f64; uint32_t u; memcpy(&u, &v, 4); b = u; } break;
    case SSGPU_DOUBLE: memcpy(&b, &f64, 8); break;
    case SSGPU_BOOL: b = i64 != 0; break;
  }
  return b;
}

static BExprP make_const(int dtype, uint64_t bits) {
  BExprP e(new BExpr);
  e->kind = BExpr::CONST; e->dtype = dtype; e->nullable = false; e->bits = bits;
  e->name = std::string("CONST_") + dtype_name(dtype);
  return e;
}
static BExprP make_null(int dtype) {
  BExprP e(new BExpr);
  e->kind = BExpr::NULLCONST; e->dtype = dtype; e->nullable = true; e->name = "NULL";
  return e;
}
static bool is_constant(const BExprP& e) { return e->kind == BExpr::CONST || e->kind == BExpr::NULLCONST; }

// typed scalar views of constant bits (bind-time constant folding only)
static double bits_to_double(int dtype, uint64_t b) {
  switch (dtype) {
    case SSGPU_INT32: case SSGPU_DATE: case SSGPU_STRING: return (double)(int32_t)(uint32_t)b;
    case SSGPU_UINT32: return (double)(uint32_t)b;
    case SSGPU_INT64: case SSGPU_DATETIME: return (double)(int64_t)b;
    case SSGPU_UINT64: return (double)b;
    case SSGPU_FLOAT: { float f; uint32_t u = (uint32_t)b; memcpy(&f, &u, 4); return f; }
    case SSGPU_DOUBLE: { double d; memcpy(&d, &b, 8); return d; }
    case SSGPU_BOOL: return b ? 1.0 : 0.0;
  }
  return 0;
}
static int64_t bits_to_i64(int dtype, uint64_t b) {
  switch (dtype) {
    case SSGPU_INT32: case SSGPU_DATE: case SSGPU_STRING: return (int32_t)(uint32_t)b;
    case SSGPU_UINT32: return (uint32_t)b;
    case SSGPU_BOOL: return b != 0;
    default: return (int64_t)b;
  }
}

// cast of a constant (same conversions the device CAST_* instructions perform)
static uint64_t cast_const_bits(int from, int to, uint64_t b) {
  if (dtype_is_float(to)) {
    double v = (from == SSGPU_UINT64) ? (double)b : (dtype_is_float(from) ? bits_to_double(from, b) : (double)bits_to_i64(from, b));
    if (from == SSGPU_UINT64 && to == SSGPU_FLOAT) { float f = (float)b; uint32_t u; memcpy(&u, &f, 4); return u; }
    if (from == SSGPU_INT64 && to == SSGPU_FLOAT) { float f = (float)(int64_t)b; uint32_t u; memcpy(&u, &f, 4); return u; }
    return const_bits(to, 0, v);
  }
  if (dtype_is_float(from)) {
    double v = bits_to_double(from, b);
    if (to == SSGPU_UINT64) { uint64_t u = (uint64_t)v; return u; }
    return const_bits(to, (int64_t)v, 0);
  }
  return const_bits(to, bits_to_i64(from, b), 0);
}

// ---- casts -----------------------------------------------------------------------
static Status make_cast(BExprP child, int to, bool is_implicit, BExprP* out) {
  const int from = child->dtype;
  if (from == to) { *out = child; return Status::OK(); }
  auto bad = [&](const char* why) {
    return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH,
                         std::string("Cannot cast ") + dtype_name(from) + " to " + dtype_name(to) + " in " + child->name + ". " + why);
  };
  if (from == SSGPU_DATE && to == SSGPU_DATETIME) {
    // OPERATOR_DATE_TO_DATETIME (cast_bound_expression.cc:129-136, expression_traits.h:417-425):
    // days since the epoch times the microseconds of a day
    const int64_t micros_per_day = 86400000000LL;
    if (child->kind == BExpr::NULLCONST) { *out = make_null(to); return Status::OK(); }
    if (child->kind == BExpr::CONST) { *out = make_const(to, const_bits(to, bits_to_i64(from, child->bits) * micros_per_day, 0)); return Status::OK(); }
    BExprP wide(new BExpr);
    wide->kind = BExpr::CAST; wide->op = OP_CAST_QUIET; wide->dtype = to; wide->nullable = child->nullable;
    wide->filter_depth = child->filter_depth; wide->name = child->name; wide->args.push_back(child);
    BExprP e(new BExpr);
    e->kind = BExpr::OP; e->op = OP_MULTIPLY; e->dtype = to; e->nullable = child->nullable;
    e->filter_depth = child->filter_depth;
    e->name = "CAST_DATE_TO_DATETIME(" + child->name + ")";
    e->args.push_back(wide); e->args.push_back(make_const(to, const_bits(to, micros_per_day, 0)));
    *out = e;
    return Status::OK();
  }
  if (!dtype_is_numeric(from) || !dtype_is_numeric(to)) return bad("Only numeric casts are supported.");
  if (dtype_is_float(from) && dtype_is_integer(to)) return bad("Casts from floating point to integer types are not allowed.");
  bool down = ((from == SSGPU_INT64 || from == SSGPU_UINT64) && (to == SSGPU_INT32 || to == SSGPU_UINT32 || to == SSGPU_FLOAT)) ||
              (from == SSGPU_DOUBLE && to == SSGPU_FLOAT);
  // UINT64 -> FLOAT is a downcast in the reference too (cast_bound_expression.cc:401)
  if (down && is_implicit) return bad("Implicit downcasts are disallowed in Supersonic, to obtain a downcast use an explicit cast.");
  if (child->kind == BExpr::NULLCONST) { *out = make_null(to); return Status::OK(); }
  if (child->kind == BExpr::CONST) { *out = make_const(to, cast_const_bits(from, to, child->bits)); return Status::OK(); }
  BExprP e(new BExpr);
  e->kind = BExpr::CAST; e->op = OP_CAST_QUIET; e->dtype = to; e->nullable = child->nullable;
  e->filter_depth = child->filter_depth;
  e->name = std::string("CAST_") + dtype_name(from) + "_TO_" + dtype_name(to) + "(" + child->name + ")";
  e->args.push_back(child);
  *out = e;
  return Status::OK();
}

// CommonTypeCalculator (bound_expression_factory.cc:67-107)
static Status common_type(int t1, int t2, int* out) {
  if (t1 == t2) { *out = t1; return Status::OK(); }
  struct Row { int a, b, r; };
  static const Row table[] = {
      {SSGPU_DOUBLE, SSGPU_INT32, SSGPU_DOUBLE}, {SSGPU_DOUBLE, SSGPU_INT64, SSGPU_DOUBLE},
      {SSGPU_DOUBLE, SSGPU_UINT32, SSGPU_DOUBLE}, {SSGPU_DOUBLE, SSGPU_UINT64, SSGPU_DOUBLE},
      {SSGPU_DOUBLE, SSGPU_FLOAT, SSGPU_DOUBLE}, {SSGPU_FLOAT, SSGPU_INT32, SSGPU_FLOAT},
      {SSGPU_FLOAT, SSGPU_UINT32, SSGPU_FLOAT}, {SSGPU_FLOAT, SSGPU_UINT64, SSGPU_DOUBLE},
      {SSGPU_FLOAT, SSGPU_INT64, SSGPU_DOUBLE}, {SSGPU_INT64, SSGPU_INT32, SSGPU_INT64},
      {SSGPU_INT64, SSGPU_UINT32, SSGPU_INT64}, {SSGPU_INT64, SSGPU_UINT64, SSGPU_INT64},
      {SSGPU_UINT64, SSGPU_INT32, SSGPU_INT64}, {SSGPU_UINT64, SSGPU_UINT32, SSGPU_UINT64},
      {SSGPU_UINT32, SSGPU_INT32, SSGPU_INT64}, {SSGPU_DATE, SSGPU_DATETIME, SSGPU_DATETIME}};
  for (const Row& r : table)
    if ((r.a == t1 && r.b == t2) || (r.a == t2 && r.b == t1)) { *out = r.r; return Status::OK(); }
  return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH,
                       std::string("Cannot reconcile types: ") + dtype_name(t1) + " and " + dtype_name(t2) + ".");
}

// ---- bind-time constant folding (host scalars; never row data) ---------------------
template <typename T> static T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }
template <typename T> static uint64_t to_bits(T v) { uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }

static bool fold_binary(int op, int t, uint64_t a, uint64_t b, int* out_type, uint64_t* out, bool* out_null) {
  *out_null = false;
  auto cmp = [&](int c) {  // c: 0 lt, 1 le, 2 eq, 3 ne
    bool r = false;
    if (t == SSGPU_DOUBLE) { double x = from_bits<double>(a), y = from_bits<double>(b); r = c == 0 ? x < y : c == 1 ? x <= y : c == 2 ? x == y : x != y; }
    else if (t == SSGPU_FLOAT) { float x = from_bits<float>(a), y = from_bits<float>(b); r = c == 0 ? x < y : c == 1 ? x <= y : c == 2 ? x == y : x != y; }
    else if (t == SSGPU_UINT64 || t == SSGPU_UINT32 || t == SSGPU_BOOL) { uint64_t x = a, y = b; r = c == 0 ? x < y : c == 1 ? x <= y : c == 2 ? x == y : x != y; }
    else { int64_t x = bits_to_i64(t, a), y = bits_to_i64(t, b); r = c == 0 ? x < y : c == 1 ? x <= y : c == 2 ? x == y : x != y; }
    *out_type = SSGPU_BOOL; *out = r; return true;
  };
  switch (op) {
    case OP_LESS: return cmp(0);
    case OP_LESS_OR_EQUAL: return cmp(1);
    case OP_EQUAL: return cmp(2);
    case OP_NOT_EQUAL: return cmp(3);
    case OP_AND: *out_type = SSGPU_BOOL; *out = (a && b); return true;
    case OP_OR: *out_type = SSGPU_BOOL; *out = (a || b); return true;
    case OP_XOR: *out_type = SSGPU_BOOL; *out = ((a != 0) != (b != 0)); return true;
    case OP_AND_NOT: *out_type = SSGPU_BOOL; *out = (!a && b); return true;
  }
  *out_type = t;
  if (t == SSGPU_DOUBLE) {
    double x = from_bits<double>(a), y = from_bits<double>(b), r;
    switch (op) {
      case OP_ADD: r = x + y; break; case OP_SUBTRACT: r = x - y; break; case OP_MULTIPLY: r = x * y; break;
      case OP_DIVIDE_QUIET: case OP_DIVIDE_SIGNALING: r = x / y; break;
      case OP_DIVIDE_NULLING: if (y == 0) { *out_null = true; *out = 0; return true; } r = x / y; break;
      case OP_ATAN2: r = atan2(x, y); break;
      case OP_ROUND_WITH_MULTIPLIER: r = round(x * y) / y; break;   // operators::RoundWithMultiplier, math_evaluators.h:117-121
      case OP_POW_QUIET: r = pow(x, y); break;
      case OP_POW_NULLING: if (x < 0 && y != trunc(y)) { *out_null = true; *out = 0; return true; } r = pow(x, y); break;
      default: return false;
    }
    *out = to_bits(r); return true;
  }
  if (t == SSGPU_FLOAT) {
    float x = from_bits<float>(a), y = from_bits<float>(b), r;
    switch (op) { case OP_ADD: r = x + y; break; case OP_SUBTRACT: r = x - y; break; case OP_MULTIPLY: r = x * y; break; default: return false; }
    *out = to_bits(r); return true;
  }
  if (dtype_is_integer(t)) {
    const bool w32 = dtype_width(t) == 4;
    uint64_t x = w32 ? (uint32_t)a : a, y = w32 ? (uint32_t)b : b, r;
    const bool is_signed = t == SSGPU_INT32 || t == SSGPU_INT64;
    const int64_t sx = w32 ? (int64_t)(int32_t)(uint32_t)a : (int64_t)a, sy = w32 ? (int64_t)(int32_t)(uint32_t)b : (int64_t)b;
    switch (op) {
      case OP_ADD: r = x + y; break; case OP_SUBTRACT: r = x - y; break; case OP_MULTIPLY: r = x * y; break;
      case OP_BITWISE_AND: r = x & y; break; case OP_BITWISE_OR: r = x | y; break; case OP_BITWISE_XOR: r = x ^ y; break;
      case OP_BITWISE_ANDNOT: r = (~x) & y; break;
      case OP_SHIFT_LEFT: r = x << (y & (w32 ? 31 : 63)); break;
      case OP_SHIFT_RIGHT: r = is_signed ? (uint64_t)(sx >> (y & (w32 ? 31 : 63))) : x >> (y & (w32 ? 31 : 63)); break;
      case OP_CPP_DIVIDE_NULLING: case OP_CPP_DIVIDE_SIGNALING: case OP_MODULUS_NULLING: case OP_MODULUS_SIGNALING: {
        const bool div = op == OP_CPP_DIVIDE_NULLING || op == OP_CPP_DIVIDE_SIGNALING;
        if (y == 0) {
          if (op == OP_CPP_DIVIDE_SIGNALING || op == OP_MODULUS_SIGNALING) return false;   // fails when evaluated, not when bound
          *out_null = true; *out = 0; return true;
        }
        if (is_signed) r = sy == -1 ? (div ? 0ull - (uint64_t)sx : 0ull) : (uint64_t)(div ? sx / sy : sx % sy);
        else r = div ? x / y : x % y;
      } break;
      default: return false;
    }
    *out = w32 ? (uint32_t)r : r; return true;
  }
  return false;
}

// ---- operators ---------------------------------------------------------------------
static std::string fmt_binary(int op, const std::string& l, const std::string& r) {
  switch (op) {
    case OP_ADD: return "(" + l + " + " + r + ")";
    case OP_SUBTRACT: return "(" + l + " - " + r + ")";
    case OP_MULTIPLY: return "(" + l + " * " + r + ")";
    case OP_DIVIDE_QUIET: case OP_DIVIDE_NULLING: case OP_DIVIDE_SIGNALING: return "(" + l + " /. " + r + ")";
    case OP_CPP_DIVIDE_NULLING: case OP_CPP_DIVIDE_SIGNALING: return "(" + l + " / " + r + ")";
    case OP_MODULUS_NULLING: case OP_MODULUS_SIGNALING: return "(" + l + " % " + r + ")";
    case OP_EQUAL: return "(" + l + " == " + r + ")";
    case OP_NOT_EQUAL: return "(" + l + " <> " + r + ")";
    case OP_LESS: return "(" + l + " < " + r + ")";
    case OP_LESS_OR_EQUAL: return "(" + l + " <= " + r + ")";
    case OP_AND: return "(" + l + " AND " + r + ")";
    case OP_OR: return "(" + l + " OR " + r + ")";
    case OP_AND_NOT: return "(" + l + " !&& " + r + ")";
    case OP_XOR: return "(" + l + " XOR " + r + ")";
    case OP_BITWISE_AND: return "(" + l + " & " + r + ")";
    case OP_BITWISE_OR: return "(" + l + " | " + r + ")";
    case OP_BITWISE_XOR: return "(" + l + " ^ " + r + ")";
    case OP_BITWISE_ANDNOT: return "(~" + l + " & " + r + ")";   // expression_traits.h:1527-1536
    case OP_SHIFT_LEFT: return "(" + l + " << " + r + ")";
    case OP_SHIFT_RIGHT: return "(" + l + " >> " + r + ")";
    case OP_IF_NULL: return "IFNULL(" + l + ", " + r + ")";
  }
  return "?";
}

static BExprP make_op(int op, int dtype, bool nullable, const std::string& name, std::vector<BExprP> args, int depth) {
  BExprP e(new BExpr);
  e->kind = BExpr::OP; e->op = op; e->dtype = dtype; e->nullable = nullable; e->name = name;
  e->filter_depth = depth; e->args = args;
  return e;
}

// result of an operator with all-constant children (InitBasicExpression folding)
static bool try_fold(const BExprP& e, BExprP* out) {
  if (e->kind != BExpr::OP && e->kind != BExpr::CAST) return false;
  for (auto& a : e->args) if (!is_constant(a)) return false;
  bool any_null = false;
  for (auto& a : e->args) any_null = any_null || a->kind == BExpr::NULLCONST;
  if (e->op == OP_IS_NULL) { *out = make_const(SSGPU_BOOL, e->args[0]->kind == BExpr::NULLCONST); return true; }
  if (e->args.size() == 2 && (e->op == OP_AND || e->op == OP_OR) && any_null) {
    // three-valued logic with a NULL constant
    const BExprP& x = e->args[0]; const BExprP& y = e->args[1];
    bool xn = x->kind == BExpr::NULLCONST, yn = y->kind == BExpr::NULLCONST;
    bool decided = e->op == OP_AND ? ((!xn && !x->bits) || (!yn && !y->bits)) : ((!xn && x->bits) || (!yn && y->bits));
    if (decided) *out = make_const(SSGPU_BOOL, e->op == OP_AND ? 0 : 1); else *out = make_null(SSGPU_BOOL);
    return true;
  }
  if (any_null) { *out = make_null(e->dtype); return true; }
  if (e->args.size() == 2) {
    int ot; uint64_t ob; bool on;
    if (!fold_binary(e->op, e->args[0]->dtype, e->args[0]->bits, e->args[1]->bits, &ot, &ob, &on)) return false;
    *out = on ? make_null(e->dtype) : make_const(e->dtype, ob);
    return true;
  }
  if (e->args.size() == 1) {
    const BExprP& x = e->args[0];
    if (e->op == OP_NOT) { *out = make_const(SSGPU_BOOL, !x->bits); return true; }
    if (e->op == OP_BITWISE_NOT) { *out = make_const(e->dtype, dtype_width(x->dtype) == 4 ? (uint64_t)(uint32_t)~(uint32_t)x->bits : ~x->bits); return true; }
    {  // exact math family on a constant (same libm calls as math_evaluators.h:82-146,206-220)
      const bool f32 = x->dtype == SSGPU_FLOAT;
      const double d = x->dtype == SSGPU_DOUBLE ? from_bits<double>(x->bits) : f32 ? (double)from_bits<float>(x->bits) : 0.0;
      auto flt = [&](double r) { return f32 ? to_bits((float)r) : to_bits(r); };
      switch (e->op) {
        case OP_ROUND: *out = make_const(e->dtype, f32 ? to_bits(roundf((float)d)) : to_bits(round(d))); return true;
        case OP_CEIL: *out = make_const(e->dtype, flt(ceil(d))); return true;
        case OP_FLOOR: *out = make_const(e->dtype, flt(floor(d))); return true;
        case OP_TRUNC: *out = make_const(e->dtype, flt(trunc(d))); return true;
        case OP_CEIL_TO_INT: *out = make_const(e->dtype, (uint64_t)(int64_t)ceil(d)); return true;
        case OP_FLOOR_TO_INT: *out = make_const(e->dtype, (uint64_t)(int64_t)floor(d)); return true;
        case OP_IS_FINITE: *out = make_const(SSGPU_BOOL, std::isfinite(d)); return true;
        case OP_IS_NAN: *out = make_const(SSGPU_BOOL, std::isnan(d)); return true;
        case OP_IS_INF: *out = make_const(SSGPU_BOOL, std::isinf(d)); return true;
        case OP_IS_NORMAL: *out = make_const(SSGPU_BOOL, std::isnormal(d)); return true;
        case OP_EXP: *out = make_const(e->dtype, to_bits(exp(d))); return true;
        case OP_LN_QUIET: *out = make_const(e->dtype, to_bits(log(d))); return true;
        case OP_LN_NULLING: *out = d <= 0 ? make_null(e->dtype) : make_const(e->dtype, to_bits(log(d))); return true;
        case OP_LOG10_QUIET: *out = make_const(e->dtype, to_bits(log10(d))); return true;
        case OP_LOG10_NULLING: *out = d <= 0 ? make_null(e->dtype) : make_const(e->dtype, to_bits(log10(d))); return true;
        case OP_LOG2_QUIET: *out = make_const(e->dtype, to_bits(log2(d))); return true;
        case OP_LOG2_NULLING: *out = d <= 0 ? make_null(e->dtype) : make_const(e->dtype, to_bits(log2(d))); return true;
        case OP_SIN: *out = make_const(e->dtype, to_bits(sin(d))); return true;
        case OP_COS: *out = make_const(e->dtype, to_bits(cos(d))); return true;
        case OP_TAN: *out = make_const(e->dtype, to_bits(tan(d))); return true;
        case OP_ASIN: *out = make_const(e->dtype, to_bits(asin(d))); return true;
        case OP_ACOS: *out = make_const(e->dtype, to_bits(acos(d))); return true;
        case OP_ATAN: *out = make_const(e->dtype, to_bits(atan(d))); return true;
        case OP_SINH: *out = make_const(e->dtype, to_bits(sinh(d))); return true;
        case OP_COSH: *out = make_const(e->dtype, to_bits(cosh(d))); return true;
        case OP_TANH: *out = make_const(e->dtype, to_bits(tanh(d))); return true;
        case OP_ASINH: *out = make_const(e->dtype, to_bits(asinh(d))); return true;
        case OP_ACOSH: *out = make_const(e->dtype, to_bits(acosh(d))); return true;
        case OP_ATANH: *out = make_const(e->dtype, to_bits(atanh(d))); return true;
        case OP_SQRT_QUIET: *out = make_const(e->dtype, to_bits(sqrt(d))); return true;
        case OP_SQRT_NULLING: *out = d < 0 ? make_null(e->dtype) : make_const(e->dtype, to_bits(sqrt(d))); return true;
        case OP_IS_ODD: case OP_IS_EVEN: {
          const int64_t v = dtype_width(x->dtype) == 4 ? (x->dtype == SSGPU_UINT32 ? (int64_t)(uint32_t)x->bits : (int64_t)(int32_t)(uint32_t)x->bits) : (int64_t)x->bits;
          const bool odd = x->dtype == SSGPU_UINT64 ? (x->bits & 1) : (v % 2) != 0;
          *out = make_const(SSGPU_BOOL, e->op == OP_IS_ODD ? odd : !odd); return true;
        }
        case OP_ABS:
          if (x->dtype == SSGPU_DOUBLE || f32) { *out = make_const(e->dtype, flt(d < 0 ? -d : d)); return true; }
          if (x->dtype == SSGPU_INT32) { const int32_t v = (int32_t)(uint32_t)x->bits; *out = make_const(e->dtype, v < 0 ? (uint32_t)(0u - (uint32_t)v) : (uint32_t)v); return true; }
          if (x->dtype == SSGPU_INT64) { const int64_t v = (int64_t)x->bits; *out = make_const(e->dtype, v < 0 ? 0ull - (uint64_t)v : (uint64_t)v); return true; }
          break;
        default: break;
      }
    }
    if (e->op == OP_NEGATE) {
      uint64_t b;
      if (x->dtype == SSGPU_DOUBLE) b = to_bits(-from_bits<double>(x->bits));
      else if (x->dtype == SSGPU_FLOAT) b = to_bits(-from_bits<float>(x->bits));
      else if (dtype_width(x->dtype) == 4) b = (uint32_t)(0u - (uint32_t)x->bits);
      else b = 0ull - x->bits;
      *out = make_const(e->dtype, b); return true;
    }
  }
  return false;
}
static BExprP fold(BExprP e) { BExprP f; return try_fold(e, &f) ? f : e; }

static Status check_type(int expected, const BExprP& e) {
  if (e->dtype != expected)
    return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH,
                         std::string("Expected ") + dtype_name(expected) + ", got " + dtype_name(e->dtype) + " in " + e->name);
  return Status::OK();
}

static Status bind_arith(int op, BExprP l, BExprP r, int depth, bool integer_only, BExprP* out) {
  int t;
  SS_RETURN_IF_ERROR(common_type(l->dtype, r->dtype, &t));
  if (!dtype_is_numeric(t) || (integer_only && !dtype_is_integer(t)))
    return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH,
                         std::string("Operator not defined for type ") + dtype_name(t));
  BExprP lc, rc;
  SS_RETURN_IF_ERROR(make_cast(l, t, true, &lc));
  SS_RETURN_IF_ERROR(make_cast(r, t, true, &rc));
  bool can_null = op == OP_CPP_DIVIDE_NULLING || op == OP_MODULUS_NULLING;
  *out = fold(make_op(op, t, lc->nullable || rc->nullable || can_null, fmt_binary(op, lc->name, rc->name), {lc, rc}, depth));
  return Status::OK();
}

static Status bind_divide(int op, BExprP l, BExprP r, int depth, BExprP* out) {
  // Divide always computes in DOUBLE (arithmetic_bound_expressions.cc:47-72)
  BExprP lc, rc;
  if (!dtype_is_numeric(l->dtype) || !dtype_is_numeric(r->dtype))
    return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH, "DIVIDE needs numeric arguments");
  SS_RETURN_IF_ERROR(make_cast(l, SSGPU_DOUBLE, true, &lc));
  SS_RETURN_IF_ERROR(make_cast(r, SSGPU_DOUBLE, true, &rc));
  bool can_null = op == OP_DIVIDE_NULLING;
  *out = fold(make_op(op, SSGPU_DOUBLE, lc->nullable || rc->nullable || can_null, fmt_binary(op, lc->name, rc->name), {lc, rc}, depth));
  return Status::OK();
}

static Status bind_compare(int op, BExprP l, BExprP r, int depth, BExprP* out) {
  // GenerateComparison (comparison_bound_expressions.cc:587-638)
  int lt = l->dtype, rt = r->dtype;
  BExprP lc = l, rc = r;
  if (lt != rt) {
    if (!dtype_is_numeric(lt) || !dtype_is_numeric(rt))
      return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH,
                           "Cannot compare expressions of different, non-numeric types");
    if (lt == SSGPU_DOUBLE || rt == SSGPU_DOUBLE) {
      SS_RETURN_IF_ERROR(make_cast(l, SSGPU_DOUBLE, true, &lc));
      SS_RETURN_IF_ERROR(make_cast(r, SSGPU_DOUBLE, true, &rc));
    } else if (lt == SSGPU_FLOAT || rt == SSGPU_FLOAT) {
      SS_RETURN_IF_ERROR(make_cast(l, SSGPU_FLOAT, false, &lc));
      SS_RETURN_IF_ERROR(make_cast(r, SSGPU_FLOAT, false, &rc));
    }
    // two different integer types: compared directly, no casts (value-correct functors)
  } else if (dtype_width(lt) == 0) {
    return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "comparison of variable-length types is outside the device hot path");
  }
  BExprP e = make_op(op, SSGPU_BOOL, lc->nullable || rc->nullable, fmt_binary(op, lc->name, rc->name), {lc, rc}, depth);
  // fold only equal-typed constants (mixed integer constants are rare; keep them on device)
  if (lc->dtype == rc->dtype) e = fold(e);
  *out = e;
  return Status::OK();
}

static ssgpu_expr x_expr_for(int op) { ssgpu_expr e; memset(&e, 0, sizeof(e)); e.kind = SSGPU_EXPR_OP; e.op = op; return e; }

static Status bind_operator(const ssgpu_expr& x, std::vector<BExprP> args, int depth, BExprP* out) {
  const int op = x.op;
  auto need = [&](size_t n) -> Status {
    if (args.size() != n)
      return Status::Error(SSGPU_ERROR_ATTRIBUTE_COUNT_MISMATCH, "operator " + std::to_string(op) + " expects " +
                                                                     std::to_string(n) + " single-attribute arguments");
    return Status::OK();
  };
  switch (op) {
    case OP_ADD: case OP_SUBTRACT: case OP_MULTIPLY:
      SS_RETURN_IF_ERROR(need(2)); return bind_arith(op, args[0], args[1], depth, false, out);
    case OP_CPP_DIVIDE_NULLING: case OP_CPP_DIVIDE_SIGNALING:
      SS_RETURN_IF_ERROR(need(2)); return bind_arith(op, args[0], args[1], depth, false, out);
    case OP_MODULUS_NULLING: case OP_MODULUS_SIGNALING:
      SS_RETURN_IF_ERROR(need(2)); return bind_arith(op, args[0], args[1], depth, true, out);
    case OP_BITWISE_AND: case OP_BITWISE_OR: case OP_BITWISE_XOR: case OP_BITWISE_ANDNOT:
      SS_RETURN_IF_ERROR(need(2)); return bind_arith(op, args[0], args[1], depth, true, out);
    case OP_SHIFT_LEFT: case OP_SHIFT_RIGHT:
      // CreateShiftExpression (elementary_bound_expressions.cc:1446-1489): the result inherits the LEFT
      // type; the shift count may be any integer type and is not promoted
      SS_RETURN_IF_ERROR(need(2));
      if (!dtype_is_integer(args[0]->dtype) || !dtype_is_integer(args[1]->dtype))
        return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH, "Shift needs integer arguments");
      *out = fold(make_op(op, args[0]->dtype, args[0]->nullable || args[1]->nullable,
                          fmt_binary(op, args[0]->name, args[1]->name), {args[0], args[1]}, depth));
      return Status::OK();
    case OP_DIVIDE_QUIET: case OP_DIVIDE_NULLING: case OP_DIVIDE_SIGNALING:
      SS_RETURN_IF_ERROR(need(2)); return bind_divide(op, args[0], args[1], depth, out);
    case OP_EQUAL: case OP_NOT_EQUAL: case OP_LESS: case OP_LESS_OR_EQUAL:
      SS_RETURN_IF_ERROR(need(2)); return bind_compare(op, args[0], args[1], depth, out);
    case SSGPU_OP_GREATER:  // a > b  ==  Less(b, a)   (comparison_bound_expressions.cc:832-839)
      SS_RETURN_IF_ERROR(need(2)); return bind_compare(OP_LESS, args[1], args[0], depth, out);
    case SSGPU_OP_GREATER_OR_EQUAL:
      SS_RETURN_IF_ERROR(need(2)); return bind_compare(OP_LESS_OR_EQUAL, args[1], args[0], depth, out);
    case OP_AND: case OP_OR: case OP_AND_NOT: case OP_XOR: {
      SS_RETURN_IF_ERROR(need(2));
      SS_RETURN_IF_ERROR(check_type(SSGPU_BOOL, args[0]));
      SS_RETURN_IF_ERROR(check_type(SSGPU_BOOL, args[1]));
      *out = fold(make_op(op, SSGPU_BOOL, args[0]->nullable || args[1]->nullable,
                          fmt_binary(op, args[0]->name, args[1]->name), {args[0], args[1]}, depth));
      return Status::OK();
    }
    case OP_NOT:
      SS_RETURN_IF_ERROR(need(1));
      SS_RETURN_IF_ERROR(check_type(SSGPU_BOOL, args[0]));
      *out = fold(make_op(op, SSGPU_BOOL, args[0]->nullable, "(NOT " + args[0]->name + ")", {args[0]}, depth));
      return Status::OK();
    case OP_NEGATE: {
      SS_RETURN_IF_ERROR(need(1));
      // CreateUnarySignedNumericExpression: unsigned inputs are promoted to the signed type
      int t = args[0]->dtype;
      if (!dtype_is_numeric(t)) return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH, "NEGATE needs a numeric argument");
      int st = t == SSGPU_UINT32 ? SSGPU_INT32 : t == SSGPU_UINT64 ? SSGPU_INT64 : t;
      BExprP c = args[0];
      if (st != t) {  // projecting cast (same width), cast_bound_expression.cc:328,349
        if (c->kind == BExpr::CONST) c = make_const(st, c->bits);
        else if (c->kind == BExpr::NULLCONST) c = make_null(st);
        else {
          BExprP e(new BExpr); e->kind = BExpr::CAST; e->op = OP_CAST_QUIET; e->dtype = st; e->nullable = c->nullable;
          e->filter_depth = depth;
          e->name = std::string("CAST_") + dtype_name(t) + "_TO_" + dtype_name(st) + "(" + c->name + ")";
          e->args.push_back(c); c = e;
        }
      }
      // the signed-type unary factory reinterprets an unsigned argument without a CAST node of its own: the name shows
      // the argument itself (arithmetic_expressions_test.cc:28, arithmetic_bound_expressions_test.cc:26-29: "(-$0)" for UINT32)
      *out = fold(make_op(op, st, c->nullable, "(-" + args[0]->name + ")", {c}, depth));
      return Status::OK();
    }
    case OP_BITWISE_NOT: {
      SS_RETURN_IF_ERROR(need(1));
      if (!dtype_is_integer(args[0]->dtype)) return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH, "BITWISE NOT needs an integer argument");
      *out = fold(make_op(op, args[0]->dtype, args[0]->nullable, "(~" + args[0]->name + ")", {args[0]}, depth));
      return Status::OK();
    }
    case OP_IS_NULL:
      SS_RETURN_IF_ERROR(need(1));
      if (!args[0]->nullable) { *out = make_const(SSGPU_BOOL, 0); return Status::OK(); }  // elementary_bound_expressions.cc:1419-1424
      *out = fold(make_op(op, SSGPU_BOOL, false, "ISNULL(" + args[0]->name + ")", {args[0]}, depth));
      return Status::OK();
    case OP_IF_NULL: {
      SS_RETURN_IF_ERROR(need(2));
      int t; SS_RETURN_IF_ERROR(common_type(args[0]->dtype, args[1]->dtype, &t));
      BExprP lc, rc;
      SS_RETURN_IF_ERROR(make_cast(args[0], t, true, &lc));
      SS_RETURN_IF_ERROR(make_cast(args[1], t, true, &rc));
      if (!lc->nullable) { *out = lc; return Status::OK(); }
      *out = make_op(op, t, rc->nullable, fmt_binary(op, lc->name, rc->name), {lc, rc}, depth);
      return Status::OK();
    }
    // ---- exact math family: math_bound_expressions.cc:150-170,318-456,473-486 (names "OP(child)") ----
    case OP_ABS: {
      SS_RETURN_IF_ERROR(need(1));
      const int t = args[0]->dtype;
      if (t == SSGPU_UINT32 || t == SSGPU_UINT64) { *out = args[0]; return Status::OK(); }
      int ot = t == SSGPU_INT32 ? SSGPU_UINT32 : t == SSGPU_INT64 ? SSGPU_UINT64 : (t == SSGPU_FLOAT || t == SSGPU_DOUBLE) ? t : -1;
      if (ot < 0) return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH, std::string("ABS is not defined for ") + dtype_name(t));
      *out = fold(make_op(op, ot, args[0]->nullable, "ABS(" + args[0]->name + ")", {args[0]}, depth));
      return Status::OK();
    }
    case OP_ROUND: case OP_CEIL: case OP_FLOOR: case OP_TRUNC: case OP_CEIL_TO_INT: case OP_FLOOR_TO_INT: case OP_ROUND_TO_INT: {
      SS_RETURN_IF_ERROR(need(1));
      const int t = args[0]->dtype;
      if (dtype_is_integer(t)) { *out = args[0]; return Status::OK(); }   // integers are already rounded
      if (t != SSGPU_FLOAT && t != SSGPU_DOUBLE)
        return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH, std::string("rounding is not defined for ") + dtype_name(t));
      auto unary = [&](int o, const char* nm, int ot, BExprP child) {
        return fold(make_op(o, ot, child->nullable, std::string(nm) + "(" + child->name + ")", {child}, depth));
      };
      switch (op) {
        case OP_ROUND: *out = unary(op, "ROUND", t, args[0]); break;
        case OP_CEIL: *out = unary(op, "CEIL", t, args[0]); break;
        case OP_FLOOR: *out = unary(op, "FLOOR", t, args[0]); break;
        case OP_TRUNC: *out = unary(op, "TRUNC", t, args[0]); break;
        case OP_CEIL_TO_INT: *out = unary(op, "CEIL_TO_INT", SSGPU_INT64, args[0]); break;
        case OP_FLOOR_TO_INT: *out = unary(op, "FLOOR_TO_INT", SSGPU_INT64, args[0]); break;
        default:  // RoundToInt binds as CEIL_TO_INT(ROUND(x)) (math_bound_expressions.cc:327-339)
          *out = unary(OP_CEIL_TO_INT, "CEIL_TO_INT", SSGPU_INT64, unary(OP_ROUND, "ROUND", t, args[0])); break;
      }
      return Status::OK();
    }
    // libm family (math_bound_expressions.cc:44-92,126-283, expression_traits.h:718-1000,1329-1375): promoting
    // unary / binary expressions over DOUBLE; LN / LOG10 / LOG2 NULLING are NULL for x <= 0, POW NULLING /
    // SIGNALING for a negative base with a non-integer exponent
    case OP_EXP: case OP_LN_QUIET: case OP_LN_NULLING: case OP_LOG10_QUIET: case OP_LOG10_NULLING: case OP_LOG2_QUIET: case OP_LOG2_NULLING:
    case OP_SIN: case OP_COS: case OP_TAN: case OP_ASIN: case OP_ACOS: case OP_ATAN: case OP_SINH: case OP_COSH: case OP_TANH:
    case OP_ASINH: case OP_ACOSH: case OP_ATANH: {
      SS_RETURN_IF_ERROR(need(1));
      if (!dtype_is_numeric(args[0]->dtype))
        return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH, std::string("Cannot cast ") + dtype_name(args[0]->dtype) + " to DOUBLE");
      BExprP c;
      SS_RETURN_IF_ERROR(make_cast(args[0], SSGPU_DOUBLE, true, &c));
      const char* nm = "?";
      switch (op) {
        case OP_EXP: nm = "EXP"; break; case OP_LN_QUIET: case OP_LN_NULLING: nm = "LN"; break;
        case OP_LOG10_QUIET: case OP_LOG10_NULLING: nm = "LOG10"; break; case OP_LOG2_QUIET: case OP_LOG2_NULLING: nm = "LOG2"; break;
        case OP_SIN: nm = "SIN"; break; case OP_COS: nm = "COS"; break; case OP_TAN: nm = "TAN"; break; case OP_ASIN: nm = "ASIN"; break;
        case OP_ACOS: nm = "ACOS"; break; case OP_ATAN: nm = "ATAN"; break; case OP_SINH: nm = "SINH"; break; case OP_COSH: nm = "COSH"; break;
        case OP_TANH: nm = "TANH"; break; case OP_ASINH: nm = "ASINH"; break; case OP_ACOSH: nm = "ACOSH"; break; default: nm = "ATANH"; break;
      }
      const bool nulling = op == OP_LN_NULLING || op == OP_LOG10_NULLING || op == OP_LOG2_NULLING;
      *out = fold(make_op(op, SSGPU_DOUBLE, c->nullable || nulling, std::string(nm) + "(" + c->name + ")", {c}, depth));
      if (nulling && (*out)->kind == BExpr::OP) (*out)->nullable = true;
      return Status::OK();
    }
    case SSGPU_OP_ROUND_WITH_PRECISION: {
      // BoundRoundWithPrecision (math_bound_expressions.cc:341-382): round(x * m) / m with m = POW_QUIET(10.0, precision)
      SS_RETURN_IF_ERROR(need(2));
      if (!dtype_is_integer(args[1]->dtype))
        return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH, std::string("Wrong type of argument supplied. Precision has to be an integer; is: ") + dtype_name(args[1]->dtype));
      if (!dtype_is_numeric(args[0]->dtype))
        return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH, std::string("Cannot cast ") + dtype_name(args[0]->dtype) + " to DOUBLE");
      BExprP x, pw, ten = make_const(SSGPU_DOUBLE, to_bits(10.0));
      SS_RETURN_IF_ERROR(make_cast(args[0], SSGPU_DOUBLE, true, &x));
      ssgpu_expr px = x_expr_for(OP_POW_QUIET);
      SS_RETURN_IF_ERROR(bind_operator(px, {ten, args[1]}, depth, &pw));
      *out = fold(make_op(OP_ROUND_WITH_MULTIPLIER, SSGPU_DOUBLE, x->nullable || pw->nullable,
                          "ROUND_WITH_MULTIPLIER(" + x->name + ", " + pw->name + ")", {x, pw}, depth));
      return Status::OK();
    }
    case OP_POW_QUIET: case OP_POW_NULLING: case OP_POW_SIGNALING: case OP_ATAN2: {
      SS_RETURN_IF_ERROR(need(2));
      if (!dtype_is_numeric(args[0]->dtype) || !dtype_is_numeric(args[1]->dtype))
        return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH, std::string("Cannot cast ") + dtype_name(args[0]->dtype) + " to DOUBLE");
      BExprP l, r;
      SS_RETURN_IF_ERROR(make_cast(args[0], SSGPU_DOUBLE, true, &l));
      SS_RETURN_IF_ERROR(make_cast(args[1], SSGPU_DOUBLE, true, &r));
      const std::string name = std::string(op == OP_ATAN2 ? "ATAN2" : "POW") + "(" + l->name + ", " + r->name + ")";
      BExprP e = make_op(op, SSGPU_DOUBLE, l->nullable || r->nullable || op == OP_POW_NULLING, name, {l, r}, depth);
      *out = op == OP_POW_SIGNALING ? e : fold(e);   // a constant failing POW_SIGNALING must still fail when evaluated
      if (op == OP_POW_NULLING && (*out)->kind == BExpr::OP) (*out)->nullable = true;
      return Status::OK();
    }
    case OP_SQRT_QUIET: case OP_SQRT_NULLING: case OP_SQRT_SIGNALING:
    case OP_IS_FINITE: case OP_IS_INF: case OP_IS_NAN: case OP_IS_NORMAL: {
      SS_RETURN_IF_ERROR(need(1));
      if (!dtype_is_numeric(args[0]->dtype))
        return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH, std::string("Cannot cast ") + dtype_name(args[0]->dtype) + " to DOUBLE");
      BExprP c;
      SS_RETURN_IF_ERROR(make_cast(args[0], SSGPU_DOUBLE, true, &c));   // promoting unary expressions
      const bool is_sqrt = op == OP_SQRT_QUIET || op == OP_SQRT_NULLING || op == OP_SQRT_SIGNALING;
      const char* nm = is_sqrt ? "SQRT" : op == OP_IS_FINITE ? "IS_FINITE" : op == OP_IS_INF ? "IS_INF" : op == OP_IS_NAN ? "IS_NAN" : "IS_NORMAL";
      BExprP e = make_op(op, is_sqrt ? SSGPU_DOUBLE : SSGPU_BOOL, c->nullable || op == OP_SQRT_NULLING, std::string(nm) + "(" + c->name + ")", {c}, depth);
      // a constant negative SQRT_SIGNALING must still fail at evaluation time: never folded
      *out = op == OP_SQRT_SIGNALING ? e : fold(e);
      if (op == OP_SQRT_NULLING && (*out)->kind == BExpr::OP) (*out)->nullable = true;
      return Status::OK();
    }
    case OP_IS_ODD: case OP_IS_EVEN: {
      SS_RETURN_IF_ERROR(need(1));
      if (!dtype_is_integer(args[0]->dtype))
        return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH, std::string(op == OP_IS_ODD ? "IS_ODD" : "IS_EVEN") + " needs an integer argument");
      *out = fold(make_op(op, SSGPU_BOOL, args[0]->nullable, std::string(op == OP_IS_ODD ? "IS_ODD(" : "IS_EVEN(") + args[0]->name + ")", {args[0]}, depth));
      return Status::OK();
    }
    case OP_CASE: {
      // CASE arg0 WHEN arg2 THEN arg3 ... ELSE arg1 (BoundCase, elementary_bound_expressions.cc:1297-1356).
      // Bound as a chain of plain IFs over equality tests, which has exactly the reference's
      // semantics (:556-600): a NULL CASE value or a NULL WHEN never matches, the first match
      // wins, and the result is NULL iff the selected THEN / OTHERWISE is.
      const size_t n = args.size();
      if (n < 2) return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_VALUE, "Case expects at least 2 arguments (make sense from 4 arguments).");
      if (n % 2 != 0) return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_VALUE, "Case expects odd number of arguments.");
      int test_type = args[0]->dtype, out_type = args[1]->dtype;
      for (size_t i = 2; i < n; ++i) {
        int* expected = (i % 2 == 0) ? &test_type : &out_type;
        if (args[i]->dtype == *expected) continue;
        if (!dtype_is_numeric(args[i]->dtype) || !dtype_is_numeric(*expected))
          return Status::Error(SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH, std::string("Bind failed: Case: Cannot cast attribute ") + std::to_string(i) +
                               " (" + dtype_name(args[i]->dtype) + " to " + dtype_name(*expected) + ")");
        int t; SS_RETURN_IF_ERROR(common_type(args[i]->dtype, *expected, &t));
        *expected = t;
      }
      std::vector<BExprP> c(n);
      std::string name = "CASE(";
      bool nullable = false, all_const = true;
      for (size_t i = 0; i < n; ++i) {
        SS_RETURN_IF_ERROR(make_cast(args[i], i % 2 == 0 ? test_type : out_type, true, &c[i]));
        if (i % 2 == 1 && c[i]->nullable) nullable = true;
        if (c[i]->kind != BExpr::CONST && c[i]->kind != BExpr::NULLCONST) all_const = false;
        name += (i ? ", " : "") + c[i]->name;
      }
      name += ")";
      BExprP result = c[1];
      for (size_t i = n; i >= 4; i -= 2) {   // last WHEN/THEN pair innermost
        BExprP cond;
        SS_RETURN_IF_ERROR(bind_compare(OP_EQUAL, c[0], c[i - 2], depth, &cond));
        if (cond->kind == BExpr::NULLCONST || (cond->kind == BExpr::CONST && !cond->bits)) continue;   // can never match
        if (cond->kind == BExpr::CONST) { result = c[i - 1]; continue; }                                 // always matches
        result = make_op(OP_IF, out_type, c[i - 1]->nullable || result->nullable, "", {cond, c[i - 1], result}, depth);
      }
      if (all_const) { *out = result; return Status::OK(); }   // every argument constant: folded, as InitBasicExpression does
      // Otherwise the schema is the unfolded BoundCase's -- its name, and NULLABLE iff any THEN / OTHERWISE
      // is -- also when WHENs that can never (or always) match were dropped from the chain.  A chain that
      // collapsed to one of its arguments stays an expression node of its own, so that nothing above folds
      // through it and its schema entry is not mistaken for the argument's.
      bool collapsed = false;   // the chain is one of the (cast) arguments itself: a constant, a column, ...
      for (size_t i = 1; i < n; i += 2) collapsed = collapsed || result == c[i];
      if (collapsed)
        result = make_op(OP_IF, out_type, nullable, "", {make_const(SSGPU_BOOL, 1), result, result}, depth);
      BExprP named(new BExpr(*result));
      named->name = name;
      named->nullable = nullable;
      *out = named;
      return Status::OK();
    }
    case OP_IN: {
      // expr IN (value, ...): BoundInSet, comparison_bound_expressions.cc:759-813.  Bound as the
      // three-valued OR of equality tests: TRUE on a match, else NULL if the needle or a list
      // element is NULL, else FALSE (comparison_expressions.h:75-84).
      const size_t n = args.size();
      if (n < 1) return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_VALUE, "IN needs a needle expression");
      int t = args[0]->dtype;
      for (size_t i = 1; i < n; ++i) { int u; SS_RETURN_IF_ERROR(common_type(t, args[i]->dtype, &u)); t = u; }
      std::vector<BExprP> c(n);
      bool nullable = false;
      for (size_t i = 0; i < n; ++i) {
        SS_RETURN_IF_ERROR(make_cast(args[i], t, true, &c[i]));
        nullable = nullable || c[i]->nullable;
      }
      std::string name = c[0]->name + " IN (";
      for (size_t i = 1; i < n; ++i) name += (i > 1 ? ", " : "") + c[i]->name;
      name += ")";
      BExprP result;
      if (n == 1) {
        // "$0 IN ()": FALSE, or NULL for a NULL needle -- (x == x) XOR (x == x) has exactly that
        // value for every x, NaN included, and keeps the node non-constant like the reference's
        BExprP eq;
        SS_RETURN_IF_ERROR(bind_compare(OP_EQUAL, c[0], c[0], depth, &eq));
        result = make_op(OP_XOR, SSGPU_BOOL, eq->nullable, "", {eq, eq}, depth);
      }
      for (size_t i = 1; i < n; ++i) {
        BExprP eq;
        SS_RETURN_IF_ERROR(bind_compare(OP_EQUAL, c[0], c[i], depth, &eq));
        if (i == 1) { result = eq; continue; }
        result = fold(make_op(OP_OR, SSGPU_BOOL, result->nullable || eq->nullable, "", {result, eq}, depth));
      }
      BExprP named(new BExpr(*result));
      if (named->kind != BExpr::CONST && named->kind != BExpr::NULLCONST) { named->name = name; named->nullable = nullable; }
      *out = named;
      return Status::OK();
    }
    case OP_IF: case SSGPU_OP_NULLING_IF: {
      SS_RETURN_IF_ERROR(need(3));
      SS_RETURN_IF_ERROR(check_type(SSGPU_BOOL, args[0]));
      int t; SS_RETURN_IF_ERROR(common_type(args[1]->dtype, args[2]->dtype, &t));
      BExprP tc, ec;
      SS_RETURN_IF_ERROR(make_cast(args[1], t, true, &tc));
      SS_RETURN_IF_ERROR(make_cast(args[2], t, true, &ec));
      // plain (non-nulling) IF: a NULL condition takes the ELSE branch and does not make the
      // result NULL (elementary_bound_expressions.cc:893-904,1010-1025)
      // NullingIf: NULLs of the condition are viral too (CreateIfSchema, :1010-1016)
      *out = make_op(op, t, tc->nullable || ec->nullable || (op == SSGPU_OP_NULLING_IF && args[0]->nullable),
                     "IF " + args[0]->name + " THEN " + tc->name + " ELSE " + ec->name, {args[0], tc, ec}, depth);
      return Status::OK();
    }
  }
  return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED,
                       "OperatorId " + std::to_string(op) + " is outside the device hot path (SURVEY 8: math/date/string/regexp)");
}

Status bind_expression(const PlanDesc& d, int idx, const Schema& schema, int depth, std::vector<BExprP>* out) {
  if (idx < 0 || idx >= (int)d.exprs.size()) return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_VALUE, "bad expression index");
  const ssgpu_expr& x = d.exprs[idx];
  auto input = [&](int pos) {
    BExprP e(new BExpr);
    e->kind = BExpr::INPUT; e->input_col = pos; e->dtype = schema[pos].dtype; e->nullable = schema[pos].nullable;
    e->name = schema[pos].name; e->filter_depth = depth;
    return e;
  };
  switch (x.kind) {
    case SSGPU_EXPR_ATTR_NAMED: {
      int pos = lookup_attr(schema, x.name);
      if (pos < 0)
        return Status::Error(SSGPU_ERROR_ATTRIBUTE_MISSING,
                             std::string("No attribute '") + x.name + "' in the schema:\n '" + schema_to_string(schema) + "'");
      out->push_back(input(pos));
      return Status::OK();
    }
    case SSGPU_EXPR_ATTR_AT:
      if (x.i64 < 0 || x.i64 >= (int64_t)schema.size())
        return Status::Error(SSGPU_ERROR_ATTRIBUTE_COUNT_MISMATCH,
                             "source schema has too few attributes (" + std::to_string(schema.size()) + " vs " + std::to_string(x.i64) + ")");
      out->push_back(input((int)x.i64));
      return Status::OK();
    case SSGPU_EXPR_CONST:
      if (dtype_width(x.dtype) == 0) return Status::Error(SSGPU_ERROR_NOT_IMPLEMENTED, "variable-length constants are outside the device hot path");
      out->push_back(make_const(x.dtype, const_bits(x.dtype, x.i64, x.f64)));
      return Status::OK();
    case SSGPU_EXPR_NULL:
      out->push_back(make_null(x.dtype));
      return Status::OK();
    case SSGPU_EXPR_ALIAS: {
      if (x.nargs != 1) return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_VALUE, "ALIAS takes one child");
      std::vector<BExprP> child;
      SS_RETURN_IF_ERROR(bind_expression(d, d.expr_args[x.first_arg], schema, depth, &child));
      if (child.size() != 1)
        return Status::Error(SSGPU_ERROR_ATTRIBUTE_COUNT_MISMATCH,
                             "Number of aliases (1) does not match the attribute count in source schema (" + std::to_string(child.size()) + ")");
      BExprP e(new BExpr(*child[0]));  // same computation, new name
      e->name = x.name;
      out->push_back(e);
      return Status::OK();
    }
    case SSGPU_EXPR_COMPOUND: {
      for (int i = 0; i < x.nargs; ++i)
        SS_RETURN_IF_ERROR(bind_expression(d, d.expr_args[x.first_arg + i], schema, depth, out));
      for (size_t i = 0; i < out->size(); ++i)
        for (size_t j = i + 1; j < out->size(); ++j)
          if ((*out)[i]->name == (*out)[j]->name)
            return Status::Error(SSGPU_ERROR_ATTRIBUTE_EXISTS, "Duplicate attribute name \"" + (*out)[i]->name + "\" in result schema");
      return Status::OK();
    }
    case SSGPU_EXPR_CAST: {
      if (x.nargs != 1) return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_VALUE, "CAST takes one child");
      std::vector<BExprP> child;
      SS_RETURN_IF_ERROR(bind_expression(d, d.expr_args[x.first_arg], schema, depth, &child));
      if (child.size() != 1) return Status::Error(SSGPU_ERROR_ATTRIBUTE_COUNT_MISMATCH, "Cast expects a single attribute");
      BExprP e;
      SS_RETURN_IF_ERROR(make_cast(child[0], x.dtype, false, &e));
      out->push_back(e);
      return Status::OK();
    }
    case SSGPU_EXPR_OP: {
      std::vector<BExprP> args;
      for (int i = 0; i < x.nargs; ++i) {
        std::vector<BExprP> child;
        SS_RETURN_IF_ERROR(bind_expression(d, d.expr_args[x.first_arg + i], schema, depth, &child));
        if (child.size() != 1)
          return Status::Error(SSGPU_ERROR_ATTRIBUTE_COUNT_MISMATCH, "operator arguments must have exactly one attribute");
        args.push_back(child[0]);
      }
      BExprP e;
      SS_RETURN_IF_ERROR(bind_operator(x, args, depth, &e));
      out->push_back(e);
      return Status::OK();
    }
  }
  return Status::Error(SSGPU_ERROR_INVALID_ARGUMENT_VALUE, "unknown expression kind");
}

std::string bexpr_to_string(const BExprP& e) {
  std::string s = e->name + ":" + dtype_name(e->dtype) + (e->nullable ? "?" : "");
  return s;
}

}  // namespace ssgpu
