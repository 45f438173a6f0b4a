/*
 * ss_oracle.c -- CPU restatement of Supersonic's Filter -> Project/Compute ->
 * Aggregate (+Sort) path, in plain C.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it; the product path
 * (libssgpu.so) never links, calls or falls back to anything in oracle/.
 *
 * It restates the reference's algorithms the way the reference executes them
 * ("reference mode"): a pull model over <=1024-row blocks, materialised
 * intermediate columns per expression node, an int64 row-id list + gather for
 * Filter, one pass per aggregate column, a chained-bucket hash set for
 * GroupAggregate that emits groups in first-seen order, and a comparison sort
 * of an int64 permutation for Sort.  Every function cites the reference
 * file:line it follows (paths relative to the reference tree).
 *
 * Pinning: the reference itself cannot be built in this image without writing
 * stand-ins for glog/gflags/protobuf/boost, so the oracle is
 * pinned against the golden vectors of the reference's own tests, transcribed
 * as data under tests/golden/ (tests/test_oracle_golden.py).
 *
 * Build: gcc -O3 -msse2 -funsigned-char -shared -fPIC (the reference's flags,
 * configure.ac:36-40).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_BLOCK 1024 /* Cursor::kDefaultRowCount, cursor/base/cursor.h:133 */

/* DataType values, supersonic/proto/supersonic.proto:15-36 */
enum { T_INT32 = 1, T_INT64 = 2, T_UINT64 = 3, T_DATETIME = 4, T_DOUBLE = 5, T_BOOL = 6,
       T_UINT32 = 8, T_FLOAT = 9, T_DATE = 10, T_STRING = 0, T_BINARY = 7 };
/* Aggregation values, supersonic.proto:86-94 */
enum { A_SUM = 0, A_MIN = 1, A_MAX = 2, A_COUNT = 3, A_CONCAT = 4, A_FIRST = 5, A_LAST = 6,
       A_SUM_RESIDUAL = 100 /* not a reference aggregation (include/ssgpu.h): what the product's compensated DOUBLE sum did not fit in
                               its rounded result; the reference's sequential fold has no such term -- here always 0.0 (NULL with its SUM) */ };
/* ReturnCode values, supersonic.proto:40-82 */
enum { RC_OK = 0, RC_NOT_IMPLEMENTED = 103, RC_EVALUATION_ERROR = 104, RC_COUNT_MISMATCH = 401,
       RC_TYPE_MISMATCH = 402, RC_ATTRIBUTE_MISSING = 403, RC_ATTRIBUTE_EXISTS = 404,
       RC_INVALID_ARGUMENT_TYPE = 405, RC_INVALID_ARGUMENT_VALUE = 407 };
/* OperatorId values, supersonic/expression/proto/operators.proto */
enum { OP_ADD = 0, OP_MULTIPLY = 4, OP_SUBTRACT = 8, OP_DIVIDE_QUIET = 13, OP_DIVIDE_NULLING = 14,
       OP_DIVIDE_SIGNALING = 15, OP_CPP_DIVIDE_NULLING = 18, OP_CPP_DIVIDE_SIGNALING = 19,
       OP_MODULUS_NULLING = 26, OP_MODULUS_SIGNALING = 27, OP_NEGATE = 36, OP_AND = 40, OP_OR = 44,
       OP_AND_NOT = 48, OP_NOT = 52, OP_XOR = 56, OP_BITWISE_AND = 60, OP_BITWISE_OR = 64, OP_BITWISE_NOT = 68,
       OP_BITWISE_XOR = 72, OP_SHIFT_LEFT = 76, OP_SHIFT_RIGHT = 80, OP_BITWISE_ANDNOT = 84, OP_EQUAL = 100, OP_NOT_EQUAL = 104, OP_LESS = 116,
       OP_LESS_OR_EQUAL = 120, OP_IS_ODD = 140, OP_IS_EVEN = 144, OP_IS_FINITE = 148, OP_IS_INF = 152, OP_IS_NAN = 156, OP_IS_NORMAL = 160,
       OP_ROUND = 300, OP_TRUNC = 304, OP_CEIL_TO_INT = 308, OP_FLOOR_TO_INT = 312, OP_ROUND_TO_INT = 316,
       OP_EXP = 320, OP_LN_QUIET = 325, OP_LN_NULLING = 326, OP_LOG10_QUIET = 329, OP_LOG10_NULLING = 330, OP_POW_QUIET = 353,
       OP_POW_NULLING = 354, OP_POW_SIGNALING = 355, OP_LOG2_QUIET = 357, OP_LOG2_NULLING = 358, OP_SIN = 800, OP_COS = 804, OP_TAN = 808,
       OP_ASIN = 812, OP_ACOS = 816, OP_ATAN = 820, OP_ATAN2 = 824, OP_SINH = 828, OP_COSH = 832, OP_TANH = 836, OP_ASINH = 840,
       OP_ACOSH = 844, OP_ATANH = 848,
       OP_SQRT_QUIET = 333, OP_SQRT_NULLING = 334, OP_SQRT_SIGNALING = 335, OP_CEIL = 342, OP_FLOOR = 346, OP_ABS = 360,
       OP_CASE = 200, OP_IF = 204, OP_IN = 208, OP_IF_NULL = 220, OP_IS_NULL = 224, OP_CAST = 265,
       OP_ROUND_WITH_MULTIPLIER = 364, OP_GREATER = 100001, OP_GREATER_OR_EQUAL = 100002, OP_NULLING_IF = 100003,
       OP_ROUND_WITH_PRECISION = 100004 };

typedef struct { int code; char msg[512]; } orc_error;
static void set_err(orc_error* e, int code, const char* fmt, const char* a, const char* b) {
  if (!e || e->code) return;
  e->code = code;
  snprintf(e->msg, sizeof(e->msg), fmt, a ? a : "", b ? b : "");
}

static const char* type_name(int t) {
  switch (t) {
    case T_INT32: return "INT32"; case T_INT64: return "INT64"; case T_UINT32: return "UINT32";
    case T_UINT64: return "UINT64"; case T_FLOAT: return "FLOAT"; case T_DOUBLE: return "DOUBLE";
    case T_BOOL: return "BOOL"; case T_DATE: return "DATE"; case T_DATETIME: return "DATETIME";
    case T_STRING: return "STRING"; case T_BINARY: return "BINARY";
  }
  return "?";
}
static int type_width(int t) {
  switch (t) {
    case T_INT32: case T_UINT32: case T_FLOAT: case T_DATE: return 4;
    case T_INT64: case T_UINT64: case T_DOUBLE: case T_DATETIME: return 8;
    case T_BOOL: return 1;
    /* STRING columns reach this restatement as INT32 codes of an order-preserving dictionary
     * (oracle.py encodes them): comparisons, MIN/MAX, keys and sort order of the codes are those
     * of the strings (StringPiece compare = memcmp then length) */
    case T_STRING: return 4;
  }
  return 0;
}
static int is_integer(int t) { return t == T_INT32 || t == T_INT64 || t == T_UINT32 || t == T_UINT64; }
static int is_float(int t) { return t == T_FLOAT || t == T_DOUBLE; }
static int is_numeric(int t) { return is_integer(t) || is_float(t); }

/* ---- schema / views (base/infrastructure/tuple_schema.h:77,126; block.h:288) ---- */
#define ORC_MAX_COLS 64
typedef struct { char name[256]; int type; int nullable; } orc_attr;
typedef struct { int n; orc_attr a[ORC_MAX_COLS]; } orc_schema;
typedef struct { const void* data; const uint8_t* is_null; } orc_col; /* is_null NULL => no NULLs */
typedef struct { int n; int64_t rows; orc_col c[ORC_MAX_COLS]; } orc_view;

static int schema_lookup(const orc_schema* s, const char* name) {
  for (int i = 0; i < s->n; ++i) if (strcmp(s->a[i].name, name) == 0) return i;
  return -1;
}
static int schema_add(orc_schema* s, const char* name, int type, int nullable) {
  if (schema_lookup(s, name) >= 0 || s->n >= ORC_MAX_COLS) return 0;
  snprintf(s->a[s->n].name, sizeof(s->a[s->n].name), "%s", name);
  s->a[s->n].type = type; s->a[s->n].nullable = nullable; s->n++;
  return 1;
}

/* =========================== expressions ====================================== */
#define ORC_MAX_ARGS 128
enum { E_NAMED = 1, E_AT = 2, E_CONST = 3, E_NULL = 4, E_OP = 5, E_ALIAS = 6, E_COMPOUND = 7, E_CAST = 8 };
typedef struct orc_expr {
  int kind, op, dtype, nargs;
  struct orc_expr* args[ORC_MAX_ARGS];
  int64_t i64; double f64;
  char name[256];
} orc_expr;

orc_expr* orc_expr_new(int kind, int op, int dtype, int64_t i64, double f64, const char* name) {
  orc_expr* e = (orc_expr*)calloc(1, sizeof(orc_expr));
  e->kind = kind; e->op = op; e->dtype = dtype; e->i64 = i64; e->f64 = f64;
  if (name) snprintf(e->name, sizeof(e->name), "%s", name);
  return e;
}
/* more arguments than the checker holds must not pass for a shorter expression: abort (test infrastructure) */
void orc_expr_add_arg(orc_expr* e, orc_expr* a) {
  if (e->nargs >= ORC_MAX_ARGS) { fprintf(stderr, "ss_oracle: more than %d arguments\n", ORC_MAX_ARGS); abort(); }
  e->args[e->nargs++] = a;
}

/* bound node: typed, owns a 1024-row result block
 * (BasicBoundExpression, expression/infrastructure/basic_bound_expression.h:49) */
enum { B_INPUT, B_CONST, B_NULLCONST, B_OP, B_CAST };
typedef struct bnode {
  int kind, op, dtype, nullable, input_col, nargs;
  struct bnode* args[ORC_MAX_ARGS];
  uint64_t bits;
  char name[256];
  void* buf;          /* result data, ORC_BLOCK rows */
  uint8_t* nullbuf;   /* result is_null, ORC_BLOCK rows */
  uint8_t* skipbuf;   /* skip vector handed to a child, ORC_BLOCK rows (allocated on first use) */
  const void* data;   /* evaluation result for the current block */
  const uint8_t* nulls;
} bnode;

/* Bound nodes of the cursors one thread creates while `tl_nodes` is set are remembered there, so that orc_cursor_free can
 * release them (the bench harness below creates thousands of cursors; the tests never free theirs). */
typedef struct { bnode** node; uint8_t* owns_bufs; int64_t n, cap; } node_list;
static __thread node_list* tl_nodes = NULL;
static void track_node(bnode* b, int owns_bufs) {
  node_list* l = tl_nodes;
  if (!l) return;
  if (l->n == l->cap) {
    l->cap = l->cap ? l->cap * 2 : 64;
    l->node = (bnode**)realloc(l->node, sizeof(bnode*) * (size_t)l->cap);
    l->owns_bufs = (uint8_t*)realloc(l->owns_bufs, (size_t)l->cap);
  }
  l->node[l->n] = b; l->owns_bufs[l->n] = (uint8_t)owns_bufs; ++l->n;
}

static bnode* bnode_new(int kind, int op, int dtype, int nullable, const char* name) {
  bnode* b = (bnode*)calloc(1, sizeof(bnode));
  b->kind = kind; b->op = op; b->dtype = dtype; b->nullable = nullable;
  snprintf(b->name, sizeof(b->name), "%s", name ? name : "");
  b->buf = calloc(ORC_BLOCK, 8);
  b->nullbuf = (uint8_t*)calloc(ORC_BLOCK, 1);
  track_node(b, 1);
  return b;
}

static uint64_t const_bits(int dtype, int64_t i64, double f64) {
  uint64_t b = 0;
  switch (dtype) {
    case T_INT32: case T_DATE: case T_STRING: { int32_t v = (int32_t)i64; uint32_t u; memcpy(&u, &v, 4); b = u; } break;
    case T_UINT32: b = (uint32_t)i64; break;
    case T_INT64: case T_DATETIME: case T_UINT64: memcpy(&b, &i64, 8); break;
    case T_FLOAT: { float v = (float)f64; uint32_t u; memcpy(&u, &v, 4); b = u; } break;
    case T_DOUBLE: memcpy(&b, &f64, 8); break;
    case T_BOOL: b = i64 != 0; break;
  }
  return b;
}

/* BoundConstExpression names "CONST_<TYPE>", NOT_NULLABLE
 * (expression/infrastructure/elementary_bound_const_expressions.h:36-41);
 * BoundNullExpression is named "NULL", NULLABLE (terminal_bound_expressions.cc:142) */
static bnode* make_const(int dtype, uint64_t bits) {
  char nm[64]; snprintf(nm, sizeof(nm), "CONST_%s", type_name(dtype));
  bnode* b = bnode_new(B_CONST, 0, dtype, 0, nm); b->bits = bits; return b;
}
static bnode* make_null(int dtype) { return bnode_new(B_NULLCONST, 0, dtype, 1, "NULL"); }

/* ---- vector primitives: the inner loops
 * (expression/vector/vector_primitives.h:99-105,393-412; functors operators.h:68-296) ---- */
#define LOOP2(TA, TB, TD, EXPR) { const TA* pa = (const TA*)A; const TB* pb = (const TB*)B; TD* pd = (TD*)D; \
  for (int64_t i = 0; i < n; ++i) { TA a = pa[i]; TB b = pb[i]; pd[i] = (TD)(EXPR); } }
#define LOOP1(TA, TD, EXPR) { const TA* pa = (const TA*)A; TD* pd = (TD*)D; \
  for (int64_t i = 0; i < n; ++i) { TA a = pa[i]; pd[i] = (TD)(EXPR); } }

static int arith_kind(int t) { /* 0:i32 1:u32 2:i64 3:u64 4:f32 5:f64 6:bool */
  switch (t) { case T_INT32: case T_DATE: case T_STRING: return 0; case T_UINT32: return 1; case T_INT64: case T_DATETIME: return 2;
               case T_UINT64: return 3; case T_FLOAT: return 4; case T_DOUBLE: return 5; case T_BOOL: return 6; }
  return -1;
}

/* D = A op B, both operands already of type t (after promotion) */
static int eval_binary(int op, int t, const void* A, const void* B, void* D, int64_t n) {
  const int k = arith_kind(t);
  switch (op) {
#define ARITH(OPID, EXPR) case OPID: switch (k) { \
      case 0: case 1: LOOP2(uint32_t, uint32_t, uint32_t, EXPR) return 1; \
      case 2: case 3: LOOP2(uint64_t, uint64_t, uint64_t, EXPR) return 1; \
      case 4: LOOP2(float, float, float, EXPR) return 1; \
      case 5: LOOP2(double, double, double, EXPR) return 1; } return 0;
    ARITH(OP_ADD, a + b)       /* integer + - * wrap (operators.h:68-81) */
    ARITH(OP_SUBTRACT, a - b)
    ARITH(OP_MULTIPLY, a * b)
    case OP_DIVIDE_QUIET: case OP_DIVIDE_NULLING: case OP_DIVIDE_SIGNALING:
      if (k == 5) { LOOP2(double, double, double, a / b) return 1; }
      if (k == 4) { LOOP2(float, float, float, a / b) return 1; }
      return 0;
    case OP_CPP_DIVIDE_NULLING: case OP_CPP_DIVIDE_SIGNALING:
      switch (k) {
        case 0: LOOP2(int32_t, int32_t, int32_t, (b == 0 ? 0 : (b == -1 ? (int32_t)(0u - (uint32_t)a) : a / b))) return 1;
        case 1: LOOP2(uint32_t, uint32_t, uint32_t, (b == 0 ? 0 : a / b)) return 1;
        case 2: LOOP2(int64_t, int64_t, int64_t, (b == 0 ? 0 : (b == -1 ? (int64_t)(0ull - (uint64_t)a) : a / b))) return 1;
        case 3: LOOP2(uint64_t, uint64_t, uint64_t, (b == 0 ? 0 : a / b)) return 1;
        case 4: LOOP2(float, float, float, a / b) return 1;
        case 5: LOOP2(double, double, double, a / b) return 1;
      } return 0;
/* operators.h:145-173: & | ^ and (~a) & b on the common integer type */
#define BITOP(OPID, EXPR) case OPID: switch (k) { \
      case 0: case 1: LOOP2(uint32_t, uint32_t, uint32_t, EXPR) return 1; \
      case 2: case 3: LOOP2(uint64_t, uint64_t, uint64_t, EXPR) return 1; } return 0;
    BITOP(OP_BITWISE_AND, a & b) BITOP(OP_BITWISE_OR, a | b) BITOP(OP_BITWISE_XOR, a ^ b) BITOP(OP_BITWISE_ANDNOT, (~a) & b)
    case OP_MODULUS_NULLING: case OP_MODULUS_SIGNALING:
      switch (k) {
        case 0: LOOP2(int32_t, int32_t, int32_t, (b == 0 || b == -1 ? 0 : a % b)) return 1;
        case 1: LOOP2(uint32_t, uint32_t, uint32_t, (b == 0 ? 0 : a % b)) return 1;
        case 2: LOOP2(int64_t, int64_t, int64_t, (b == 0 || b == -1 ? 0 : a % b)) return 1;
        case 3: LOOP2(uint64_t, uint64_t, uint64_t, (b == 0 ? 0 : a % b)) return 1;
      } return 0;
#define CMP(OPID, EXPR) case OPID: switch (k) { \
      case 0: LOOP2(int32_t, int32_t, uint8_t, EXPR) return 1; case 1: LOOP2(uint32_t, uint32_t, uint8_t, EXPR) return 1; \
      case 2: LOOP2(int64_t, int64_t, uint8_t, EXPR) return 1; case 3: LOOP2(uint64_t, uint64_t, uint8_t, EXPR) return 1; \
      case 4: LOOP2(float, float, uint8_t, EXPR) return 1; case 5: LOOP2(double, double, uint8_t, EXPR) return 1; \
      case 6: LOOP2(uint8_t, uint8_t, uint8_t, EXPR) return 1; } return 0;
    CMP(OP_LESS, a < b)
    CMP(OP_LESS_OR_EQUAL, a <= b)
    CMP(OP_EQUAL, a == b)
    CMP(OP_NOT_EQUAL, a != b)
    case OP_XOR: LOOP2(uint8_t, uint8_t, uint8_t, ((a != 0) != (b != 0))) return 1;
  }
  return 0;
}

/* comparisons of two DIFFERENT integer types are not cast: the functors are
 * value-correct for mixed signedness (operators.h:189-214,243-268) */
static int64_t load_signed(int t, const void* p, int64_t i, int* is_big_unsigned, uint64_t* u) {
  *is_big_unsigned = 0;
  switch (arith_kind(t)) {
    case 0: return ((const int32_t*)p)[i];
    case 1: return ((const uint32_t*)p)[i];
    case 2: return ((const int64_t*)p)[i];
    case 3: *u = ((const uint64_t*)p)[i]; if (*u > (uint64_t)INT64_MAX) { *is_big_unsigned = 1; return 0; } return (int64_t)*u;
  }
  return 0;
}
static void eval_mixed_int_compare(int op, int ta, const void* A, int tb, const void* B, uint8_t* D, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    int ba, bb; uint64_t ua = 0, ub = 0;
    int64_t a = load_signed(ta, A, i, &ba, &ua), b = load_signed(tb, B, i, &bb, &ub);
    int lt, eq;
    if (ba && bb) { lt = ua < ub; eq = ua == ub; }
    else if (ba) { lt = 0; eq = 0; }      /* a > INT64_MAX >= b */
    else if (bb) { lt = 1; eq = 0; }
    else { lt = a < b; eq = a == b; }
    D[i] = op == OP_LESS ? lt : op == OP_LESS_OR_EQUAL ? (lt || eq) : op == OP_EQUAL ? eq : !eq;
  }
}

static int eval_cast(int from, int to, const void* A, void* D, int64_t n) {
  /* OPERATOR_DATE_TO_DATETIME (templated/cast_bound_expression.cc:129-136): days * microseconds per day */
  if (from == T_DATE && to == T_DATETIME) { LOOP1(int32_t, int64_t, (int64_t)a * 86400000000LL) return 1; }
  const int kf = arith_kind(from), kt = arith_kind(to);
  if (kf < 0 || kt < 0) return 0;
#define CASTROW(TA) switch (kt) { \
    case 0: LOOP1(TA, int32_t, a) return 1; case 1: LOOP1(TA, uint32_t, a) return 1; \
    case 2: LOOP1(TA, int64_t, a) return 1; case 3: LOOP1(TA, uint64_t, a) return 1; \
    case 4: LOOP1(TA, float, a) return 1; case 5: LOOP1(TA, double, a) return 1; } return 0;
  switch (kf) {
    case 0: CASTROW(int32_t) case 1: CASTROW(uint32_t) case 2: CASTROW(int64_t)
    case 3: CASTROW(uint64_t) case 4: CASTROW(float) case 5: CASTROW(double)
  }
  return 0;
}

/* the libm family: expression/core/math_evaluators.h:92-204 (one libm call per row) */
static const char* libm1_name(int op) {
  switch (op) { case OP_EXP: return "EXP"; case OP_LN_QUIET: case OP_LN_NULLING: return "LN"; case OP_LOG10_QUIET: case OP_LOG10_NULLING: return "LOG10";
    case OP_LOG2_QUIET: case OP_LOG2_NULLING: return "LOG2"; case OP_SIN: return "SIN"; case OP_COS: return "COS"; case OP_TAN: return "TAN";
    case OP_ASIN: return "ASIN"; case OP_ACOS: return "ACOS"; case OP_ATAN: return "ATAN"; case OP_SINH: return "SINH"; case OP_COSH: return "COSH";
    case OP_TANH: return "TANH"; case OP_ASINH: return "ASINH"; case OP_ACOSH: return "ACOSH"; case OP_ATANH: return "ATANH"; }
  return "?";
}
static double libm1(int op, double a) {
  switch (op) { case OP_EXP: return exp(a); case OP_LN_QUIET: case OP_LN_NULLING: return log(a); case OP_LOG10_QUIET: case OP_LOG10_NULLING: return log10(a);
    case OP_LOG2_QUIET: case OP_LOG2_NULLING: return log2(a); case OP_SIN: return sin(a); case OP_COS: return cos(a); case OP_TAN: return tan(a);
    case OP_ASIN: return asin(a); case OP_ACOS: return acos(a); case OP_ATAN: return atan(a); case OP_SINH: return sinh(a); case OP_COSH: return cosh(a);
    case OP_TANH: return tanh(a); case OP_ASINH: return asinh(a); case OP_ACOSH: return acosh(a); case OP_ATANH: return atanh(a); }
  return 0;
}

/* ---- binding ---------------------------------------------------------------------- */
/* CommonTypeCalculator, expression/templated/bound_expression_factory.cc:67-107 */
static int common_type(int t1, int t2, orc_error* err) {
  if (t1 == t2) return t1;
  static const int tab[][3] = {
      {T_DOUBLE, T_INT32, T_DOUBLE}, {T_DOUBLE, T_INT64, T_DOUBLE}, {T_DOUBLE, T_UINT32, T_DOUBLE},
      {T_DOUBLE, T_UINT64, T_DOUBLE}, {T_DOUBLE, T_FLOAT, T_DOUBLE}, {T_FLOAT, T_INT32, T_FLOAT},
      {T_FLOAT, T_UINT32, T_FLOAT}, {T_FLOAT, T_UINT64, T_DOUBLE}, {T_FLOAT, T_INT64, T_DOUBLE},
      {T_INT64, T_INT32, T_INT64}, {T_INT64, T_UINT32, T_INT64}, {T_INT64, T_UINT64, T_INT64},
      {T_UINT64, T_INT32, T_INT64}, {T_UINT64, T_UINT32, T_UINT64}, {T_UINT32, T_INT32, T_INT64},
      {T_DATE, T_DATETIME, T_DATETIME}};
  for (size_t i = 0; i < sizeof(tab) / sizeof(tab[0]); ++i)
    if ((tab[i][0] == t1 && tab[i][1] == t2) || (tab[i][0] == t2 && tab[i][1] == t1)) return tab[i][2];
  set_err(err, RC_TYPE_MISMATCH, "Cannot reconcile types: %s and %s.", type_name(t1), type_name(t2));
  return -1;
}

static void eval_node(bnode* b, const orc_view* in, int64_t n, const uint8_t* skip, orc_error* err);
static int all_const(bnode* b) {
  for (int i = 0; i < b->nargs; ++i) if (b->args[i]->kind != B_CONST && b->args[i]->kind != B_NULLCONST) return 0;
  return b->nargs > 0;
}
/* InitBasicExpression: an expression with only constant children is evaluated once
 * (for one row) and replaced by the constant
 * (expression/infrastructure/basic_bound_expression.cc:57-92) */
static bnode* fold(bnode* b, orc_error* err) {
  if ((b->kind != B_OP && b->kind != B_CAST) || !all_const(b)) return b;
  orc_view dummy; memset(&dummy, 0, sizeof(dummy)); dummy.rows = 1;
  orc_error local; memset(&local, 0, sizeof(local));
  eval_node(b, &dummy, 1, NULL, &local);
  if (local.code) { if (err && !err->code) *err = local; return b; }
  if (b->nulls && b->nulls[0]) return make_null(b->dtype);
  uint64_t bits = 0; memcpy(&bits, b->data, (size_t)type_width(b->dtype));
  return make_const(b->dtype, bits);
}

/* BoundInternalCast (expression/templated/cast_bound_expression.cc:285-470):
 * result name "CAST_<FROM>_TO_<TO>(child)"; implicit downcasts and float->int are
 * bind errors (ERROR_ATTRIBUTE_TYPE_MISMATCH). */
static bnode* make_cast(bnode* child, int to, int is_implicit, orc_error* err) {
  const int from = child->dtype;
  if (from == to) return child;
  if (from == T_DATE && to == T_DATETIME) {
    char nm[256]; snprintf(nm, sizeof(nm), "CAST_DATE_TO_DATETIME(%s)", child->name);
    bnode* b = bnode_new(B_CAST, OP_CAST, to, child->nullable, nm);
    b->args[0] = child; b->nargs = 1;
    return fold(b, err);
  }
  if (!is_numeric(from) || !is_numeric(to)) { set_err(err, RC_TYPE_MISMATCH, "Cannot cast %s to %s.", type_name(from), type_name(to)); return child; }
  if (is_float(from) && is_integer(to)) { set_err(err, RC_TYPE_MISMATCH, "Cannot cast %s to %s (floating to integer).", type_name(from), type_name(to)); return child; }
  int down = ((from == T_INT64 || from == T_UINT64) && (to == T_INT32 || to == T_UINT32 || to == T_FLOAT)) || (from == T_DOUBLE && to == T_FLOAT);
  if (down && is_implicit) { set_err(err, RC_TYPE_MISMATCH, "Cannot cast %s to %s. Implicit downcasts are disallowed.", type_name(from), type_name(to)); return child; }
  char nm[256]; snprintf(nm, sizeof(nm), "CAST_%s_TO_%s(%s)", type_name(from), type_name(to), child->name);
  bnode* b = bnode_new(B_CAST, OP_CAST, to, child->nullable, nm);
  b->args[0] = child; b->nargs = 1;
  return fold(b, err);
}

/* FormatDescription strings, expression/vector/expression_traits.h:1204-1570 */
static void fmt_binary(char* out, size_t cap, int op, const char* l, const char* r) {
  const char* s = "?";
  switch (op) {
    case OP_ADD: s = "+"; break; case OP_SUBTRACT: s = "-"; break; case OP_MULTIPLY: s = "*"; break;
    case OP_DIVIDE_QUIET: case OP_DIVIDE_NULLING: case OP_DIVIDE_SIGNALING: s = "/."; break;
    case OP_CPP_DIVIDE_NULLING: case OP_CPP_DIVIDE_SIGNALING: s = "/"; break;
    case OP_MODULUS_NULLING: case OP_MODULUS_SIGNALING: s = "%"; break;
    case OP_EQUAL: s = "=="; break; case OP_NOT_EQUAL: s = "<>"; break; case OP_LESS: s = "<"; break;
    case OP_LESS_OR_EQUAL: s = "<="; break; case OP_AND: s = "AND"; break; case OP_OR: s = "OR"; break;
    case OP_AND_NOT: s = "!&&"; break; case OP_XOR: s = "XOR"; break;
    case OP_BITWISE_AND: s = "&"; break; case OP_BITWISE_OR: s = "|"; break; case OP_BITWISE_XOR: s = "^"; break;
    case OP_SHIFT_LEFT: s = "<<"; break; case OP_SHIFT_RIGHT: s = ">>"; break;
  }
  if (op == OP_BITWISE_ANDNOT) { snprintf(out, cap, "(~%s & %s)", l, r); return; }   /* expression_traits.h:1527-1536 */
  if (op == OP_IF_NULL) snprintf(out, cap, "IFNULL(%s, %s)", l, r);
  else snprintf(out, cap, "(%s %s %s)", l, s, r);
}

static bnode* make_op2(int op, int dtype, int nullable, bnode* l, bnode* r, orc_error* err) {
  char nm[256]; fmt_binary(nm, sizeof(nm), op, l->name, r->name);
  bnode* b = bnode_new(B_OP, op, dtype, nullable, nm);
  b->args[0] = l; b->args[1] = r; b->nargs = 2;
  return fold(b, err);
}

static bnode* bind_expr(const orc_expr* e, const orc_schema* s, bnode** multi, int* nmulti, orc_error* err);

static bnode* bind_single(const orc_expr* e, const orc_schema* s, orc_error* err) {
  bnode* m[ORC_MAX_COLS]; int nm = 0;
  bnode* b = bind_expr(e, s, m, &nm, err);
  if (err->code) return NULL;
  if (nm != 1) { set_err(err, RC_COUNT_MISMATCH, "expression argument must have exactly one attribute%s%s", "", ""); return NULL; }
  (void)b;
  return m[0];
}

/* GenerateComparison, expression/core/comparison_bound_expressions.cc:587-638 */
static bnode* bind_compare(int op, bnode* l, bnode* r, orc_error* err) {
  int lt = l->dtype, rt = r->dtype;
  if (lt != rt) {
    if (!is_numeric(lt) || !is_numeric(rt)) { set_err(err, RC_TYPE_MISMATCH, "Cannot compare expressions of different, non-numeric types%s%s", "", ""); return NULL; }
    if (lt == T_DOUBLE || rt == T_DOUBLE) { l = make_cast(l, T_DOUBLE, 1, err); r = make_cast(r, T_DOUBLE, 1, err); }
    else if (lt == T_FLOAT || rt == T_FLOAT) { l = make_cast(l, T_FLOAT, 0, err); r = make_cast(r, T_FLOAT, 0, err); }
  }
  if (err->code) return NULL;
  return make_op2(op, T_BOOL, l->nullable || r->nullable, l, r, err);
}

/* CASE arg0 WHEN arg2 THEN arg3 ... ELSE arg1: BoundCase, elementary_bound_expressions.cc:1297-1356 */
static bnode* bind_case(const orc_expr* e, const orc_schema* s, orc_error* err) {
  bnode* a[ORC_MAX_ARGS];
  const int n = e->nargs;
  if (n < 2) { set_err(err, RC_INVALID_ARGUMENT_VALUE, "Case expects at least 2 arguments (make sense from 4 arguments).%s%s", "", ""); return NULL; }
  if (n % 2 != 0) { set_err(err, RC_INVALID_ARGUMENT_VALUE, "Case expects odd number of arguments.%s%s", "", ""); return NULL; }
  for (int i = 0; i < n; ++i) { a[i] = bind_single(e->args[i], s, err); if (err->code) return NULL; }
  int test_type = a[0]->dtype, out_type = a[1]->dtype;
  for (int i = 2; i < n; ++i) {
    int* expected = (i % 2 == 0) ? &test_type : &out_type;
    if (a[i]->dtype != *expected) {
      if (!is_numeric(a[i]->dtype) || !is_numeric(*expected)) {
        set_err(err, RC_TYPE_MISMATCH, "Bind failed: Case: Cannot cast attribute (%s to %s)", type_name(a[i]->dtype), type_name(*expected)); return NULL;
      }
      *expected = common_type(a[i]->dtype, *expected, err); if (err->code) return NULL;
    }
  }
  char nm[256]; size_t len = (size_t)snprintf(nm, sizeof(nm), "CASE(");
  int nullable = 0;
  for (int i = 0; i < n; ++i) {
    a[i] = make_cast(a[i], i % 2 == 0 ? test_type : out_type, 1, err); if (err->code) return NULL;
    if (i % 2 == 1 && a[i]->nullable) nullable = 1;   /* DetermineNullability :773-780 */
    if (len < sizeof(nm)) len += (size_t)snprintf(nm + len, sizeof(nm) - len, "%s%s", i ? ", " : "", a[i]->name);
  }
  if (len < sizeof(nm)) snprintf(nm + len, sizeof(nm) - len, ")");
  bnode* b = bnode_new(B_OP, OP_CASE, out_type, nullable, nm);
  for (int i = 0; i < n; ++i) b->args[i] = a[i];
  b->nargs = n;
  return fold(b, err);
}

/* expr IN (value, ...): BoundInSet, comparison_bound_expressions.cc:759-813 (+ :642-700, :150-156) */
static bnode* bind_in(const orc_expr* e, const orc_schema* s, orc_error* err) {
  bnode* a[ORC_MAX_ARGS];
  const int n = e->nargs;
  if (n < 1) { set_err(err, RC_INVALID_ARGUMENT_VALUE, "IN needs a needle%s%s", "", ""); return NULL; }
  for (int i = 0; i < n; ++i) { a[i] = bind_single(e->args[i], s, err); if (err->code) return NULL; }
  int t = a[0]->dtype;
  for (int i = 1; i < n; ++i) { t = common_type(t, a[i]->dtype, err); if (err->code) return NULL; }
  int nullable = 0;
  for (int i = 0; i < n; ++i) { a[i] = make_cast(a[i], t, 1, err); if (err->code) return NULL; if (a[i]->nullable) nullable = 1; }
  char nm[256]; size_t len = (size_t)snprintf(nm, sizeof(nm), "%s IN (", a[0]->name);
  for (int i = 1; i < n; ++i) if (len < sizeof(nm)) len += (size_t)snprintf(nm + len, sizeof(nm) - len, "%s%s", i > 1 ? ", " : "", a[i]->name);
  if (len < sizeof(nm)) snprintf(nm + len, sizeof(nm) - len, ")");
  bnode* b = bnode_new(B_OP, OP_IN, T_BOOL, nullable, nm);
  for (int i = 0; i < n; ++i) b->args[i] = a[i];
  b->nargs = n;
  return fold(b, err);
}

static bnode* bind_op(const orc_expr* e, const orc_schema* s, orc_error* err) {
  if (e->op == OP_CASE) return bind_case(e, s, err);
  if (e->op == OP_IN) return bind_in(e, s, err);
  bnode* a[3] = {0, 0, 0};
  for (int i = 0; i < e->nargs && i < 3; ++i) { a[i] = bind_single(e->args[i], s, err); if (err->code) return NULL; }
  const int op = e->op;
  switch (op) {
    case OP_ADD: case OP_SUBTRACT: case OP_MULTIPLY: case OP_CPP_DIVIDE_NULLING: case OP_CPP_DIVIDE_SIGNALING:
    case OP_MODULUS_NULLING: case OP_MODULUS_SIGNALING: {
      /* CreateBinaryNumericExpression, bound_expression_factory.h:506-535 */
      int t = common_type(a[0]->dtype, a[1]->dtype, err);
      if (err->code) return NULL;
      int integer_only = op == OP_MODULUS_NULLING || op == OP_MODULUS_SIGNALING;
      if (!is_numeric(t) || (integer_only && !is_integer(t))) { set_err(err, RC_TYPE_MISMATCH, "Operator not defined for type %s%s", type_name(t), ""); return NULL; }
      bnode* l = make_cast(a[0], t, 1, err); bnode* r = make_cast(a[1], t, 1, err);
      if (err->code) return NULL;
      int can_null = op == OP_CPP_DIVIDE_NULLING || op == OP_MODULUS_NULLING;
      return make_op2(op, t, l->nullable || r->nullable || can_null, l, r, err);
    }
    case OP_BITWISE_AND: case OP_BITWISE_OR: case OP_BITWISE_XOR: case OP_BITWISE_ANDNOT: {
      /* CreateBinaryIntegerExpression, bound_expression_factory.h:520-535: common type, integers only */
      int t = common_type(a[0]->dtype, a[1]->dtype, err);
      if (err->code) return NULL;
      if (!is_integer(t)) { set_err(err, RC_TYPE_MISMATCH, "Operator not defined for type %s%s", type_name(t), ""); return NULL; }
      bnode* l = make_cast(a[0], t, 1, err); bnode* r = make_cast(a[1], t, 1, err);
      if (err->code) return NULL;
      return make_op2(op, t, l->nullable || r->nullable, l, r, err);
    }
    case OP_SHIFT_LEFT: case OP_SHIFT_RIGHT: {
      /* CreateShiftExpression, elementary_bound_expressions.cc:1446-1489: the result inherits the LEFT
       * type, the shift count is any integer type and is not promoted */
      if (!is_integer(a[0]->dtype) || !is_integer(a[1]->dtype)) { set_err(err, RC_TYPE_MISMATCH, "Shift needs integer arguments%s%s", "", ""); return NULL; }
      return make_op2(op, a[0]->dtype, a[0]->nullable || a[1]->nullable, a[0], a[1], err);
    }
    case OP_BITWISE_NOT: {
      /* CreateIntegerUnaryFactory, elementary_bound_expressions.cc:1433-1443,1490-1503 */
      if (!is_integer(a[0]->dtype)) { set_err(err, RC_TYPE_MISMATCH, "BITWISE NOT needs an integer argument%s%s", "", ""); return NULL; }
      char nm[256]; snprintf(nm, sizeof(nm), "(~%s)", a[0]->name);
      bnode* b = bnode_new(B_OP, op, a[0]->dtype, a[0]->nullable, nm); b->args[0] = a[0]; b->nargs = 1;
      return fold(b, err);
    }
    case OP_DIVIDE_QUIET: case OP_DIVIDE_NULLING: case OP_DIVIDE_SIGNALING: {
      /* always DOUBLE, arithmetic_bound_expressions.cc:47-72 */
      bnode* l = make_cast(a[0], T_DOUBLE, 1, err); bnode* r = make_cast(a[1], T_DOUBLE, 1, err);
      if (err->code) return NULL;
      return make_op2(op, T_DOUBLE, l->nullable || r->nullable || op == OP_DIVIDE_NULLING, l, r, err);
    }
    case OP_EQUAL: case OP_NOT_EQUAL: case OP_LESS: case OP_LESS_OR_EQUAL: return bind_compare(op, a[0], a[1], err);
    /* a > b is bound as Less(b, a): comparison_bound_expressions.cc:832-848 */
    case OP_GREATER: return bind_compare(OP_LESS, a[1], a[0], err);
    case OP_GREATER_OR_EQUAL: return bind_compare(OP_LESS_OR_EQUAL, a[1], a[0], err);
    case OP_AND: case OP_OR: case OP_AND_NOT: case OP_XOR:
      /* BoundBooleanBinary: no promotions, both BOOL (elementary_bound_expressions.cc:1122-1148) */
      if (a[0]->dtype != T_BOOL || a[1]->dtype != T_BOOL) { set_err(err, RC_TYPE_MISMATCH, "Expected BOOL in %s%s", a[0]->name, ""); return NULL; }
      return make_op2(op, T_BOOL, a[0]->nullable || a[1]->nullable, a[0], a[1], err);
    case OP_NOT: {
      if (a[0]->dtype != T_BOOL) { set_err(err, RC_TYPE_MISMATCH, "Expected BOOL in %s%s", a[0]->name, ""); return NULL; }
      char nm[256]; snprintf(nm, sizeof(nm), "(NOT %s)", a[0]->name);
      bnode* b = bnode_new(B_OP, op, T_BOOL, a[0]->nullable, nm); b->args[0] = a[0]; b->nargs = 1;
      return fold(b, err);
    }
    case OP_NEGATE: {
      int t = a[0]->dtype;
      if (!is_numeric(t)) { set_err(err, RC_TYPE_MISMATCH, "NEGATE needs a numeric argument%s%s", "", ""); return NULL; }
      int st = t == T_UINT32 ? T_INT32 : t == T_UINT64 ? T_INT64 : t;
      bnode* c = a[0];
      if (st != t) {  /* projecting cast to the signed type of the same width */
        char cn[256]; snprintf(cn, sizeof(cn), "CAST_%s_TO_%s(%s)", type_name(t), type_name(st), c->name);
        bnode* k = bnode_new(B_CAST, OP_CAST, st, c->nullable, cn); k->args[0] = c; k->nargs = 1; c = fold(k, err);
      }
      char nm[256]; snprintf(nm, sizeof(nm), "(-%s)", a[0]->name);   /* the reinterpreting factory adds no CAST to the name (arithmetic_expressions_test.cc:28) */
      bnode* b = bnode_new(B_OP, op, st, c->nullable, nm); b->args[0] = c; b->nargs = 1;
      return fold(b, err);
    }
    /* ---- exact math family (expression/core/math_bound_expressions.cc:150-170,318-456,473-486) ---- */
    case OP_ABS: {
      int t = a[0]->dtype;
      if (t == T_UINT32 || t == T_UINT64) return a[0];
      int ot = t == T_INT32 ? T_UINT32 : t == T_INT64 ? T_UINT64 : (t == T_FLOAT || t == T_DOUBLE) ? t : -1;
      if (ot < 0) { set_err(err, RC_TYPE_MISMATCH, "ABS is not defined for %s%s", type_name(t), ""); return NULL; }
      char nm[256]; snprintf(nm, sizeof(nm), "ABS(%s)", a[0]->name);
      bnode* b = bnode_new(B_OP, op, ot, a[0]->nullable, nm); b->args[0] = a[0]; b->nargs = 1;
      return fold(b, err);
    }
    case OP_ROUND: case OP_CEIL: case OP_FLOOR: case OP_TRUNC: case OP_CEIL_TO_INT: case OP_FLOOR_TO_INT: case OP_ROUND_TO_INT: {
      int t = a[0]->dtype;
      if (is_integer(t)) return a[0];
      if (!is_float(t)) { set_err(err, RC_TYPE_MISMATCH, "rounding is not defined for %s%s", type_name(t), ""); return NULL; }
      bnode* child = a[0];
      int ops[2] = {op, -1};
      if (op == OP_ROUND_TO_INT) { ops[0] = OP_ROUND; ops[1] = OP_CEIL_TO_INT; }   /* BoundRoundToInt :327-339 */
      for (int k = 0; k < 2 && ops[k] >= 0; ++k) {
        const int o = ops[k];
        const char* n = o == OP_ROUND ? "ROUND" : o == OP_CEIL ? "CEIL" : o == OP_FLOOR ? "FLOOR" : o == OP_TRUNC ? "TRUNC" : o == OP_CEIL_TO_INT ? "CEIL_TO_INT" : "FLOOR_TO_INT";
        const int ot = (o == OP_CEIL_TO_INT || o == OP_FLOOR_TO_INT) ? T_INT64 : child->dtype;
        char nm[256]; snprintf(nm, sizeof(nm), "%s(%s)", n, child->name);
        bnode* b = bnode_new(B_OP, o, ot, child->nullable, nm); b->args[0] = child; b->nargs = 1;
        child = fold(b, err);
      }
      return child;
    }
    case OP_ROUND_WITH_PRECISION: {
      /* BoundRoundWithPrecision, math_bound_expressions.cc:341-382: ROUND_WITH_MULTIPLIER(x, POW_QUIET(10.0, precision)) */
      if (!is_integer(a[1]->dtype)) { set_err(err, RC_TYPE_MISMATCH, "Precision has to be an integer; is: %s%s", type_name(a[1]->dtype), ""); return NULL; }
      bnode* x = make_cast(a[0], T_DOUBLE, 1, err); bnode* pr = make_cast(a[1], T_DOUBLE, 1, err); if (err->code) return NULL;
      double ten_v = 10.0; uint64_t ten_bits; memcpy(&ten_bits, &ten_v, 8);
      bnode* ten = make_const(T_DOUBLE, ten_bits);
      char pn[256]; snprintf(pn, sizeof(pn), "POW(%s, %s)", ten->name, pr->name);
      bnode* pw = bnode_new(B_OP, OP_POW_QUIET, T_DOUBLE, pr->nullable, pn); pw->args[0] = ten; pw->args[1] = pr; pw->nargs = 2;
      pw = fold(pw, err);
      char nm[256]; snprintf(nm, sizeof(nm), "ROUND_WITH_MULTIPLIER(%s, %s)", x->name, pw->name);
      bnode* b = bnode_new(B_OP, OP_ROUND_WITH_MULTIPLIER, T_DOUBLE, x->nullable || pw->nullable, nm); b->args[0] = x; b->args[1] = pw; b->nargs = 2;
      return fold(b, err);
    }
    case OP_POW_QUIET: case OP_POW_NULLING: case OP_POW_SIGNALING: case OP_ATAN2: {
      /* promoting binary expressions over DOUBLE (math_bound_expressions.cc:126-148,221-228) */
      bnode* l = make_cast(a[0], T_DOUBLE, 1, err); bnode* r = make_cast(a[1], T_DOUBLE, 1, err); if (err->code) return NULL;
      char nm[256]; snprintf(nm, sizeof(nm), "%s(%s, %s)", op == OP_ATAN2 ? "ATAN2" : "POW", l->name, r->name);
      bnode* b = bnode_new(B_OP, op, T_DOUBLE, l->nullable || r->nullable || op == OP_POW_NULLING, nm); b->args[0] = l; b->args[1] = r; b->nargs = 2;
      return op == OP_POW_SIGNALING ? b : fold(b, err);
    }
    case OP_SQRT_QUIET: case OP_SQRT_NULLING: case OP_SQRT_SIGNALING:
    case OP_IS_FINITE: case OP_IS_INF: case OP_IS_NAN: case OP_IS_NORMAL: {
      bnode* c = make_cast(a[0], T_DOUBLE, 1, err); if (err->code) return NULL;
      const int is_sqrt = op == OP_SQRT_QUIET || op == OP_SQRT_NULLING || op == OP_SQRT_SIGNALING;
      const char* n = is_sqrt ? "SQRT" : op == OP_IS_FINITE ? "IS_FINITE" : op == OP_IS_INF ? "IS_INF" : op == OP_IS_NAN ? "IS_NAN" : "IS_NORMAL";
      char nm[256]; snprintf(nm, sizeof(nm), "%s(%s)", n, c->name);
      bnode* b = bnode_new(B_OP, op, is_sqrt ? T_DOUBLE : T_BOOL, c->nullable || op == OP_SQRT_NULLING, nm); b->args[0] = c; b->nargs = 1;
      return op == OP_SQRT_SIGNALING ? b : fold(b, err);
    }
    case OP_EXP: case OP_LN_QUIET: case OP_LN_NULLING: case OP_LOG10_QUIET: case OP_LOG10_NULLING: case OP_LOG2_QUIET: case OP_LOG2_NULLING:
    case OP_SIN: case OP_COS: case OP_TAN: case OP_ASIN: case OP_ACOS: case OP_ATAN: case OP_SINH: case OP_COSH: case OP_TANH:
    case OP_ASINH: case OP_ACOSH: case OP_ATANH: {
      /* promoting unary expressions over DOUBLE (math_bound_expressions.cc:44-92,150-283) */
      bnode* c = make_cast(a[0], T_DOUBLE, 1, err); if (err->code) return NULL;
      const int nulling = op == OP_LN_NULLING || op == OP_LOG10_NULLING || op == OP_LOG2_NULLING;
      char nm[256]; snprintf(nm, sizeof(nm), "%s(%s)", libm1_name(op), c->name);
      bnode* b = bnode_new(B_OP, op, T_DOUBLE, c->nullable || nulling, nm); b->args[0] = c; b->nargs = 1;
      return fold(b, err);
    }
    case OP_IS_ODD: case OP_IS_EVEN: {
      if (!is_integer(a[0]->dtype)) { set_err(err, RC_TYPE_MISMATCH, "IS_ODD / IS_EVEN need an integer argument%s%s", "", ""); return NULL; }
      char nm[256]; snprintf(nm, sizeof(nm), "%s(%s)", op == OP_IS_ODD ? "IS_ODD" : "IS_EVEN", a[0]->name);
      bnode* b = bnode_new(B_OP, op, T_BOOL, a[0]->nullable, nm); b->args[0] = a[0]; b->nargs = 1;
      return fold(b, err);
    }
    case OP_IS_NULL: {
      /* non-nullable argument binds to ConstBool(false): elementary_bound_expressions.cc:1419-1424 */
      if (!a[0]->nullable) return make_const(T_BOOL, 0);
      char nm[256]; snprintf(nm, sizeof(nm), "ISNULL(%s)", a[0]->name);
      bnode* b = bnode_new(B_OP, op, T_BOOL, 0, nm); b->args[0] = a[0]; b->nargs = 1;
      return fold(b, err);
    }
    case OP_IF_NULL: {
      int t = common_type(a[0]->dtype, a[1]->dtype, err); if (err->code) return NULL;
      bnode* l = make_cast(a[0], t, 1, err); bnode* r = make_cast(a[1], t, 1, err); if (err->code) return NULL;
      if (!l->nullable) return l;
      char nm[256]; fmt_binary(nm, sizeof(nm), op, l->name, r->name);
      bnode* b = bnode_new(B_OP, op, t, r->nullable, nm); b->args[0] = l; b->args[1] = r; b->nargs = 2;
      return b;
    }
    case OP_IF: case OP_NULLING_IF: {
      /* BoundIfInternal, elementary_bound_expressions.cc:1084-1120 */
      if (a[0]->dtype != T_BOOL) { set_err(err, RC_TYPE_MISMATCH, "Expected BOOL in %s%s", a[0]->name, ""); return NULL; }
      int t = common_type(a[1]->dtype, a[2]->dtype, err); if (err->code) return NULL;
      bnode* x = make_cast(a[1], t, 1, err); bnode* y = make_cast(a[2], t, 1, err); if (err->code) return NULL;
      char nm[256]; snprintf(nm, sizeof(nm), "IF %s THEN %s ELSE %s", a[0]->name, x->name, y->name);
      /* plain IF: nullable iff THEN or OTHERWISE is (CreateIfSchema :1010-1025) */
      /* NullingIf: the condition's NULLs are viral as well (:1013-1016) */
      bnode* b = bnode_new(B_OP, op, t, x->nullable || y->nullable || (op == OP_NULLING_IF && a[0]->nullable), nm);
      b->args[0] = a[0]; b->args[1] = x; b->args[2] = y; b->nargs = 3;
      return b;
    }
  }
  set_err(err, RC_NOT_IMPLEMENTED, "operator outside the restated path%s%s", "", "");
  return NULL;
}

static bnode* bind_expr(const orc_expr* e, const orc_schema* s, bnode** multi, int* nmulti, orc_error* err) {
  bnode* b = NULL;
  switch (e->kind) {
    case E_NAMED: {
      /* BoundInputProjectionExpression: zero-copy column reference
       * (expression/core/projecting_bound_expressions.cc:54-82) */
      int pos = schema_lookup(s, e->name);
      if (pos < 0) { set_err(err, RC_ATTRIBUTE_MISSING, "No attribute '%s' in the schema%s", e->name, ""); return NULL; }
      b = bnode_new(B_INPUT, 0, s->a[pos].type, s->a[pos].nullable, s->a[pos].name); b->input_col = pos;
    } break;
    case E_AT:
      if (e->i64 < 0 || e->i64 >= s->n) { set_err(err, RC_COUNT_MISMATCH, "source schema has too few attributes%s%s", "", ""); return NULL; }
      b = bnode_new(B_INPUT, 0, s->a[e->i64].type, s->a[e->i64].nullable, s->a[e->i64].name); b->input_col = (int)e->i64;
      break;
    case E_CONST: b = make_const(e->dtype, const_bits(e->dtype, e->i64, e->f64)); break;
    case E_NULL: b = make_null(e->dtype); break;
    case E_ALIAS: {
      bnode* c = bind_single(e->args[0], s, err); if (err->code) return NULL;
      b = (bnode*)malloc(sizeof(bnode)); *b = *c;   /* same computation, new name */
      track_node(b, 0);
      snprintf(b->name, sizeof(b->name), "%s", e->name);
    } break;
    case E_COMPOUND:
      for (int i = 0; i < e->nargs; ++i) {
        bnode* m[ORC_MAX_COLS]; int nm = 0;
        bind_expr(e->args[i], s, m, &nm, err); if (err->code) return NULL;
        for (int j = 0; j < nm; ++j) {
          for (int q = 0; q < *nmulti; ++q)
            if (strcmp(multi[q]->name, m[j]->name) == 0) { set_err(err, RC_ATTRIBUTE_EXISTS, "Duplicate attribute name \"%s\" in result schema%s", m[j]->name, ""); return NULL; }
          multi[(*nmulti)++] = m[j];
        }
      }
      return NULL;
    case E_CAST: { bnode* c = bind_single(e->args[0], s, err); if (err->code) return NULL; b = make_cast(c, e->dtype, 0, err); } break;
    case E_OP: b = bind_op(e, s, err); break;
  }
  if (err->code || !b) return NULL;
  multi[(*nmulti)++] = b;
  return b;
}

/* ---- evaluation: BoundExpression::DoEvaluate over one <=1024-row view
 * (expression/templated/abstract_bound_expressions.h:129-147; NULL flow
 *  projecting_bound_expressions.cc:66-82) ---------------------------------------------- */
static void or_nulls(uint8_t* dst, const uint8_t* a, const uint8_t* b, int64_t n) {
  for (int64_t i = 0; i < n; ++i) dst[i] = (a ? a[i] : 0) | (b ? b[i] : 0);
}
static void fill_const(bnode* b, int64_t n) {
  const int w = type_width(b->dtype);
  for (int64_t i = 0; i < n; ++i) memcpy((char*)b->buf + i * w, &b->bits, (size_t)w);
}

/* `skip` (NULL = no row skipped): rows the parent does not need.  The reference hands every child a skip vector --
 * the branches of IF only see the rows that chose them (elementary_bound_expressions.cc:935-955), the right side of
 * AND / OR / AND_NOT only the rows the left side has not decided (:279-318), a WHEN / THEN of CASE only the rows not
 * yet written / matched (:640-690), and the children of every other operator share ONE vector that each ORs its NULLs
 * into before the next child runs (abstract_bound_expressions.h:129-147) -- and a failing operator never fails on a
 * skipped row (BinaryFailureChecker takes the skip vector as is_null).  Values are still computed on every row here
 * (they are unspecified on skipped rows in the reference); only the failure checks look at `skip`. */
static const uint8_t* child_skip(bnode* b, const uint8_t* skip, const uint8_t* extra, int negate_extra, int64_t n) {
  if (!extra && !negate_extra) return skip;
  if (!b->skipbuf) b->skipbuf = (uint8_t*)calloc(ORC_BLOCK, 1);
  for (int64_t i = 0; i < n; ++i) {
    const int e = extra ? extra[i] != 0 : 0;
    b->skipbuf[i] = (uint8_t)((skip ? skip[i] != 0 : 0) || (negate_extra ? !e : e));
  }
  return b->skipbuf;
}

static void eval_node(bnode* b, const orc_view* in, int64_t n, const uint8_t* skip, orc_error* err) {
  switch (b->kind) {
    case B_INPUT: b->data = in->c[b->input_col].data; b->nulls = in->c[b->input_col].is_null; return;
    case B_CONST: fill_const(b, n); b->data = b->buf; b->nulls = NULL; return;  /* PostInit pre-fill */
    case B_NULLCONST: memset(b->buf, 0, (size_t)n * 8); memset(b->nullbuf, 1, (size_t)n); b->data = b->buf; b->nulls = b->nullbuf; return;
    default: break;
  }
  if (b->kind == B_OP && (b->op == OP_IF || b->op == OP_NULLING_IF) && b->nargs == 3) {
    eval_node(b->args[0], in, n, skip, err); if (err->code) return;
    bnode* c = b->args[0];
    uint8_t* choice = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));      /* condition TRUE and not NULL */
    for (int64_t i = 0; i < n; ++i) choice[i] = (uint8_t)(((const uint8_t*)c->data)[i] != 0 && !(c->nulls && c->nulls[i]));
    eval_node(b->args[1], in, n, child_skip(b, skip, choice, 1, n), err);
    if (!err->code) {
      if (b->op == OP_NULLING_IF && c->nulls) for (int64_t i = 0; i < n; ++i) choice[i] = (uint8_t)(choice[i] || c->nulls[i]);   /* a NULL condition skips both */
      eval_node(b->args[2], in, n, child_skip(b, skip, choice, 0, n), err);
    }
    free(choice);
    if (err->code) return;
  } else if (b->kind == B_OP && (b->op == OP_AND || b->op == OP_OR || b->op == OP_AND_NOT) && b->nargs == 2) {
    eval_node(b->args[0], in, n, skip, err); if (err->code) return;
    bnode* l = b->args[0];
    uint8_t* decided = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));     /* left not NULL and TRUE (OR, AND_NOT) / FALSE (AND) */
    for (int64_t i = 0; i < n; ++i) {
      const int lv = ((const uint8_t*)l->data)[i] != 0, ln = l->nulls && l->nulls[i];
      decided[i] = (uint8_t)(!ln && (b->op == OP_AND ? !lv : lv));
    }
    eval_node(b->args[1], in, n, child_skip(b, skip, decided, 0, n), err);
    free(decided);
    if (err->code) return;
  } else if (b->kind == B_OP && b->op == OP_IF_NULL && b->nargs == 2) {
    /* the replacement is evaluated only where the first argument IS NULL (elementary_expressions_test.cc:377-394) */
    eval_node(b->args[0], in, n, skip, err); if (err->code) return;
    if (b->args[0]->nulls) eval_node(b->args[1], in, n, child_skip(b, skip, b->args[0]->nulls, 1, n), err);
    else { uint8_t* all = (uint8_t*)malloc((size_t)(n > 0 ? n : 1)); memset(all, 1, (size_t)(n > 0 ? n : 1)); eval_node(b->args[1], in, n, all, err); free(all); }
    if (err->code) return;
  } else if (b->kind == B_OP && b->op == OP_CASE) {
    /* CASE value, OTHERWISE, then (WHEN, THEN) pairs: a WHEN sees the rows no earlier WHEN matched, its THEN the rows it matched */
    eval_node(b->args[0], in, n, skip, err); if (err->code) return;
    bnode* cv = b->args[0]; const int tw = type_width(cv->dtype);
    uint8_t* written = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    uint8_t* match = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) written[i] = (uint8_t)((skip && skip[i]) || (cv->nulls && cv->nulls[i]));
    for (int a = 2; a + 1 < b->nargs && !err->code; a += 2) {
      bnode* wn = b->args[a];
      eval_node(wn, in, n, child_skip(b, NULL, written, 0, n), err); if (err->code) break;
      for (int64_t i = 0; i < n; ++i) {
        int eq = 0;
        if (!written[i] && !(wn->nulls && wn->nulls[i])) {
          if (cv->dtype == T_DOUBLE) eq = ((const double*)cv->data)[i] == ((const double*)wn->data)[i];
          else if (cv->dtype == T_FLOAT) eq = ((const float*)cv->data)[i] == ((const float*)wn->data)[i];
          else eq = memcmp((const char*)cv->data + i * tw, (const char*)wn->data + i * tw, (size_t)tw) == 0;
        }
        match[i] = (uint8_t)eq;
      }
      eval_node(b->args[a + 1], in, n, child_skip(b, NULL, match, 1, n), err);
      for (int64_t i = 0; i < n; ++i) written[i] = (uint8_t)(written[i] || match[i]);
    }
    if (!err->code) {   /* OTHERWISE: rows not written, plus the rows whose CASE value is NULL (but not the parent's skipped rows) */
      for (int64_t i = 0; i < n; ++i) written[i] = (uint8_t)((skip && skip[i]) || (written[i] && !(cv->nulls && cv->nulls[i])));
      eval_node(b->args[1], in, n, child_skip(b, NULL, written, 0, n), err);
    }
    free(written); free(match);
    if (err->code) return;
  } else {
    /* one shared skip vector: every child adds its NULLs before the next one is evaluated */
    const uint8_t* cur = skip; uint8_t* acc = NULL;
    for (int i = 0; i < b->nargs; ++i) {
      eval_node(b->args[i], in, n, cur, err);
      if (err->code) { free(acc); return; }
      if (b->args[i]->nulls && i + 1 < b->nargs) {
        if (!acc) { acc = (uint8_t*)malloc((size_t)(n > 0 ? n : 1)); for (int64_t r = 0; r < n; ++r) acc[r] = (uint8_t)(cur ? cur[r] != 0 : 0); }
        for (int64_t r = 0; r < n; ++r) acc[r] = (uint8_t)(acc[r] || b->args[i]->nulls[r]);
        cur = acc;
      }
    }
    free(acc);
  }
  bnode* x = b->args[0]; bnode* y = b->nargs > 1 ? b->args[1] : NULL;
  b->data = b->buf; b->nulls = NULL;
  if (b->kind == B_CAST) {
    const int same_width_int = is_integer(x->dtype) && is_integer(b->dtype) && type_width(x->dtype) == type_width(b->dtype);
    if (same_width_int) b->data = x->data;  /* projecting cast */
    else if (!eval_cast(x->dtype, b->dtype, x->data, b->buf, n)) set_err(err, RC_NOT_IMPLEMENTED, "cast%s%s", "", "");
    b->nulls = x->nulls;
    return;
  }
  switch (b->op) {
    case OP_AND: case OP_OR: case OP_AND_NOT: {
      /* three-valued logic, elementary_bound_expressions.cc:343-404: FALSE AND NULL = FALSE,
       * TRUE OR NULL = TRUE; AND_NOT(a,b) = (NOT a) AND b */
      const uint8_t* av = (const uint8_t*)x->data; const uint8_t* bv = (const uint8_t*)y->data;
      uint8_t* d = (uint8_t*)b->buf; int any_null = x->nulls || y->nulls;
      for (int64_t i = 0; i < n; ++i) {
        int A = av[i] != 0, B = bv[i] != 0, NA = x->nulls ? x->nulls[i] : 0, NB = y->nulls ? y->nulls[i] : 0;
        if (b->op == OP_AND_NOT) A = !A;
        if (b->op == OP_OR) { int dec = (!NA && A) || (!NB && B); b->nullbuf[i] = (NA || NB) && !dec; d[i] = dec || A || B; }
        else { int dec = (!NA && !A) || (!NB && !B); b->nullbuf[i] = (NA || NB) && !dec; d[i] = !dec && A && B; }
      }
      b->nulls = any_null ? b->nullbuf : NULL;
      return;
    }
    case OP_NOT: { const uint8_t* A = (const uint8_t*)x->data; uint8_t* D = (uint8_t*)b->buf; for (int64_t i = 0; i < n; ++i) D[i] = !A[i]; b->nulls = x->nulls; return; }
    case OP_NEGATE: {
      const void* A = x->data; void* D = b->buf;
      switch (arith_kind(b->dtype)) {
        case 0: case 1: LOOP1(uint32_t, uint32_t, 0u - a) break;
        case 2: case 3: LOOP1(uint64_t, uint64_t, 0ull - a) break;
        case 4: LOOP1(float, float, -a) break;
        case 5: LOOP1(double, double, -a) break;
      }
      b->nulls = x->nulls; return;
    }
    case OP_BITWISE_NOT: {
      const void* A = x->data; void* D = b->buf;
      if (type_width(b->dtype) == 4) LOOP1(uint32_t, uint32_t, ~a) else LOOP1(uint64_t, uint64_t, ~a)
      b->nulls = x->nulls; return;
    }
    case OP_SHIFT_LEFT: case OP_SHIFT_RIGHT: {
      /* operators.h:175-183: T1(a << b) / T1(a >> b); >> of a signed left operand is arithmetic (gcc / x86).
       * Counts outside [0, width) are undefined in the reference and are not exercised. */
      const int left = b->op == OP_SHIFT_LEFT, kx = arith_kind(x->dtype), ky = arith_kind(y->dtype);
      if (x->nulls || y->nulls) { or_nulls(b->nullbuf, x->nulls, y->nulls, n); b->nulls = b->nullbuf; }
      for (int64_t i = 0; i < n; ++i) {
        uint64_t c;
        switch (ky) { case 0: c = (uint64_t)(int64_t)((const int32_t*)y->data)[i]; break; case 1: c = ((const uint32_t*)y->data)[i]; break;
                      default: c = ((const uint64_t*)y->data)[i]; break; }
        switch (kx) {
          case 0: { int32_t v = ((const int32_t*)x->data)[i]; c &= 31; ((int32_t*)b->buf)[i] = left ? (int32_t)((uint32_t)v << c) : (v >> c); } break;
          case 1: { uint32_t v = ((const uint32_t*)x->data)[i]; c &= 31; ((uint32_t*)b->buf)[i] = left ? (v << c) : (v >> c); } break;
          case 2: { int64_t v = ((const int64_t*)x->data)[i]; c &= 63; ((int64_t*)b->buf)[i] = left ? (int64_t)((uint64_t)v << c) : (v >> c); } break;
          default: { uint64_t v = ((const uint64_t*)x->data)[i]; c &= 63; ((uint64_t*)b->buf)[i] = left ? (v << c) : (v >> c); } break;
        }
      }
      return;
    }
    case OP_IS_NULL: { uint8_t* D = (uint8_t*)b->buf; for (int64_t i = 0; i < n; ++i) D[i] = x->nulls ? x->nulls[i] : 0; return; }
    /* ---- exact math family: the functors of expression/core/math_evaluators.h:82-146,206-220 ---- */
    case OP_ABS: {
      const void* A = x->data; void* D = b->buf;
      switch (x->dtype) {
        case T_INT32: LOOP1(int32_t, uint32_t, (a < 0) ? 0u - (uint32_t)a : (uint32_t)a) break;
        case T_INT64: LOOP1(int64_t, uint64_t, (a < 0) ? 0ull - (uint64_t)a : (uint64_t)a) break;
        case T_FLOAT: LOOP1(float, float, (a < 0) ? -a : a) break;
        default: LOOP1(double, double, (a < 0) ? -a : a) break;
      }
      b->nulls = x->nulls; return;
    }
    case OP_ROUND: case OP_CEIL: case OP_FLOOR: case OP_TRUNC: {
      const void* A = x->data; void* D = b->buf; const int o = b->op;
      if (x->dtype == T_FLOAT) {
        if (o == OP_ROUND) LOOP1(float, float, roundf(a))
        else if (o == OP_CEIL) LOOP1(float, float, (float)ceil((double)a))
        else if (o == OP_FLOOR) LOOP1(float, float, (float)floor((double)a))
        else LOOP1(float, float, (float)trunc((double)a))
      } else {
        if (o == OP_ROUND) LOOP1(double, double, round(a))
        else if (o == OP_CEIL) LOOP1(double, double, ceil(a))
        else if (o == OP_FLOOR) LOOP1(double, double, floor(a))
        else LOOP1(double, double, trunc(a))
      }
      b->nulls = x->nulls; return;
    }
    case OP_CEIL_TO_INT: case OP_FLOOR_TO_INT: {
      const void* A = x->data; void* D = b->buf; const int up = b->op == OP_CEIL_TO_INT;
      if (x->dtype == T_FLOAT) { if (up) LOOP1(float, int64_t, (int64_t)ceil((double)a)) else LOOP1(float, int64_t, (int64_t)floor((double)a)) }
      else { if (up) LOOP1(double, int64_t, (int64_t)ceil(a)) else LOOP1(double, int64_t, (int64_t)floor(a)) }
      b->nulls = x->nulls; return;
    }
    case OP_EXP: case OP_LN_QUIET: case OP_LN_NULLING: case OP_LOG10_QUIET: case OP_LOG10_NULLING: case OP_LOG2_QUIET: case OP_LOG2_NULLING:
    case OP_SIN: case OP_COS: case OP_TAN: case OP_ASIN: case OP_ACOS: case OP_ATAN: case OP_SINH: case OP_COSH: case OP_TANH:
    case OP_ASINH: case OP_ACOSH: case OP_ATANH: {
      const double* A = (const double*)x->data; double* D = (double*)b->buf;
      for (int64_t i = 0; i < n; ++i) D[i] = libm1(b->op, A[i]);
      b->nulls = x->nulls;
      if (b->op == OP_LN_NULLING || b->op == OP_LOG10_NULLING || b->op == OP_LOG2_NULLING)   /* IsNonPositiveNuller, expression_traits.h:849-860 */
        for (int64_t i = 0; i < n; ++i) {
          if (!(A[i] <= 0)) continue;
          if (b->nulls != b->nullbuf) { if (x->nulls) memcpy(b->nullbuf, x->nulls, (size_t)n); else memset(b->nullbuf, 0, (size_t)n); b->nulls = b->nullbuf; }
          b->nullbuf[i] = 1;
        }
      return;
    }
    case OP_SQRT_QUIET: case OP_SQRT_NULLING: case OP_SQRT_SIGNALING: {
      const double* A = (const double*)x->data; double* D = (double*)b->buf;
      for (int64_t i = 0; i < n; ++i) D[i] = sqrt(A[i]);
      b->nulls = x->nulls;
      if (b->op != OP_SQRT_QUIET)   /* IsNegativeNuller / IsNegativeFailer, expression_traits.h:922-947 */
        for (int64_t i = 0; i < n; ++i) {
          if (!(A[i] < 0)) continue;
          const int already_null = x->nulls ? x->nulls[i] : 0;
          if (b->op == OP_SQRT_NULLING) {
            if (b->nulls != b->nullbuf) { if (x->nulls) memcpy(b->nullbuf, x->nulls, (size_t)n); else memset(b->nullbuf, 0, (size_t)n); b->nulls = b->nullbuf; }
            b->nullbuf[i] = 1;
          } else if (!already_null && !(skip && skip[i])) { set_err(err, RC_EVALUATION_ERROR, "Evaluation error: negative argument in %s%s", b->name, ""); return; }
        }
      return;
    }
    case OP_ROUND_WITH_MULTIPLIER: {   /* operators::RoundWithMultiplier, math_evaluators.h:117-121 */
      const double* A = (const double*)x->data; const double* B = (const double*)y->data; double* D = (double*)b->buf;
      for (int64_t i = 0; i < n; ++i) D[i] = round(A[i] * B[i]) / B[i];
      if (x->nulls || y->nulls) { or_nulls(b->nullbuf, x->nulls, y->nulls, n); b->nulls = b->nullbuf; } else b->nulls = NULL;
      return;
    }
    case OP_POW_QUIET: case OP_POW_NULLING: case OP_POW_SIGNALING: case OP_ATAN2: {
      const double* A = (const double*)x->data; const double* B = (const double*)y->data; double* D = (double*)b->buf;
      for (int64_t i = 0; i < n; ++i) D[i] = b->op == OP_ATAN2 ? atan2(A[i], B[i]) : pow(A[i], B[i]);
      if (x->nulls || y->nulls) { or_nulls(b->nullbuf, x->nulls, y->nulls, n); b->nulls = b->nullbuf; } else b->nulls = NULL;
      if (b->op == OP_POW_NULLING || b->op == OP_POW_SIGNALING)   /* FirstColumnNegativeAndSecondNonInteger, expression_traits.h:1329-1358 */
        for (int64_t i = 0; i < n; ++i) {
          if (!(A[i] < 0 && B[i] != trunc(B[i]))) continue;
          const int already_null = b->nulls ? b->nulls[i] : 0;
          if (b->op == OP_POW_NULLING) {
            if (!b->nulls) { memset(b->nullbuf, 0, (size_t)n); b->nulls = b->nullbuf; }
            b->nullbuf[i] = 1;
          } else if (!already_null && !(skip && skip[i])) { set_err(err, RC_EVALUATION_ERROR, "Evaluation error: negative base with a non-integer exponent in %s%s", b->name, ""); return; }
        }
      return;
    }
    case OP_IS_FINITE: case OP_IS_INF: case OP_IS_NAN: case OP_IS_NORMAL: {
      const double* A = (const double*)x->data; uint8_t* D = (uint8_t*)b->buf;
      for (int64_t i = 0; i < n; ++i)
        D[i] = (uint8_t)(b->op == OP_IS_FINITE ? isfinite(A[i]) != 0 : b->op == OP_IS_INF ? isinf(A[i]) != 0 : b->op == OP_IS_NAN ? isnan(A[i]) != 0 : isnormal(A[i]) != 0);
      b->nulls = x->nulls; return;
    }
    case OP_IS_ODD: case OP_IS_EVEN: {
      /* operators::IsOdd: arg % 2 (operators.h:101-120) */
      uint8_t* D = (uint8_t*)b->buf; const int w = type_width(x->dtype);
      for (int64_t i = 0; i < n; ++i) {
        int odd;
        if (x->dtype == T_INT32) odd = (((const int32_t*)x->data)[i] % 2) != 0;
        else if (x->dtype == T_UINT32) odd = (((const uint32_t*)x->data)[i] % 2) != 0;
        else if (x->dtype == T_INT64) odd = (((const int64_t*)x->data)[i] % 2) != 0;
        else odd = (((const uint64_t*)x->data)[i] % 2) != 0;
        D[i] = (uint8_t)(b->op == OP_IS_ODD ? odd : !odd);
      }
      (void)w; b->nulls = x->nulls; return;
    }
    case OP_CASE: {
      /* BoundCaseExpression::DoEvaluate (elementary_bound_expressions.cc:595-760): the first WHEN
       * that is non-NULL and equal to a non-NULL CASE value selects its THEN; otherwise (no match,
       * or CASE value NULL) the OTHERWISE argument; the result is NULL iff the selected one is.
       * (Values are computed on all rows; which rows each argument may FAIL on is settled by the skip
       * vectors eval_node handed them.) */
      const int w = type_width(b->dtype), tw = type_width(x->dtype);
      int any = 0;
      for (int a = 1; a < b->nargs; a += 2) if (b->args[a]->nulls) any = 1;
      for (int64_t i = 0; i < n; ++i) {
        bnode* src = b->args[1];
        if (!(x->nulls && x->nulls[i])) {
          for (int a = 2; a + 1 < b->nargs; a += 2) {
            bnode* wn = b->args[a];
            if (wn->nulls && wn->nulls[i]) continue;
            int eq;
            if (x->dtype == T_DOUBLE) eq = ((const double*)x->data)[i] == ((const double*)wn->data)[i];
            else if (x->dtype == T_FLOAT) eq = ((const float*)x->data)[i] == ((const float*)wn->data)[i];
            else eq = memcmp((const char*)x->data + i * tw, (const char*)wn->data + i * tw, (size_t)tw) == 0;
            if (eq) { src = b->args[a + 1]; break; }
          }
        }
        memcpy((char*)b->buf + i * w, (const char*)src->data + i * w, (size_t)w);
        b->nullbuf[i] = src->nulls ? src->nulls[i] : 0;
      }
      b->nulls = any ? b->nullbuf : NULL; return;
    }
    case OP_IN: {
      /* SQL IN (comparison_expressions.h:75-84): TRUE on a match; else NULL if the needle or any
       * list element is NULL; else FALSE */
      const int tw = type_width(x->dtype);
      uint8_t* D = (uint8_t*)b->buf; int any = 0;
      for (int a = 0; a < b->nargs; ++a) if (b->args[a]->nulls) any = 1;
      for (int64_t i = 0; i < n; ++i) {
        int found = 0, saw_null = x->nulls && x->nulls[i];
        for (int a = 1; a < b->nargs && !found; ++a) {
          bnode* h = b->args[a];
          if (h->nulls && h->nulls[i]) { saw_null = 1; continue; }
          if (x->nulls && x->nulls[i]) continue;
          if (x->dtype == T_DOUBLE) found = ((const double*)x->data)[i] == ((const double*)h->data)[i];
          else if (x->dtype == T_FLOAT) found = ((const float*)x->data)[i] == ((const float*)h->data)[i];
          else found = memcmp((const char*)x->data + i * tw, (const char*)h->data + i * tw, (size_t)tw) == 0;
        }
        D[i] = (uint8_t)found; b->nullbuf[i] = (uint8_t)(!found && saw_null);
      }
      b->nulls = any ? b->nullbuf : NULL; return;
    }
    case OP_IF_NULL: {
      const int w = type_width(b->dtype);
      for (int64_t i = 0; i < n; ++i) {
        int an = x->nulls ? x->nulls[i] : 0;
        memcpy((char*)b->buf + i * w, (const char*)(an ? y->data : x->data) + i * w, (size_t)w);
        b->nullbuf[i] = an && (y->nulls ? y->nulls[i] : 0);
      }
      b->nulls = y->nulls ? b->nullbuf : NULL; return;
    }
    case OP_IF: case OP_NULLING_IF: {
      bnode* z = b->args[2]; const int w = type_width(b->dtype); const uint8_t* cv = (const uint8_t*)x->data;
      const int nulling = b->op == OP_NULLING_IF;
      /* non-nulling IF: THEN iff the condition is non-NULL TRUE, otherwise OTHERWISE; NULLs come
       * from the chosen branch only (elementary_bound_expressions.cc:893-1008) */
      int any = y->nulls || z->nulls || (nulling && x->nulls);
      for (int64_t i = 0; i < n; ++i) {
        const int cnull = x->nulls && x->nulls[i];
        int c = cv[i] != 0 && !cnull; bnode* src = c ? y : z;
        memcpy((char*)b->buf + i * w, (const char*)src->data + i * w, (size_t)w);
        b->nullbuf[i] = (src->nulls ? src->nulls[i] : 0) || (nulling && cnull);   /* NullingIf: NULL condition, NULL result (:55-61) */
      }
      b->nulls = any ? b->nullbuf : NULL; return;
    }
  }
  /* standard binary operators: NULL in -> NULL out */
  if (x->nulls || y->nulls) { or_nulls(b->nullbuf, x->nulls, y->nulls, n); b->nulls = b->nullbuf; }
  const int is_cmp = b->op == OP_LESS || b->op == OP_LESS_OR_EQUAL || b->op == OP_EQUAL || b->op == OP_NOT_EQUAL;
  if (is_cmp && x->dtype != y->dtype) { eval_mixed_int_compare(b->op, x->dtype, x->data, y->dtype, y->data, (uint8_t*)b->buf, n); return; }
  if (!eval_binary(b->op, x->dtype, x->data, y->data, b->buf, n)) { set_err(err, RC_NOT_IMPLEMENTED, "operator/type not restated%s%s", "", ""); return; }
  /* failers / nullers (expression/vector/column_validity_checkers.h): zero divisor */
  const int nulling = b->op == OP_DIVIDE_NULLING || b->op == OP_CPP_DIVIDE_NULLING || b->op == OP_MODULUS_NULLING;
  const int signaling = b->op == OP_DIVIDE_SIGNALING || b->op == OP_CPP_DIVIDE_SIGNALING || b->op == OP_MODULUS_SIGNALING;
  if (nulling || signaling) {
    const int w = type_width(y->dtype); const int k = arith_kind(y->dtype);
    for (int64_t i = 0; i < n; ++i) {
      int zero;
      if (k == 5) zero = ((const double*)y->data)[i] == 0.0; else if (k == 4) zero = ((const float*)y->data)[i] == 0.0f;
      else if (w == 8) zero = ((const uint64_t*)y->data)[i] == 0; else zero = ((const uint32_t*)y->data)[i] == 0;
      if (!zero) continue;
      int already_null = b->nulls ? b->nulls[i] : 0;
      if (nulling) { if (!b->nulls) { memset(b->nullbuf, 0, (size_t)n); b->nulls = b->nullbuf; } b->nullbuf[i] = 1; }
      else if (!already_null && !(skip && skip[i])) { set_err(err, RC_EVALUATION_ERROR, "Evaluation error: division by zero in %s%s", b->name, ""); return; }
    }
  }
}

/* =========================== cursors ============================================ */
enum { C_SCAN = 1, C_COMPUTE, C_FILTER, C_PROJECT, C_SCALAR_AGG, C_GROUP_AGG, C_CLUSTERS, C_SORT, C_HASH_JOIN };
enum { P_ALL = 1, P_NAMED = 2, P_AT = 3, P_NAMED_AS = 4 };
typedef struct { int kind, position; char name[256], alias[256]; int source; } orc_proj;
typedef struct { int aggregation, distinct, output_type; char input[256], output[256]; } orc_agg;
typedef struct { char name[256]; int order; } orc_sortkey;

typedef struct orc_op {
  int kind; struct orc_op* child; orc_expr* expr;
  orc_proj projs[ORC_MAX_COLS]; int nproj;
  orc_agg aggs[ORC_MAX_COLS]; int nagg;
  orc_sortkey sortkeys[16]; int nsort;
  orc_schema scan_schema; orc_view scan_view;
  /* hash join (cursor/core/hash_join.h:37-56): projs = lhs key selector */
  struct orc_op* child2; orc_proj projs2[ORC_MAX_COLS]; int nproj2; orc_proj projs3[ORC_MAX_COLS]; int nproj3;
  int join_type;
  /* GroupAggregateOptions::max_unique_keys_in_result (cursor/core/aggregate.h:160-205): < 0 = no limit */
  int64_t max_unique_keys;
  /* BestEffortGroupAggregate (cursor/core/aggregate.h:230-250): > 0 = the result block holds this many groups; the cursor emits
   * what it aggregated when the next unseen key does not fit and starts anew at that row (aggregate_groups.cc:332-433) */
  int64_t best_effort_groups;
  int64_t best_effort_quota;    /* ... or GroupAggregateOptions::memory_quota in bytes (0 = none): groups = quota / bytes of a result row */
} orc_op;

orc_op* orc_op_new(int kind, orc_op* child, orc_expr* expr) {
  orc_op* o = (orc_op*)calloc(1, sizeof(orc_op)); o->kind = kind; o->child = child; o->expr = expr; o->max_unique_keys = -1; return o;
}
void orc_op_set_max_unique_keys(orc_op* o, int64_t limit) { o->max_unique_keys = limit; }
void orc_op_set_best_effort(orc_op* o, int64_t groups) { o->best_effort_groups = groups < 1 ? 1 : groups; }
/* BestEffortGroupAggregate under GroupAggregateOptions::set_memory_quota(bytes) (aggregate.h:170-175; 0 = no quota): the result block
 * holds quota / (bytes of one result row: every column's width + 1 byte of is_null where NULLABLE, block.cc:20-36) rows, at least 1
 * -- aggregate_groups_test.cc:601-626: INT32 key + INT32 SUM, both NULLABLE = 10 bytes, quota 20 => 2 groups per view */
void orc_op_set_best_effort_quota(orc_op* o, int64_t bytes) { o->best_effort_groups = INT64_MAX; o->best_effort_quota = bytes < 0 ? 0 : bytes; }
void orc_op_add_proj(orc_op* o, int kind, int position, const char* name, const char* alias) {
  orc_proj* p = &o->projs[o->nproj++]; p->kind = kind; p->position = position;
  snprintf(p->name, sizeof(p->name), "%s", name ? name : ""); snprintf(p->alias, sizeof(p->alias), "%s", alias ? alias : "");
}
/* join_type: 0 INNER, 1 LEFT_OUTER; bit 8 set = rhs keys declared UNIQUE (hash_join.h:37-56) */
void orc_op_set_join(orc_op* o, orc_op* rhs, int join_type) { o->child2 = rhs; o->join_type = join_type; }
void orc_op_add_proj_to(orc_op* o, int which, int source, int kind, int position, const char* name, const char* alias) {
  orc_proj* list = which == 2 ? o->projs2 : o->projs3; int* n = which == 2 ? &o->nproj2 : &o->nproj3;
  if (*n >= ORC_MAX_COLS) return;
  orc_proj* p = &list[(*n)++]; memset(p, 0, sizeof(*p));
  p->kind = kind; p->position = position; p->source = source;
  if (name) snprintf(p->name, sizeof(p->name), "%s", name);
  if (alias) snprintf(p->alias, sizeof(p->alias), "%s", alias);
}
void orc_op_add_agg(orc_op* o, int aggregation, int distinct, int output_type, const char* input, const char* output) {
  orc_agg* a = &o->aggs[o->nagg++]; a->aggregation = aggregation; a->distinct = distinct; a->output_type = output_type;
  snprintf(a->input, sizeof(a->input), "%s", input ? input : ""); snprintf(a->output, sizeof(a->output), "%s", output ? output : "");
}
void orc_op_add_sortkey(orc_op* o, const char* name, int order) {
  orc_sortkey* k = &o->sortkeys[o->nsort++]; snprintf(k->name, sizeof(k->name), "%s", name); k->order = order;
}
void orc_scan_add_column(orc_op* o, const char* name, int type, int nullable, const void* data, const uint8_t* is_null) {
  int i = o->scan_schema.n;
  schema_add(&o->scan_schema, name, type, nullable);
  o->scan_view.c[i].data = data; o->scan_view.c[i].is_null = is_null; o->scan_view.n = i + 1;
}
void orc_scan_set_rows(orc_op* o, int64_t rows) { o->scan_view.rows = rows; }

/* owned result block of a cursor */
typedef struct { int n; int64_t cap; void* data[ORC_MAX_COLS]; uint8_t* nulls[ORC_MAX_COLS]; int width[ORC_MAX_COLS]; } orc_block;
static void block_init(orc_block* b, const orc_schema* s, int64_t cap) {
  b->n = s->n; b->cap = cap;
  for (int i = 0; i < s->n; ++i) {
    b->width[i] = type_width(s->a[i].type);
    b->data[i] = calloc((size_t)(cap > 0 ? cap : 1), (size_t)(b->width[i] ? b->width[i] : 1));
    b->nulls[i] = (uint8_t*)calloc((size_t)(cap > 0 ? cap : 1), 1);
  }
}
static void block_grow(orc_block* b, int64_t cap) {
  if (cap <= b->cap) return;
  for (int i = 0; i < b->n; ++i) {
    b->data[i] = realloc(b->data[i], (size_t)cap * (size_t)b->width[i]);
    b->nulls[i] = (uint8_t*)realloc(b->nulls[i], (size_t)cap);
    memset((char*)b->data[i] + b->cap * b->width[i], 0, (size_t)(cap - b->cap) * (size_t)b->width[i]);
    memset(b->nulls[i] + b->cap, 0, (size_t)(cap - b->cap));
  }
  b->cap = cap;
}

/* distinct: only the first occurrence of every value of a group is aggregated -- DistinctAggregator keeps one hash set of
 * the values seen per result row (cursor/core/column_aggregator.cc:308-376); here ONE set of (result row, value bits) pairs */
typedef struct agg_col { int aggregation, in_pos, in_type, out_type; int distinct; uint64_t* dval; int64_t* drow; int64_t dcap, dcount; } agg_col;

typedef struct orc_cursor {
  int kind; struct orc_cursor* child; orc_schema schema; orc_error err;
  /* scan */ orc_view scan; int64_t pos;
  /* compute */ bnode* outs[ORC_MAX_COLS]; int nouts;
  /* filter / project / group keys */ bnode* pred; int proj_pos[ORC_MAX_COLS]; int nproj;
  int64_t* ids; int64_t nids, read_ptr; orc_view cur; int have_cur, eos;
  orc_block block;                  /* result block (filter / aggregates / sort) */
  /* aggregates */ agg_col aggs[ORC_MAX_COLS]; int nagg; int done; int64_t out_rows, emit_pos;
  int has_concat;   /* the specification holds a CONCAT: bound here (schema), evaluated in oracle.py (new STRINGs) */
  /* group */ int64_t* bucket_head; int64_t* chain_next; uint64_t* row_hash; int64_t nbuckets; int64_t max_unique_keys;
  int64_t best_effort_groups;   /* > 0: BestEffortGroupAggregate with a result block of this many rows (pending input: cur / read_ptr / eos) */
  /* sort */ int sort_pos[16], sort_order[16], nsort; int64_t* perm; void* table;
  orc_view outv;
  /* hash join: rhs fully materialised, lhs streamed (hash_join.cc: LookupIndex + HashJoinCursor) */
  struct orc_cursor* rhs; orc_block rhs_rows; int64_t rhs_n;
  int jl_pos[16], jr_pos[16], nkeys, join_type, join_unique;
  int64_t join_pending, join_served;   /* joined rows of the current input view: produced / already returned */
  int out_src[ORC_MAX_COLS], out_pos[ORC_MAX_COLS], nout_cols;
  int64_t* match;
} orc_cursor;

static int bind_projector(const orc_proj* p, int np, const orc_schema* in, int* pos, char names[][256], int* nout, orc_error* err) {
  int k = 0;
  for (int i = 0; i < np; ++i) {
    switch (p[i].kind) {
      case P_ALL: for (int c = 0; c < in->n; ++c) { pos[k] = c; snprintf(names[k], 256, "%s%s", p[i].alias, in->a[c].name); ++k; } break;  /* ProjectAllAttributes(prefix) */
      case P_NAMED: case P_NAMED_AS: {
        int c = schema_lookup(in, p[i].name);
        /* projector.cc:99-105 */
        if (c < 0) { set_err(err, RC_ATTRIBUTE_MISSING, "No attribute '%s' in the schema%s", p[i].name, ""); return 0; }
        pos[k] = c; snprintf(names[k], 256, "%s", p[i].kind == P_NAMED_AS ? p[i].alias : in->a[c].name); ++k;
      } break;
      case P_AT:
        /* projector.cc:185-190 */
        if (p[i].position < 0 || p[i].position >= in->n) { set_err(err, RC_COUNT_MISMATCH, "source schema has too few attributes%s%s", "", ""); return 0; }
        pos[k] = p[i].position; snprintf(names[k], 256, "%s", in->a[p[i].position].name); ++k; break;
    }
  }
  for (int i = 0; i < k; ++i) for (int j = i + 1; j < k; ++j)
    if (strcmp(names[i], names[j]) == 0) { set_err(err, RC_ATTRIBUTE_EXISTS, "Duplicate attribute name \"%s\" in result schema%s", names[i], ""); return 0; }
  *nout = k;
  return 1;
}

/* Aggregator::Init, cursor/core/aggregator.cc:116-186; output type rule :63-78;
 * supported matrix column_aggregator.cc:484-532 */
static int bind_aggs(orc_cursor* c, const orc_op* op, const orc_schema* in) {
  for (int i = 0; i < op->nagg; ++i) {
    const orc_agg* a = &op->aggs[i]; agg_col* g = &c->aggs[i];
    g->aggregation = a->aggregation;
    if (a->aggregation == A_COUNT && !a->distinct && a->input[0] == 0) g->in_pos = -1;
    else {
      g->in_pos = schema_lookup(in, a->input);
      if (g->in_pos < 0) { set_err(&c->err, RC_ATTRIBUTE_MISSING, "Incorrect aggregation specification. Aggregation input column does not exist: %s.%s", a->input, ""); return 0; }
    }
    g->in_type = g->in_pos >= 0 ? in->a[g->in_pos].type : T_UINT64;
    g->out_type = a->output_type >= 0 ? a->output_type : (a->aggregation == A_COUNT ? T_UINT64 : g->in_type);
    if (a->aggregation == A_CONCAT) {
      /* column_aggregator.cc:496-505: CONCAT -> STRING over every printable type.  Bound here (name, STRING, NULLABLE); its
       * values are new strings, which this restatement -- STRINGs are dictionary codes here -- cannot hold: oracle.py folds them */
      if (a->output_type >= 0 && a->output_type != T_STRING) { set_err(&c->err, RC_INVALID_ARGUMENT_TYPE, "Aggregation not supported for types %s and %s.", type_name(g->in_type), type_name(a->output_type)); return 0; }
      if (g->in_type == T_BINARY) { set_err(&c->err, RC_NOT_IMPLEMENTED, "CONCAT form not restated%s%s", "", ""); return 0; }
      /* (DISTINCT CONCAT: the DistinctAggregator in front of it, column_aggregator.cc:568-590 -- folded by oracle.py as well) */
      g->out_type = T_STRING; c->has_concat = 1;
      if (!schema_add(&c->schema, a->output, T_STRING, 1)) {
        set_err(&c->err, RC_ATTRIBUTE_EXISTS, "Incorrect aggregation specification. Aggregation output column name is non-unique: '%s'.%s", a->output, ""); return 0;
      }
      continue;
    }
    /* DISTINCT inside AggregateClusters: Aggregator::Create builds the same DistinctAggregator columns for every aggregating
     * cursor (cursor/core/aggregator.cc:88-101); AggregateClustersCursor::ProcessInput resets key set and aggregator together
     * and aggregates every cluster it emits inside ONE call (the unfinished last cluster is dropped and reprocessed,
     * aggregate_clusters.cc:436-520), so the (result row, value) sets below -- result row = the cluster's number -- see exactly
     * one cluster's rows.  The reference has no test vector for this combination: pinned only through its parts (the DISTINCT
     * vectors of GroupAggregate, the cluster vectors of AggregateClusters). */
    g->distinct = a->distinct;
    if (a->aggregation == A_COUNT) { if (!is_integer(g->out_type)) { set_err(&c->err, RC_INVALID_ARGUMENT_TYPE, "COUNT output must be integer%s%s", "", ""); return 0; } }
    else {
      int ok = (is_numeric(g->in_type) && is_numeric(g->out_type)) ||
               (g->in_type == g->out_type && a->aggregation != A_SUM && (g->in_type == T_BOOL || g->in_type == T_DATE || g->in_type == T_DATETIME || g->in_type == T_STRING));
      if (!ok) { set_err(&c->err, RC_INVALID_ARGUMENT_TYPE, "Aggregation not supported for types %s and %s.", type_name(g->in_type), type_name(g->out_type)); return 0; }
    }
    if (!schema_add(&c->schema, a->output, g->out_type, a->aggregation != A_COUNT)) {
      set_err(&c->err, RC_ATTRIBUTE_EXISTS, "Incorrect aggregation specification. Aggregation output column name is non-unique: '%s'.%s", a->output, ""); return 0;
    }
  }
  c->nagg = op->nagg;
  return 1;
}

orc_cursor* orc_create_cursor(const orc_op* op);

static orc_cursor* cursor_fail(orc_cursor* c) { return c; }

orc_cursor* orc_create_cursor(const orc_op* op) {
  orc_cursor* c = (orc_cursor*)calloc(1, sizeof(orc_cursor));
  c->kind = op->kind;
  if (op->child) {
    c->child = orc_create_cursor(op->child);
    if (c->child->err.code) { c->err = c->child->err; return cursor_fail(c); }
  }
  const orc_schema* in = c->child ? &c->child->schema : NULL;
  switch (op->kind) {
    case C_SCAN: c->schema = op->scan_schema; c->scan = op->scan_view; break;
    case C_COMPUTE: {
      /* ComputeOperation::CreateCursor binds the expression to the child schema
       * (cursor/core/compute.cc:69-79) */
      bnode* m[ORC_MAX_COLS]; int nm = 0;
      bind_expr(op->expr, in, m, &nm, &c->err);
      if (c->err.code) return cursor_fail(c);
      for (int i = 0; i < nm; ++i) {
        c->outs[i] = m[i];
        if (!schema_add(&c->schema, m[i]->name, m[i]->dtype, m[i]->nullable)) { set_err(&c->err, RC_ATTRIBUTE_EXISTS, "Duplicate attribute name \"%s\" in result schema%s", m[i]->name, ""); return cursor_fail(c); }
      }
      c->nouts = nm;
    } break;
    case C_PROJECT: case C_FILTER: {
      if (op->kind == C_FILTER) {
        bnode* m[ORC_MAX_COLS]; int nm = 0;
        bind_expr(op->expr, in, m, &nm, &c->err);
        if (c->err.code) return cursor_fail(c);
        /* filter.cc:84-92 */
        if (nm != 1) { set_err(&c->err, RC_COUNT_MISMATCH, "Predicate has to return exactly one column of type BOOL%s%s", "", ""); return cursor_fail(c); }
        if (m[0]->dtype != T_BOOL) { set_err(&c->err, RC_TYPE_MISMATCH, "Predicate has to return exactly one column of type BOOL%s%s", "", ""); return cursor_fail(c); }
        c->pred = m[0];
      }
      char names[ORC_MAX_COLS][256];
      if (!bind_projector(op->projs, op->nproj, in, c->proj_pos, names, &c->nproj, &c->err)) return cursor_fail(c);
      for (int i = 0; i < c->nproj; ++i) schema_add(&c->schema, names[i], in->a[c->proj_pos[i]].type, in->a[c->proj_pos[i]].nullable);
      if (op->kind == C_FILTER) { block_init(&c->block, &c->schema, ORC_BLOCK); c->ids = (int64_t*)malloc(sizeof(int64_t) * ORC_BLOCK); }
    } break;
    case C_SCALAR_AGG:
      if (!bind_aggs(c, op, in)) return cursor_fail(c);
      block_init(&c->block, &c->schema, 1);
      break;
    case C_GROUP_AGG: case C_CLUSTERS: {
      char names[ORC_MAX_COLS][256];
      if (!bind_projector(op->projs, op->nproj, in, c->proj_pos, names, &c->nproj, &c->err)) return cursor_fail(c);
      for (int i = 0; i < c->nproj; ++i) schema_add(&c->schema, names[i], in->a[c->proj_pos[i]].type, in->a[c->proj_pos[i]].nullable);
      if (!bind_aggs(c, op, in)) return cursor_fail(c);
      c->max_unique_keys = op->kind == C_GROUP_AGG ? op->max_unique_keys : -1;
      c->best_effort_groups = op->kind == C_GROUP_AGG ? op->best_effort_groups : 0;
      if (c->best_effort_groups > 0 && op->best_effort_quota > 0) {
        int64_t row_bytes = 0;
        for (int i = 0; i < c->schema.n; ++i) row_bytes += type_width(c->schema.a[i].type) + (c->schema.a[i].nullable ? 1 : 0);
        c->best_effort_groups = op->best_effort_quota / (row_bytes > 0 ? row_bytes : 1);
        if (c->best_effort_groups < 1) c->best_effort_groups = 1;
      }
      block_init(&c->block, &c->schema, 16);  /* kDefaultResultEstimatedGroupCount, aggregate.h:162 */
    } break;
    case C_HASH_JOIN: {
      c->rhs = orc_create_cursor(op->child2);
      if (c->rhs->err.code) { c->err = c->rhs->err; return cursor_fail(c); }
      const orc_schema* rs = &c->rhs->schema;
      char names[ORC_MAX_COLS][256]; int nl = 0, nr = 0;
      if (!bind_projector(op->projs, op->nproj, in, c->jl_pos, names, &nl, &c->err)) return cursor_fail(c);
      if (!bind_projector(op->projs2, op->nproj2, rs, c->jr_pos, names, &nr, &c->err)) return cursor_fail(c);
      if (nl != nr || nl == 0) { set_err(&c->err, RC_COUNT_MISMATCH, "hash join key selectors must pick the same number of columns%s%s", "", ""); return cursor_fail(c); }
      for (int k = 0; k < nl; ++k)
        if (in->a[c->jl_pos[k]].type != rs->a[c->jr_pos[k]].type) { set_err(&c->err, RC_TYPE_MISMATCH, "hash join key types differ%s%s", "", ""); return cursor_fail(c); }
      c->nkeys = nl; c->join_type = op->join_type & 0xFF; c->join_unique = (op->join_type >> 8) & 1;
      /* BoundMultiSourceProjector: entries in order, each bound against its source schema */
      for (int q = 0; q < op->nproj3; ++q) {
        const orc_schema* src = op->projs3[q].source == 0 ? in : rs;
        int pos[ORC_MAX_COLS], n1 = 0; char nm[ORC_MAX_COLS][256];
        if (!bind_projector(&op->projs3[q], 1, src, pos, nm, &n1, &c->err)) return cursor_fail(c);
        for (int i = 0; i < n1; ++i) {
          const int nullable = src->a[pos[i]].nullable || (op->projs3[q].source == 1 && (op->join_type & 0xFF) == 1);  /* LEFT_OUTER: rhs columns nullable */
          if (!schema_add(&c->schema, nm[i], src->a[pos[i]].type, nullable)) { set_err(&c->err, RC_ATTRIBUTE_EXISTS, "Duplicate attribute name \"%s\" in result schema%s", nm[i], ""); return cursor_fail(c); }
          c->out_src[c->nout_cols] = op->projs3[q].source; c->out_pos[c->nout_cols] = pos[i]; ++c->nout_cols;
        }
      }
      block_init(&c->block, &c->schema, ORC_BLOCK);
      c->match = (int64_t*)malloc(sizeof(int64_t) * ORC_BLOCK);
      c->rhs_n = -1;
    } break;
    case C_SORT: {
      for (int i = 0; i < op->nsort; ++i) {
        int p = schema_lookup(in, op->sortkeys[i].name);
        if (p < 0) { set_err(&c->err, RC_ATTRIBUTE_MISSING, "No attribute '%s' in the schema%s", op->sortkeys[i].name, ""); return cursor_fail(c); }
        c->sort_pos[i] = p; c->sort_order[i] = op->sortkeys[i].order;
      }
      c->nsort = op->nsort;
      char names[ORC_MAX_COLS][256];
      if (!bind_projector(op->projs, op->nproj, in, c->proj_pos, names, &c->nproj, &c->err)) return cursor_fail(c);
      for (int i = 0; i < c->nproj; ++i) schema_add(&c->schema, names[i], in->a[c->proj_pos[i]].type, in->a[c->proj_pos[i]].nullable);
    } break;
  }
  return c;
}

int orc_cursor_error(const orc_cursor* c, char* buf, int cap) { if (buf && cap > 0) snprintf(buf, (size_t)cap, "%s", c->err.msg); return c->err.code; }
int orc_cursor_ncols(const orc_cursor* c) { return c->schema.n; }
const char* orc_cursor_col_name(const orc_cursor* c, int i) { return c->schema.a[i].name; }
int orc_cursor_col_type(const orc_cursor* c, int i) { return c->schema.a[i].type; }
int orc_cursor_col_nullable(const orc_cursor* c, int i) { return c->schema.a[i].nullable; }

/* returns 1 = data, 0 = end of stream, -1 = failure */
static int cursor_next(orc_cursor* c, int64_t max_rows, orc_view* out);

/* ---- aggregation operators: AssignmentOperator / AggregationOperator
 * (base/infrastructure/aggregation_operators.h:57-71,173-228); first non-NULL value is
 * ASSIGNED (clears is_null), later ones AGGREGATED; NULL inputs skipped
 * (cursor/core/column_aggregator.cc:108-124,154-166); COUNT :213-226 ------------------ */
static double load_as_double(int t, const void* p, int64_t i) {
  switch (arith_kind(t)) {
    case 0: return ((const int32_t*)p)[i]; case 1: return ((const uint32_t*)p)[i];
    case 2: return (double)((const int64_t*)p)[i]; case 3: return (double)((const uint64_t*)p)[i];
    case 4: return ((const float*)p)[i]; case 5: return ((const double*)p)[i]; case 6: return ((const uint8_t*)p)[i];
  }
  return 0;
}
static int64_t load_as_i64(int t, const void* p, int64_t i) {
  switch (arith_kind(t)) {
    case 0: return ((const int32_t*)p)[i]; case 1: return ((const uint32_t*)p)[i];
    case 2: return ((const int64_t*)p)[i]; case 3: return (int64_t)((const uint64_t*)p)[i];
    case 4: return (int64_t)((const float*)p)[i]; case 5: return (int64_t)((const double*)p)[i]; case 6: return ((const uint8_t*)p)[i];
  }
  return 0;
}

/* operators::Less across two column types (supersonic/base/infrastructure/operators.h:225-278): `a < b` after the usual
 * arithmetic conversions, except that signed-vs-unsigned integers compare by value.  MIN / MAX compare every value IN ITS
 * OWN TYPE with the running result and store the value cast to the result type (aggregation_operators.h:187-228:
 * ThreeWayCompare<InputType, OutputType>), so a value the result type cannot hold makes the reference's result depend on
 * the row order -- the sequential fold below is the reference's. */
static int less_typed(int lt, const void* lp, int64_t li, int rt, const void* rp, int64_t ri) {
  const int lk = arith_kind(lt), rk = arith_kind(rt);
  if (lk == 4 || lk == 5 || rk == 4 || rk == 5) return load_as_double(lt, lp, li) < load_as_double(rt, rp, ri);   /* (float -> double is exact) */
  const __int128 a = lk == 3 ? (__int128)((const uint64_t*)lp)[li] : (__int128)load_as_i64(lt, lp, li);
  const __int128 b = rk == 3 ? (__int128)((const uint64_t*)rp)[ri] : (__int128)load_as_i64(rt, rp, ri);
  return a < b;
}

/* DISTINCT: 1 if (result row, value) was seen before, else remembers it (NULL inputs never get here: "NULL values do
 * not count as distinct", column_aggregator.cc:326-329) */
static int distinct_seen(agg_col* g, int64_t row, const void* in, int64_t i) {
  const int w = type_width(g->in_type);
  uint64_t bits = 0; memcpy(&bits, (const char*)in + i * w, (size_t)w);
  if (g->in_type == T_DOUBLE && bits == 0x8000000000000000ull) bits = 0;      /* -0.0 == +0.0 */
  if (g->in_type == T_FLOAT && bits == 0x80000000ull) bits = 0;
  if ((g->dcount + 1) * 2 > g->dcap) {
    const int64_t ncap = g->dcap ? g->dcap * 2 : 1024;
    uint64_t* nv = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)ncap); int64_t* nr = (int64_t*)malloc(sizeof(int64_t) * (size_t)ncap);
    for (int64_t q = 0; q < ncap; ++q) nr[q] = -1;
    for (int64_t q = 0; q < g->dcap; ++q) if (g->drow[q] >= 0) {
      uint64_t h = (g->dval[q] * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)g->drow[q] * 0xC2B2AE3D27D4EB4Full); int64_t s2 = (int64_t)((h ^ (h >> 29)) & (uint64_t)(ncap - 1));
      while (nr[s2] >= 0) s2 = (s2 + 1) & (ncap - 1);
      nr[s2] = g->drow[q]; nv[s2] = g->dval[q];
    }
    free(g->dval); free(g->drow); g->dval = nv; g->drow = nr; g->dcap = ncap;
  }
  uint64_t h = (bits * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)row * 0xC2B2AE3D27D4EB4Full); int64_t s2 = (int64_t)((h ^ (h >> 29)) & (uint64_t)(g->dcap - 1));
  while (g->drow[s2] >= 0) { if (g->drow[s2] == row && g->dval[s2] == bits) return 1; s2 = (s2 + 1) & (g->dcap - 1); }
  g->drow[s2] = row; g->dval[s2] = bits; g->dcount++;
  return 0;
}

/* one aggregate column over one view: acc[map[i]] op= v[i] */
static void update_aggregation(agg_col* g, const orc_view* v, const int64_t* map, void* res, uint8_t* res_null) {
  const int64_t n = v->rows;
  if (g->aggregation == A_COUNT) {
    const uint8_t* nl = g->in_pos >= 0 ? v->c[g->in_pos].is_null : NULL;
    for (int64_t i = 0; i < n; ++i) {
      if (nl && nl[i]) continue;
      if (g->distinct && distinct_seen(g, map[i], v->c[g->in_pos].data, i)) continue;
      if (type_width(g->out_type) == 8) ((uint64_t*)res)[map[i]] += 1; else ((uint32_t*)res)[map[i]] += 1;
    }
    return;
  }
  const void* in = v->c[g->in_pos].data; const uint8_t* nl = v->c[g->in_pos].is_null;
  const int ko = arith_kind(g->out_type);
  for (int64_t i = 0; i < n; ++i) {
    if (nl && nl[i]) continue;
    const int64_t r = map[i];
    if (g->aggregation == A_SUM_RESIDUAL) { res_null[r] = 0; ((double*)res)[r] = 0.0; continue; }
    if (g->distinct && distinct_seen(g, r, in, i)) continue;
    const int first = res_null[r];
    if (first) res_null[r] = 0;
#define AGG_TYPED(TO, LOADER) { TO val = (TO)LOADER(g->in_type, in, i); TO* acc = (TO*)res + r; \
      if (first) *acc = val; \
      else switch (g->aggregation) { \
        case A_SUM: *acc += val; break; \
        case A_MIN: if (less_typed(g->in_type, in, i, g->out_type, res, r)) *acc = val; break;  /* "val < result" replaces; NaN never does */ \
        case A_MAX: if (less_typed(g->out_type, res, r, g->in_type, in, i)) *acc = val; break; \
        case A_FIRST: break; \
        case A_LAST: *acc = val; break; } }
    if (g->aggregation == A_SUM && ko <= 3 && (arith_kind(g->in_type) == 4 || arith_kind(g->in_type) == 5)) {
      /* SUM of a floating input into an integer result (AddAggregationWithDefinedOutputType; column_aggregator.cc:484-532 lists
       * the pair): AggregationOperator<SUM>, aggregation_operators.h:173-185, is `*result += val` -- C++ adds in the floating
       * type (FLOAT stays float) and converts back to the integer after EVERY row; the first value is assigned
       * (column_aggregator.cc:154-166).  Order-dependent: this loop IS the reference's order.  Floating -> integer goes
       * through int64_t like every other such store here (in range it is the plain C conversion). */
#define SEQ_SUM(TO, F) { TO* acc = (TO*)res + r; const F v = ((const F*)in)[i]; \
        if (first) *acc = (TO)(int64_t)v; else *acc = (TO)(int64_t)((F)*acc + v); }
      if (arith_kind(g->in_type) == 4) switch (ko) {
        case 0: SEQ_SUM(int32_t, float) break; case 1: SEQ_SUM(uint32_t, float) break;
        case 2: SEQ_SUM(int64_t, float) break; default: SEQ_SUM(uint64_t, float) break; }
      else switch (ko) {
        case 0: SEQ_SUM(int32_t, double) break; case 1: SEQ_SUM(uint32_t, double) break;
        case 2: SEQ_SUM(int64_t, double) break; default: SEQ_SUM(uint64_t, double) break; }
#undef SEQ_SUM
      continue;
    }
    switch (ko) {
      case 0: AGG_TYPED(int32_t, load_as_i64) break;
      case 1: AGG_TYPED(uint32_t, load_as_i64) break;
      case 2: AGG_TYPED(int64_t, load_as_i64) break;
      case 3: AGG_TYPED(uint64_t, load_as_i64) break;
      case 4: AGG_TYPED(float, load_as_double) break;
      case 5: AGG_TYPED(double, load_as_double) break;
      case 6: AGG_TYPED(uint8_t, load_as_i64) break;
    }
  }
}

static void reset_agg_rows(orc_cursor* c, int key_cols, int64_t from, int64_t to) {
  for (int j = 0; j < c->nagg; ++j) {
    const int col = key_cols + j;
    /* COUNT starts at 0 and is NOT NULL; the others start NULL (column_aggregator.cc:240-252,170-175) */
    memset((char*)c->block.data[col] + from * c->block.width[col], 0, (size_t)(to - from) * (size_t)c->block.width[col]);
    memset(c->block.nulls[col] + from, c->aggs[j].aggregation == A_COUNT ? 0 : 1, (size_t)(to - from));
  }
}

/* ---- RowHashSet semantics (cursor/infrastructure/row_hash_set.cc:458-517): key -> dense
 * group id in insertion order; NULL keys equal each other (:81-93); chained buckets, 75%
 * load, power-of-two growth (:305-319,375). ------------------------------------------- */
static uint64_t mix64(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return k; }
static uint64_t hash_row(const orc_cursor* c, const orc_view* v, int64_t i) {
  uint64_t h = 0x9e3779b97f4a7c15ull;
  for (int k = 0; k < c->nproj; ++k) {
    const orc_col* col = &v->c[c->proj_pos[k]]; const int w = type_width(c->child->schema.a[c->proj_pos[k]].type);
    uint64_t x = 0;
    if (col->is_null && col->is_null[i]) x = 0xdeadbeefcafef00dull; else memcpy(&x, (const char*)col->data + i * w, (size_t)w);
    h = mix64(h ^ x) + 0x9e3779b97f4a7c15ull * (uint64_t)(k + 1);
  }
  return h;
}
static int key_equal(const orc_cursor* c, const orc_view* v, int64_t i, int64_t group) {
  for (int k = 0; k < c->nproj; ++k) {
    const orc_col* col = &v->c[c->proj_pos[k]]; const int w = c->block.width[k];
    const int an = col->is_null ? col->is_null[i] : 0, bn = c->block.nulls[k][group];
    if (an != bn) return 0;
    if (an) continue;
    if (memcmp((const char*)col->data + i * w, (const char*)c->block.data[k] + group * w, (size_t)w) != 0) return 0;
  }
  return 1;
}
static void group_rehash(orc_cursor* c, int64_t nb) {
  free(c->bucket_head); c->bucket_head = (int64_t*)malloc(sizeof(int64_t) * (size_t)nb); c->nbuckets = nb;
  for (int64_t i = 0; i < nb; ++i) c->bucket_head[i] = -1;
  for (int64_t g = 0; g < c->out_rows; ++g) { int64_t b = (int64_t)(c->row_hash[g] & (uint64_t)(nb - 1)); c->chain_next[g] = c->bucket_head[b]; c->bucket_head[b] = g; }
}
static int64_t group_insert(orc_cursor* c, const orc_view* v, int64_t i) {
  const uint64_t h = hash_row(c, v, i);
  for (int64_t g = c->bucket_head[h & (uint64_t)(c->nbuckets - 1)]; g >= 0; g = c->chain_next[g])
    if (c->row_hash[g] == h && key_equal(c, v, i, g)) return g;
  /* row_hash_set.cc:500-511: a key that is not in the set is appended only while the set holds at most
   * max_unique_keys_in_result rows; after that every unseen key is answered with the LAST row of the set */
  if (c->max_unique_keys >= 0 && c->out_rows > c->max_unique_keys) return c->out_rows - 1;
  const int64_t g = c->out_rows;
  if (g >= c->block.cap) {  /* Aggregator grows x2 (aggregate_groups.cc:372-402) */
    const int64_t old = c->block.cap; block_grow(&c->block, old * 2);
    c->chain_next = (int64_t*)realloc(c->chain_next, sizeof(int64_t) * (size_t)c->block.cap);
    c->row_hash = (uint64_t*)realloc(c->row_hash, sizeof(uint64_t) * (size_t)c->block.cap);
    reset_agg_rows(c, c->nproj, old, c->block.cap);
  }
  for (int k = 0; k < c->nproj; ++k) {
    const orc_col* col = &v->c[c->proj_pos[k]]; const int w = c->block.width[k];
    memcpy((char*)c->block.data[k] + g * w, (const char*)col->data + i * w, (size_t)w);
    c->block.nulls[k][g] = col->is_null ? col->is_null[i] : 0;
  }
  c->row_hash[g] = h; c->out_rows = g + 1;
  if (c->out_rows * 4 > c->nbuckets * 3) group_rehash(c, c->nbuckets * 2);
  else { int64_t b = (int64_t)(h & (uint64_t)(c->nbuckets - 1)); c->chain_next[g] = c->bucket_head[b]; c->bucket_head[b] = g; }
  return g;
}

/* ---- Sort: SortPermutation over all key columns (cursor/core/sort.cc:781-805,182-238):
 * NULLs first for ASCENDING, last for DESCENDING; not stable (sort.h:42). ------------- */
typedef struct { const orc_cursor* c; const orc_block* t; } sort_ctx;
static int three_way(int t, const void* p, int64_t a, int64_t b) {
  switch (arith_kind(t)) {
    case 0: { int32_t x = ((const int32_t*)p)[a], y = ((const int32_t*)p)[b]; return x < y ? -1 : y < x; }
    case 1: { uint32_t x = ((const uint32_t*)p)[a], y = ((const uint32_t*)p)[b]; return x < y ? -1 : y < x; }
    case 2: { int64_t x = ((const int64_t*)p)[a], y = ((const int64_t*)p)[b]; return x < y ? -1 : y < x; }
    case 3: { uint64_t x = ((const uint64_t*)p)[a], y = ((const uint64_t*)p)[b]; return x < y ? -1 : y < x; }
    case 4: { float x = ((const float*)p)[a], y = ((const float*)p)[b]; return x < y ? -1 : y < x; }
    case 5: { double x = ((const double*)p)[a], y = ((const double*)p)[b]; return x < y ? -1 : y < x; }
    case 6: { uint8_t x = ((const uint8_t*)p)[a], y = ((const uint8_t*)p)[b]; return x < y ? -1 : y < x; }
  }
  return 0;
}
static int sort_cmp(const void* pa, const void* pb, void* vctx) {
  const sort_ctx* s = (const sort_ctx*)vctx; const int64_t a = *(const int64_t*)pa, b = *(const int64_t*)pb;
  for (int k = 0; k < s->c->nsort; ++k) {
    const int col = s->c->sort_pos[k]; const int desc = s->c->sort_order[k] == 1;
    const int an = s->t->nulls[col][a], bn = s->t->nulls[col][b];
    int r;
    if (an || bn) r = an == bn ? 0 : (an ? -1 : 1);           /* NULL sorts before every value */
    else r = three_way(s->c->child->schema.a[col].type, s->t->data[col], a, b);
    if (r) return desc ? -r : r;
  }
  return 0;
}

static void view_from_block(const orc_cursor* c, const orc_block* b, int64_t off, int64_t rows, orc_view* out) {
  out->n = c->schema.n; out->rows = rows;
  for (int i = 0; i < out->n; ++i) {
    out->c[i].data = (const char*)b->data[i] + off * b->width[i];
    out->c[i].is_null = c->schema.a[i].nullable ? b->nulls[i] + off : NULL;
  }
}

static int cursor_next(orc_cursor* c, int64_t max_rows, orc_view* out) {
  if (c->has_concat) { set_err(&c->err, RC_NOT_IMPLEMENTED, "a specification with CONCAT is evaluated by oracle.py (run), not pulled through this cursor%s%s", "", ""); return -1; }
  if (c->err.code) return -1;
  if (max_rows > ORC_BLOCK) max_rows = ORC_BLOCK;
  switch (c->kind) {
    case C_SCAN: {
      /* ViewCursor::Next -> ViewIterator::next: pointer bump (view_cursor.cc:51-55) */
      if (c->pos >= c->scan.rows) return 0;
      int64_t n = c->scan.rows - c->pos; if (n > max_rows) n = max_rows;
      out->n = c->scan.n; out->rows = n;
      for (int i = 0; i < c->scan.n; ++i) {
        const int w = type_width(c->schema.a[i].type);
        out->c[i].data = (const char*)c->scan.c[i].data + c->pos * w;
        out->c[i].is_null = c->scan.c[i].is_null ? c->scan.c[i].is_null + c->pos : NULL;
      }
      c->pos += n; return 1;
    }
    case C_COMPUTE: {
      /* ComputeCursor::Next, cursor/core/compute.cc:49-56 */
      orc_view in; int r = cursor_next(c->child, max_rows, &in);
      if (r <= 0) { if (r < 0) c->err = c->child->err; return r; }
      out->n = c->nouts; out->rows = in.rows;
      for (int i = 0; i < c->nouts; ++i) {
        eval_node(c->outs[i], &in, in.rows, NULL, &c->err);
        if (c->err.code) return -1;
        out->c[i].data = c->outs[i]->data; out->c[i].is_null = c->outs[i]->nulls;
      }
      return 1;
    }
    case C_HASH_JOIN: {
      /* rhs: drained once into a table; then every lhs row looks its key up (NULL keys never
       * match); INNER keeps matched rows, LEFT_OUTER keeps all with NULL rhs columns; output in
       * lhs order (hash_join.cc HashJoinCursor, UNIQUE rhs keys) */
      if (c->rhs_n < 0) {
        block_init(&c->rhs_rows, &c->rhs->schema, ORC_BLOCK);
        c->rhs_n = 0;
        for (;;) {
          orc_view rv; int r = cursor_next(c->rhs, ORC_BLOCK, &rv);
          if (r < 0) { c->err = c->rhs->err; return -1; }
          if (r == 0) break;
          if (c->rhs_n + rv.rows > c->rhs_rows.cap) block_grow(&c->rhs_rows, (c->rhs_n + rv.rows) * 2);
          for (int k = 0; k < c->rhs->schema.n; ++k) {
            const int w = c->rhs_rows.width[k];
            memcpy((char*)c->rhs_rows.data[k] + c->rhs_n * w, rv.c[k].data, (size_t)rv.rows * w);
            if (rv.c[k].is_null) memcpy(c->rhs_rows.nulls[k] + c->rhs_n, rv.c[k].is_null, (size_t)rv.rows);
            else memset(c->rhs_rows.nulls[k] + c->rhs_n, 0, (size_t)rv.rows);
          }
          c->rhs_n += rv.rows;
        }
        /* rhs keys declared UNIQUE: a repeated non-NULL key is an error (the reference's unique index would
         * keep only one of the rows; the device reports it) */
        for (int64_t a = 0; a < c->rhs_n && c->join_unique; ++a) for (int64_t b = a + 1; b < c->rhs_n && c->rhs_n <= 4096; ++b) {
          int same = 1;
          for (int k = 0; k < c->nkeys && same; ++k) {
            const int col = c->jr_pos[k]; const int w = c->rhs_rows.width[col];
            if (c->rhs_rows.nulls[col][a] || c->rhs_rows.nulls[col][b]) same = 0;
            else same = memcmp((char*)c->rhs_rows.data[col] + a * w, (char*)c->rhs_rows.data[col] + b * w, (size_t)w) == 0;
          }
          if (same) { set_err(&c->err, 407, "hash join: rhs keys declared UNIQUE but a key repeats%s%s", "", ""); return -1; }
        }
      }
      while (c->join_served >= c->join_pending) {
        orc_view in; int r = cursor_next(c->child, max_rows, &in);
        if (r <= 0) { if (r < 0) c->err = c->child->err; return r; }
        int64_t nout = 0;
        for (int64_t i = 0; i < in.rows; ++i) {
          int null_key = 0; int64_t n_found = 0;
          for (int k = 0; k < c->nkeys; ++k) if (in.c[c->jl_pos[k]].is_null && in.c[c->jl_pos[k]].is_null[i]) null_key = 1;
          /* every matching rhs row, in rhs order (RowIdSetIterator walks the equal-row list in insertion
           * order, row_hash_set.cc:581-600,650-652); one NULL-extended row for an unmatched LEFT_OUTER lhs row */
          for (int64_t j = 0; j <= c->rhs_n; ++j) {
            int64_t found = -1;
            if (j < c->rhs_n) {
              if (null_key) continue;
              int same = 1;
              for (int k = 0; k < c->nkeys && same; ++k) {
                const int col = c->jr_pos[k]; const int w = c->rhs_rows.width[col];
                if (c->rhs_rows.nulls[col][j]) same = 0;
                else same = memcmp((const char*)in.c[c->jl_pos[k]].data + i * w, (char*)c->rhs_rows.data[col] + j * w, (size_t)w) == 0;
              }
              if (!same) continue;
              found = j; ++n_found;
            } else if (n_found > 0 || c->join_type == 0) {
              break;
            }
            if (nout >= c->block.cap) block_grow(&c->block, c->block.cap * 2);
            for (int q = 0; q < c->nout_cols; ++q) {
              const int w = c->block.width[q];
              if (c->out_src[q] == 0) {
                const orc_col* src = &in.c[c->out_pos[q]];
                memcpy((char*)c->block.data[q] + nout * w, (const char*)src->data + i * w, (size_t)w);
                c->block.nulls[q][nout] = src->is_null ? src->is_null[i] : 0;
              } else if (found >= 0) {
                memcpy((char*)c->block.data[q] + nout * w, (char*)c->rhs_rows.data[c->out_pos[q]] + found * w, (size_t)w);
                c->block.nulls[q][nout] = c->rhs_rows.nulls[c->out_pos[q]][found];
              } else { memset((char*)c->block.data[q] + nout * w, 0, (size_t)w); c->block.nulls[q][nout] = 1; }
            }
            ++nout;
          }
        }
        if (nout == 0) continue;   /* nothing survived this input view: pull the next one */
        c->join_pending = nout; c->join_served = 0;
        break;
      }
      /* NOT_UNIQUE keys can multiply an input view past max_rows: serve the joined rows in slices */
      {
        int64_t take = c->join_pending - c->join_served; if (take > max_rows) take = max_rows;
        view_from_block(c, &c->block, c->join_served, take, out);
        c->join_served += take;
        return 1;
      }
    }
    case C_PROJECT: {
      /* ProjectCursor::Next: pointer re-mapping (cursor/core/project.cc:49-59) */
      orc_view in; int r = cursor_next(c->child, max_rows, &in);
      if (r <= 0) { if (r < 0) c->err = c->child->err; return r; }
      out->n = c->nproj; out->rows = in.rows;
      for (int i = 0; i < c->nproj; ++i) out->c[i] = in.c[c->proj_pos[i]];
      return 1;
    }
    case C_FILTER: {
      /* FilterCursor::Next, cursor/core/filter.cc:96-128: fill the result block until it is
       * >= 25% full (kMinimumFillPercent, :51,215-217) or the input ends */
      int64_t write_ptr = 0;
      for (;;) {
        if (c->have_cur && c->read_ptr < c->nids) {
          int64_t n = c->nids - c->read_ptr; if (n > max_rows - write_ptr) n = max_rows - write_ptr;
          const int final = 100 * (n + write_ptr) >= 25 * max_rows;
          /* gather: dst[w+i] = src[ids[i]] per projected column (copy_column.cc:199-217) */
          for (int k = 0; k < c->nproj; ++k) {
            const orc_col* src = &c->cur.c[c->proj_pos[k]]; const int w = c->block.width[k];
            const int64_t* ids = c->ids + c->read_ptr;
            if (w == 8) { const uint64_t* s = (const uint64_t*)src->data; uint64_t* d = (uint64_t*)c->block.data[k] + write_ptr; for (int64_t i = 0; i < n; ++i) d[i] = s[ids[i]]; }
            else if (w == 4) { const uint32_t* s = (const uint32_t*)src->data; uint32_t* d = (uint32_t*)c->block.data[k] + write_ptr; for (int64_t i = 0; i < n; ++i) d[i] = s[ids[i]]; }
            else { const uint8_t* s = (const uint8_t*)src->data; uint8_t* d = (uint8_t*)c->block.data[k] + write_ptr; for (int64_t i = 0; i < n; ++i) d[i] = s[ids[i]]; }
            uint8_t* dn = c->block.nulls[k] + write_ptr;
            if (src->is_null) for (int64_t i = 0; i < n; ++i) dn[i] = src->is_null[ids[i]]; else memset(dn, 0, (size_t)n);
          }
          c->read_ptr += n; write_ptr += n;
          if (final) break;
        } else {
          if (c->eos) { if (write_ptr) break; return 0; }
          int r = cursor_next(c->child, max_rows < ORC_BLOCK ? max_rows : ORC_BLOCK, &c->cur);
          if (r < 0) { c->err = c->child->err; return -1; }
          if (r == 0) { c->eos = 1; c->have_cur = 0; if (write_ptr) break; return 0; }
          /* PrepareInputRowIds, filter.cc:170-199: ids of rows whose predicate is non-NULL TRUE */
          eval_node(c->pred, &c->cur, c->cur.rows, NULL, &c->err);
          if (c->err.code) return -1;
          const uint8_t* pv = (const uint8_t*)c->pred->data; const uint8_t* pn = c->pred->nulls;
          int64_t k = 0;
          if (pn) { for (int64_t i = 0; i < c->cur.rows; ++i) if (!pn[i] && pv[i]) c->ids[k++] = i; }
          else { for (int64_t i = 0; i < c->cur.rows; ++i) if (pv[i]) c->ids[k++] = i; }
          c->nids = k; c->read_ptr = 0; c->have_cur = 1;
        }
      }
      view_from_block(c, &c->block, 0, write_ptr, out);
      return 1;
    }
    case C_SCALAR_AGG: {
      /* ScalarAggregateCursor::Next, cursor/core/aggregate_scalar.cc:53-68: drain the child,
       * UpdateAggregations(view, zeros); exactly one output row even on empty input */
      if (c->done) return 0;
      static const int64_t zeros[ORC_BLOCK] = {0};
      reset_agg_rows(c, 0, 0, 1);
      orc_view in; int r;
      while ((r = cursor_next(c->child, ORC_BLOCK, &in)) > 0)
        for (int j = 0; j < c->nagg; ++j) update_aggregation(&c->aggs[j], &in, zeros, c->block.data[j], c->block.nulls[j]);
      if (r < 0) { c->err = c->child->err; return -1; }
      c->done = 1;
      view_from_block(c, &c->block, 0, 1, out);
      return 1;
    }
    case C_GROUP_AGG: {
      if (c->best_effort_groups > 0) {
        /* BestEffortGroupAggregate: GroupAggregateCursor::Next / ProcessInput with best_effort_ (aggregate_groups.cc:211-222,
         * 332-433).  The reference aggregates input until its key set / result block cannot take another key (an allocator
         * verdict: deterministic under GuaranteeMemory, aggregate_groups.cc:160-163), emits what it has, and when that has
         * been read resets key set and aggregator (:333-345) and goes on with the input rows it had not consumed
         * (child_.truncate, :362).  Restated with the result block's row capacity as the verdict: a view aggregates the
         * longest run of input rows, starting where the last one stopped, that holds at most `best_effort_groups` keys. */
        for (;;) {
          if (c->done && c->emit_pos < c->out_rows) {
            int64_t n = c->out_rows - c->emit_pos; if (n > max_rows) n = max_rows;
            view_from_block(c, &c->block, c->emit_pos, n, out);
            c->emit_pos += n; return 1;
          }
          if (c->done && c->eos && !(c->have_cur && c->read_ptr < c->cur.rows)) return 0;
          /* ProcessInput */
          if (!c->chain_next) {
            c->chain_next = (int64_t*)malloc(sizeof(int64_t) * (size_t)c->block.cap);
            c->row_hash = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)c->block.cap);
          }
          c->nbuckets = 32; c->out_rows = 0; group_rehash(c, c->nbuckets);
          reset_agg_rows(c, c->nproj, 0, c->block.cap);
          for (int j = 0; j < c->nagg; ++j) { agg_col* g = &c->aggs[j]; if (g->distinct && g->dcap) { for (int64_t q = 0; q < g->dcap; ++q) g->drow[q] = -1; g->dcount = 0; } }
          int full = 0; int64_t consumed = 0;
          while (!full) {
            if (!(c->have_cur && c->read_ptr < c->cur.rows)) {
              if (c->eos) break;
              int r = cursor_next(c->child, ORC_BLOCK, &c->cur);
              if (r < 0) { c->err = c->child->err; return -1; }
              if (r == 0) { c->eos = 1; c->have_cur = 0; break; }
              c->have_cur = 1; c->read_ptr = 0;
            }
            /* the rest of the pending block, as a view of its own */
            orc_view in = c->cur; in.rows = c->cur.rows - c->read_ptr;
            for (int k = 0; k < in.n; ++k) {
              const int w = type_width(c->child->schema.a[k].type);
              in.c[k].data = (const char*)c->cur.c[k].data + c->read_ptr * w;
              if (in.c[k].is_null) in.c[k].is_null = c->cur.c[k].is_null + c->read_ptr;
            }
            int64_t map[ORC_BLOCK]; int64_t take = 0;
            const int64_t saved_limit = c->max_unique_keys; c->max_unique_keys = -1;
            for (; take < in.rows; ++take) {
              const int64_t before = c->out_rows;
              if (before >= c->best_effort_groups) {
                /* would this row need a new group?  look it up without inserting */
                const uint64_t h = hash_row(c, &in, take); int64_t g = c->bucket_head[h & (uint64_t)(c->nbuckets - 1)];
                for (; g >= 0; g = c->chain_next[g]) if (c->row_hash[g] == h && key_equal(c, &in, take, g)) break;
                if (g < 0) { full = 1; break; }
                map[take] = g; continue;
              }
              map[take] = group_insert(c, &in, take);
            }
            c->max_unique_keys = saved_limit;
            in.rows = take;
            for (int j = 0; j < c->nagg; ++j) update_aggregation(&c->aggs[j], &in, map, c->block.data[c->nproj + j], c->block.nulls[c->nproj + j]);
            c->read_ptr += take; consumed += take;
          }
          c->done = 1; c->emit_pos = 0;
          if (c->out_rows == 0) return 0;      /* (no input rows at all) */
        }
      }
      /* GroupAggregateCursor::ProcessInput, cursor/core/aggregate_groups.cc:332-433: consume
       * ALL input, then iterate the result (keys || aggregates, first-seen order) */
      if (!c->done) {
        c->nbuckets = 32; c->bucket_head = NULL; c->out_rows = 0;
        c->chain_next = (int64_t*)malloc(sizeof(int64_t) * (size_t)c->block.cap);
        c->row_hash = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)c->block.cap);
        group_rehash(c, c->nbuckets);
        reset_agg_rows(c, c->nproj, 0, c->block.cap);
        int64_t map[ORC_BLOCK]; orc_view in; int r;
        while ((r = cursor_next(c->child, ORC_BLOCK, &in)) > 0) {
          for (int64_t i = 0; i < in.rows; ++i) map[i] = group_insert(c, &in, i);
          for (int j = 0; j < c->nagg; ++j) update_aggregation(&c->aggs[j], &in, map, c->block.data[c->nproj + j], c->block.nulls[c->nproj + j]);
        }
        if (r < 0) { c->err = c->child->err; return -1; }
        c->done = 1; c->emit_pos = 0;
      }
      if (c->emit_pos >= c->out_rows) return 0;
      int64_t n = c->out_rows - c->emit_pos; if (n > max_rows) n = max_rows;
      view_from_block(c, &c->block, c->emit_pos, n, out);
      c->emit_pos += n; return 1;
    }
    case C_CLUSTERS: {
      /* AggregateClustersCursor, cursor/core/aggregate_clusters.cc:338-520: a new output row
       * starts whenever any key column differs from the previous input row (ColumnEqual
       * :97-122, NULL == NULL); streaming in the reference, materialised here (same rows) */
      if (!c->done) {
        c->out_rows = 0; reset_agg_rows(c, c->nproj, 0, c->block.cap);
        int64_t map[ORC_BLOCK]; orc_view in; int r;
        while ((r = cursor_next(c->child, ORC_BLOCK, &in)) > 0) {
          for (int64_t i = 0; i < in.rows; ++i) {
            int same = c->out_rows > 0 && key_equal(c, &in, i, c->out_rows - 1);
            if (!same) {
              if (c->out_rows >= c->block.cap) { int64_t old = c->block.cap; block_grow(&c->block, old * 2); reset_agg_rows(c, c->nproj, old, c->block.cap); }
              for (int k = 0; k < c->nproj; ++k) {
                const orc_col* col = &in.c[c->proj_pos[k]]; const int w = c->block.width[k];
                memcpy((char*)c->block.data[k] + c->out_rows * w, (const char*)col->data + i * w, (size_t)w);
                c->block.nulls[k][c->out_rows] = col->is_null ? col->is_null[i] : 0;
              }
              c->out_rows++;
            }
            map[i] = c->out_rows - 1;
          }
          for (int j = 0; j < c->nagg; ++j) update_aggregation(&c->aggs[j], &in, map, c->block.data[c->nproj + j], c->block.nulls[c->nproj + j]);
        }
        if (r < 0) { c->err = c->child->err; return -1; }
        c->done = 1; c->emit_pos = 0;
      }
      if (c->emit_pos >= c->out_rows) return 0;
      int64_t n = c->out_rows - c->emit_pos; if (n > max_rows) n = max_rows;
      view_from_block(c, &c->block, c->emit_pos, n, out);
      c->emit_pos += n; return 1;
    }
    case C_SORT: {
      /* SortCursor::ProcessData, cursor/core/sort.cc:636-650: deep-copy the input into a Table,
       * sort an int64 permutation, then gather 1024 rows per Next (view_cursor.cc:97-118) */
      if (!c->done) {
        orc_block* t = (orc_block*)calloc(1, sizeof(orc_block));
        block_init(t, &c->child->schema, ORC_BLOCK);
        int64_t rows = 0; orc_view in; int r;
        while ((r = cursor_next(c->child, ORC_BLOCK, &in)) > 0) {
          if (rows + in.rows > t->cap) block_grow(t, (rows + in.rows) * 2);
          for (int k = 0; k < t->n; ++k) {
            memcpy((char*)t->data[k] + rows * t->width[k], in.c[k].data, (size_t)in.rows * (size_t)t->width[k]);
            if (in.c[k].is_null) memcpy(t->nulls[k] + rows, in.c[k].is_null, (size_t)in.rows); else memset(t->nulls[k] + rows, 0, (size_t)in.rows);
          }
          rows += in.rows;
        }
        if (r < 0) { c->err = c->child->err; return -1; }
        c->perm = (int64_t*)malloc(sizeof(int64_t) * (size_t)(rows > 0 ? rows : 1));
        for (int64_t i = 0; i < rows; ++i) c->perm[i] = i;
        sort_ctx s; s.c = c; s.t = t;
        qsort_r(c->perm, (size_t)rows, sizeof(int64_t), sort_cmp, &s);
        c->out_rows = rows; c->emit_pos = 0; c->done = 1;
        block_init(&c->block, &c->schema, ORC_BLOCK);
        c->table = t;
      }
      if (c->emit_pos >= c->out_rows) return 0;
      int64_t n = c->out_rows - c->emit_pos; if (n > max_rows) n = max_rows;
      const orc_block* t = (const orc_block*)c->table;
      for (int k = 0; k < c->nproj; ++k) {
        const int col = c->proj_pos[k]; const int w = t->width[col]; const int64_t* ids = c->perm + c->emit_pos;
        for (int64_t i = 0; i < n; ++i) memcpy((char*)c->block.data[k] + i * w, (const char*)t->data[col] + ids[i] * w, (size_t)w);
        for (int64_t i = 0; i < n; ++i) c->block.nulls[k][i] = t->nulls[col][ids[i]];
      }
      view_from_block(c, &c->block, 0, n, out);
      c->emit_pos += n; return 1;
    }
  }
  return -1;
}

/* ---- public pull API --------------------------------------------------------------- */
int orc_next(orc_cursor* c, int64_t max_rows, int64_t* rows, const void** data, const uint8_t** is_null) {
  orc_view v; memset(&v, 0, sizeof(v));
  int r = cursor_next(c, max_rows, &v);
  if (r <= 0) return r;
  *rows = v.rows;
  for (int i = 0; i < c->schema.n; ++i) { data[i] = v.c[i].data; is_null[i] = v.c[i].is_null; }
  return 1;
}

/* drain a cursor completely, discarding rows: used by bench.py's cpu_baseline leg
 * (returns the number of result rows, or -1 on failure) */
int64_t orc_drain(orc_cursor* c) {
  orc_view v; int r; int64_t total = 0;
  while ((r = cursor_next(c, ORC_BLOCK, &v)) > 0) total += v.rows;
  return r < 0 ? -1 : total;
}

/* ==== bench.py's CPU baseline harness ====================================================================================
 * Not part of the restatement: the N-thread form of the CPU baseline SURVEY 8(d) asks for -- "N threads over row-range
 * shards with a final merge" -- done where it costs nothing extra: pthreads pinned to the CPUs the caller names, every thread
 * first-touches its own shard of the input (NUMA-local pages), a barrier, `passes` drains of a fresh cursor over the shard
 * per thread, a barrier, and for GroupAggregate plans the merge of the N partial tables (every thread owns the groups whose
 * key hash is its number).  Wall time is taken between the barriers by the calling thread. */
#include <pthread.h>
#include <sched.h>
#include <time.h>

static void block_free(orc_block* b) {
  for (int i = 0; i < b->n; ++i) { free(b->data[i]); free(b->nulls[i]); }
  memset(b, 0, sizeof(*b));
}

/* releases a cursor tree; bound nodes only if the creating thread tracked them (tl_nodes) */
static void cursor_free(orc_cursor* c) {
  if (!c) return;
  cursor_free(c->child); cursor_free(c->rhs);
  block_free(&c->block); block_free(&c->rhs_rows);
  free(c->ids); free(c->bucket_head); free(c->chain_next); free(c->row_hash); free(c->perm); free(c->match);
  if (c->table) { block_free((orc_block*)c->table); free(c->table); }
  for (int j = 0; j < c->nagg; ++j) { free(c->aggs[j].dval); free(c->aggs[j].drow); }
  free(c);
}
static void nodes_free(node_list* l) {
  for (int64_t i = 0; i < l->n; ++i) {
    bnode* b = l->node[i];
    if (l->owns_bufs[i]) { free(b->buf); free(b->nullbuf); }
    /* (an alias shares its source's result buffers -- copied at bind time, before any skip vector existed: skipbuf is its own) */
    free(b->skipbuf);
    free(b);
  }
  l->n = 0;
}

typedef struct {
  const orc_op* op; const orc_op* sample;     /* sample: a plan whose ScanView holds `sample rows` of the same schema, or NULL */
  int nthreads, passes, tid, cpu, merge;
  int64_t lo, hi;                              /* this thread's row range of op's ScanView */
  pthread_barrier_t* bar;
  orc_cursor* last;                            /* the last pass's cursor (kept for the merge) */
  int64_t result_rows; int failed;
  struct bench_thread_s* all;
  /* merge output */ int64_t merged_groups; double merged_checksum;
} bench_thread_base;
typedef struct bench_thread_s { bench_thread_base b; } bench_thread;

static const orc_op* scan_of(const orc_op* op) { while (op && op->kind != C_SCAN) op = op->child; return op; }
static orc_cursor* scan_cursor_of(orc_cursor* c) { while (c && c->kind != C_SCAN) c = c->child; return c; }

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

/* merge of partial GroupAggregate tables: thread t inserts the rows of EVERY partial table whose key hash % nthreads == t
 * into its own chained table and folds the aggregates with each column's own function (SUM / COUNT add, MIN / MAX compare;
 * NULL = no value yet) -- the key-range all-to-all of a sharded hash aggregate, on shared memory. */
static int merge_supported(const orc_cursor* c) {
  if (c->kind != C_GROUP_AGG || c->max_unique_keys >= 0) return 0;
  for (int j = 0; j < c->nagg; ++j) {
    const int a = c->aggs[j].aggregation;
    if (c->aggs[j].distinct || !(a == A_SUM || a == A_MIN || a == A_MAX || a == A_COUNT)) return 0;
    if (arith_kind(c->aggs[j].out_type) > 5) return 0;
  }
  return 1;
}
static uint64_t hash_block_row(const orc_cursor* c, const orc_block* b, int64_t i) {
  uint64_t h = 0x9e3779b97f4a7c15ull;
  for (int k = 0; k < c->nproj; ++k) {
    uint64_t x = 0;
    if (b->nulls[k][i]) x = 0xdeadbeefcafef00dull; else memcpy(&x, (const char*)b->data[k] + i * b->width[k], (size_t)b->width[k]);
    h = mix64(h ^ x) + 0x9e3779b97f4a7c15ull * (uint64_t)(k + 1);
  }
  return h;
}
static void merge_partials(bench_thread* me) {
  bench_thread* all = me->b.all; const int nt = me->b.nthreads;
  const orc_cursor* c0 = all[0].b.last; const int nk = c0->nproj, na = c0->nagg;
  orc_block out; block_init(&out, &c0->schema, 1024);
  int64_t nb = 2048, n = 0; int64_t* head = (int64_t*)malloc(sizeof(int64_t) * (size_t)nb);
  int64_t* next = (int64_t*)malloc(sizeof(int64_t) * (size_t)out.cap); uint64_t* hs = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)out.cap);
  for (int64_t i = 0; i < nb; ++i) head[i] = -1;
  for (int p = 0; p < nt; ++p) {
    const orc_cursor* c = all[p].b.last; const orc_block* b = &c->block;
    for (int64_t r = 0; r < c->out_rows; ++r) {
      const uint64_t h = c->row_hash[r];      /* (the hash the shard's own table computed for the row: hash_row == hash_block_row) */
      if ((int)((h >> 32) % (uint64_t)nt) != me->b.tid) continue;
      int64_t g = head[h & (uint64_t)(nb - 1)];
      for (; g >= 0; g = next[g]) {
        if (hs[g] != h) continue;
        int same = 1;
        for (int k = 0; k < nk && same; ++k) {
          if (out.nulls[k][g] != b->nulls[k][r]) same = 0;
          else if (!b->nulls[k][r] && memcmp((char*)out.data[k] + g * out.width[k], (char*)b->data[k] + r * b->width[k], (size_t)out.width[k])) same = 0;
        }
        if (same) break;
      }
      if (g < 0) {
        if (n == out.cap) {
          block_grow(&out, out.cap * 2);
          next = (int64_t*)realloc(next, sizeof(int64_t) * (size_t)out.cap); hs = (uint64_t*)realloc(hs, sizeof(uint64_t) * (size_t)out.cap);
        }
        g = n++;
        for (int k = 0; k < nk + na; ++k) { memcpy((char*)out.data[k] + g * out.width[k], (char*)b->data[k] + r * b->width[k], (size_t)out.width[k]); out.nulls[k][g] = b->nulls[k][r]; }
        hs[g] = h;
        if (n * 4 > nb * 3) {
          nb *= 2; free(head); head = (int64_t*)malloc(sizeof(int64_t) * (size_t)nb);
          for (int64_t i = 0; i < nb; ++i) head[i] = -1;
          for (int64_t q = 0; q < n; ++q) { const int64_t s = (int64_t)(hs[q] & (uint64_t)(nb - 1)); next[q] = head[s]; head[s] = q; }
        } else { const int64_t s = (int64_t)(h & (uint64_t)(nb - 1)); next[g] = head[s]; head[s] = g; }
        continue;
      }
      for (int j = 0; j < na; ++j) {
        const int col = nk + j, a = c->aggs[j].aggregation;
        if (b->nulls[col][r]) continue;
        if (out.nulls[col][g]) { memcpy((char*)out.data[col] + g * out.width[col], (char*)b->data[col] + r * b->width[col], (size_t)out.width[col]); out.nulls[col][g] = 0; continue; }
#define MERGE(T) { T* d = (T*)out.data[col] + g; const T v = ((const T*)b->data[col])[r]; \
          if (a == A_SUM || a == A_COUNT) *d = (T)(*d + v); else if (a == A_MIN) { if (v < *d) *d = v; } else { if (v > *d) *d = v; } }
        switch (arith_kind(c->aggs[j].out_type)) {
          case 0: MERGE(int32_t) break; case 1: MERGE(uint32_t) break; case 2: MERGE(int64_t) break;
          case 3: MERGE(uint64_t) break; case 4: MERGE(float) break; default: MERGE(double) break; }
#undef MERGE
      }
    }
  }
  me->b.merged_groups = n;
  /* a checksum the caller compares with the one-thread result: the sum of the first aggregate column (as double) */
  double sum = 0;
  if (na > 0) for (int64_t g = 0; g < n; ++g) if (!out.nulls[nk][g]) {
    switch (arith_kind(c0->aggs[0].out_type)) {
      case 0: sum += (double)((int32_t*)out.data[nk])[g]; break; case 1: sum += (double)((uint32_t*)out.data[nk])[g]; break;
      case 2: sum += (double)((int64_t*)out.data[nk])[g]; break; case 3: sum += (double)((uint64_t*)out.data[nk])[g]; break;
      case 4: sum += (double)((float*)out.data[nk])[g]; break; default: sum += ((double*)out.data[nk])[g]; break; }
  }
  me->b.merged_checksum = sum;
  block_free(&out); free(head); free(next); free(hs);
}

static void* bench_thread_main(void* arg) {
  bench_thread* me = (bench_thread*)arg;
  if (me->b.cpu >= 0) { cpu_set_t set; CPU_ZERO(&set); CPU_SET(me->b.cpu, &set); pthread_setaffinity_np(pthread_self(), sizeof(set), &set); }
  const orc_op* scan = scan_of(me->b.op);
  /* first touch: this thread writes its own rows (copies of the sample, continued at the shard's own offset into it) */
  if (me->b.sample) {
    const orc_op* ss = scan_of(me->b.sample); const int64_t sn = ss->scan_view.rows;
    for (int k = 0; k < scan->scan_view.n && sn > 0; ++k) {
      const int w = type_width(scan->scan_schema.a[k].type);
      for (int64_t r = me->b.lo; r < me->b.hi; ) {
        const int64_t off = r % sn; int64_t n = sn - off; if (n > me->b.hi - r) n = me->b.hi - r;
        memcpy((char*)scan->scan_view.c[k].data + r * w, (const char*)ss->scan_view.c[k].data + off * w, (size_t)n * (size_t)w);
        if (scan->scan_view.c[k].is_null && ss->scan_view.c[k].is_null) memcpy((uint8_t*)scan->scan_view.c[k].is_null + r, ss->scan_view.c[k].is_null + off, (size_t)n);
        r += n;
      }
    }
  }
  node_list nodes; memset(&nodes, 0, sizeof(nodes));
  tl_nodes = &nodes;
  pthread_barrier_wait(me->b.bar);                 /* ---- timed region starts (the caller reads the clock after this barrier) */
  for (int p = 0; p < me->b.passes; ++p) {
    if (me->b.last) { cursor_free(me->b.last); nodes_free(&nodes); }
    orc_cursor* c = orc_create_cursor(me->b.op);
    orc_cursor* sc = scan_cursor_of(c);
    if (c->err.code || !sc) { me->b.failed = 1; me->b.last = c; break; }
    for (int k = 0; k < sc->scan.n; ++k) {          /* this thread's contiguous row range of the ScanView */
      const int w = type_width(sc->schema.a[k].type);
      sc->scan.c[k].data = (const char*)sc->scan.c[k].data + me->b.lo * w;
      if (sc->scan.c[k].is_null) sc->scan.c[k].is_null += me->b.lo;
    }
    sc->scan.rows = me->b.hi - me->b.lo;
    me->b.result_rows = orc_drain(c);
    if (me->b.result_rows < 0) me->b.failed = 1;
    me->b.last = c;
  }
  pthread_barrier_wait(me->b.bar);                 /* ---- all shards done */
  if (me->b.merge) { merge_partials(me); pthread_barrier_wait(me->b.bar); }   /* ---- merged */
  tl_nodes = NULL; nodes_free(&nodes); free(nodes.node); free(nodes.owns_bufs);
  return NULL;
}

/* out[0] = seconds of the passes (all threads, barrier to barrier), out[1] = seconds of the merge (0 if none), out[2] = groups
 * after the merge, out[3] = checksum after the merge, out[4] = result rows of the last pass summed over the threads.
 * cpus: one CPU id per thread to pin to, or NULL.  merge: 1 = merge partial GroupAggregate tables when the plan allows it.
 * Returns 0, or -1 if a cursor failed, -2 if the merge was asked for and is not possible for this plan (out[1] = -1). */
int orc_bench_threads(const orc_op* op, const orc_op* sample, int nthreads, int passes, const int* cpus, int merge, double* out) {
  const orc_op* scan = scan_of(op);
  if (!scan || nthreads < 1 || passes < 1) return -1;
  const int64_t rows = scan->scan_view.rows;
  bench_thread* th = (bench_thread*)calloc((size_t)nthreads, sizeof(bench_thread));
  pthread_t* ids = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
  pthread_barrier_t bar; pthread_barrier_init(&bar, NULL, (unsigned)nthreads + 1);
  int can_merge = 0;
  if (merge) { orc_cursor* probe = orc_create_cursor(op); can_merge = !probe->err.code && merge_supported(probe); }
  for (int t = 0; t < nthreads; ++t) {
    th[t].b.op = op; th[t].b.sample = sample; th[t].b.nthreads = nthreads; th[t].b.passes = passes; th[t].b.tid = t;
    th[t].b.cpu = cpus ? cpus[t] : -1; th[t].b.merge = can_merge; th[t].b.bar = &bar; th[t].b.all = th;
    th[t].b.lo = rows * t / nthreads; th[t].b.hi = rows * (t + 1) / nthreads;
    pthread_create(&ids[t], NULL, bench_thread_main, &th[t]);
  }
  pthread_barrier_wait(&bar); const double t0 = now_s();
  pthread_barrier_wait(&bar); const double t1 = now_s();
  double t2 = t1;
  if (can_merge) { pthread_barrier_wait(&bar); t2 = now_s(); }
  int failed = 0; double groups = 0, checksum = 0, result_rows = 0;
  for (int t = 0; t < nthreads; ++t) {
    pthread_join(ids[t], NULL);
    failed |= th[t].b.failed; groups += (double)th[t].b.merged_groups; checksum += th[t].b.merged_checksum; result_rows += (double)th[t].b.result_rows;
    cursor_free(th[t].b.last);
  }
  pthread_barrier_destroy(&bar); free(th); free(ids);
  out[0] = t1 - t0; out[1] = can_merge ? t2 - t1 : (merge ? -1.0 : 0.0); out[2] = groups; out[3] = checksum; out[4] = result_rows;
  return failed ? -1 : (merge && !can_merge ? -2 : 0);
}

/* the host's streaming READ bandwidth with the same threads over the same columns (sum of 64-bit words, every column of the
 * ScanView once per pass): the ceiling the N-thread figure is read against.  Returns bytes per second. */
typedef struct { const orc_op* scan; int64_t lo, hi; int passes, cpu; pthread_barrier_t* bar; uint64_t sink; } stream_thread;
static void* stream_thread_main(void* arg) {
  stream_thread* me = (stream_thread*)arg;
  if (me->cpu >= 0) { cpu_set_t set; CPU_ZERO(&set); CPU_SET(me->cpu, &set); pthread_setaffinity_np(pthread_self(), sizeof(set), &set); }
  pthread_barrier_wait(me->bar);
  uint64_t acc = 0;
  for (int p = 0; p < me->passes; ++p)
    for (int k = 0; k < me->scan->scan_view.n; ++k) {
      const int w = type_width(me->scan->scan_schema.a[k].type);
      const uint64_t* d = (const uint64_t*)((const char*)me->scan->scan_view.c[k].data + me->lo * w);
      const int64_t words = (me->hi - me->lo) * w / 8;
      uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      int64_t i = 0;
      for (; i + 4 <= words; i += 4) { a0 += d[i]; a1 += d[i + 1]; a2 += d[i + 2]; a3 += d[i + 3]; }
      for (; i < words; ++i) a0 += d[i];
      acc += a0 + a1 + a2 + a3;
    }
  me->sink = acc;
  pthread_barrier_wait(me->bar);
  return NULL;
}
double orc_bench_stream_read(const orc_op* op, int nthreads, int passes, const int* cpus) {
  const orc_op* scan = scan_of(op);
  if (!scan || nthreads < 1 || passes < 1) return 0.0;
  const int64_t rows = scan->scan_view.rows;
  stream_thread* th = (stream_thread*)calloc((size_t)nthreads, sizeof(stream_thread));
  pthread_t* ids = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
  pthread_barrier_t bar; pthread_barrier_init(&bar, NULL, (unsigned)nthreads + 1);
  int64_t bytes_per_row = 0;
  for (int k = 0; k < scan->scan_view.n; ++k) bytes_per_row += type_width(scan->scan_schema.a[k].type);
  for (int t = 0; t < nthreads; ++t) {
    th[t].scan = scan; th[t].lo = rows * t / nthreads; th[t].hi = rows * (t + 1) / nthreads; th[t].passes = passes;
    th[t].cpu = cpus ? cpus[t] : -1; th[t].bar = &bar;
    pthread_create(&ids[t], NULL, stream_thread_main, &th[t]);
  }
  pthread_barrier_wait(&bar); const double t0 = now_s();
  pthread_barrier_wait(&bar); const double t1 = now_s();
  uint64_t sink = 0;
  for (int t = 0; t < nthreads; ++t) { pthread_join(ids[t], NULL); sink += th[t].sink; }
  pthread_barrier_destroy(&bar); free(th); free(ids);
  return (double)bytes_per_row * (double)rows * (double)passes / ((t1 - t0) > 0 ? (t1 - t0) : 1e-9) + (sink == 0x123456789abcdefull ? 1e-30 : 0.0);
}
