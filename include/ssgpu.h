/*
 * ssgpu.h -- C ABI of the MI355X-native column-block pipeline (libssgpu.so).
 *
 * This is the drop-in boundary for Supersonic's Filter -> Project/Compute ->
 * Aggregate (+Sort) hot path.  The reference has no FFI of its own; its
 * extension seams are C++ virtual interfaces.  Each entry point below names
 * the reference interface it stands in for (paths relative to the reference
 * tree):
 *
 *   ssgpu_plan_create      Operation::CreateCursor() + Expression::Bind()
 *                            supersonic/cursor/base/operation.h:62
 *                            supersonic/expression/base/expression.cc:84-94
 *                          (all binding, type promotion, naming and the
 *                           400-range bind errors happen here)
 *   ssgpu_plan_attr        Cursor::schema()            cursor/base/cursor.h:135
 *   ssgpu_plan_run         the Cursor::Next() pull loop drained to the end
 *                            cursor/base/cursor.h:148
 *                            cursor/core/aggregate_scalar.cc:53-68
 *                            cursor/core/filter.cc:96-128
 *                            cursor/core/aggregate_groups.cc:332-433
 *                            cursor/core/sort.cc:590-650
 *                          and BoundExpressionTree::Evaluate over a View
 *                            expression/base/expression.cc:57-76
 *   ssgpu_result_*         ResultView / View        cursor/base/cursor.h:42-122
 *                            base/infrastructure/block.h:288-402
 *   ssgpu_interrupt        Cursor::Interrupt()   cursor/base/cursor.h:150-186
 *   ssgpu_block_*          Block / Table (owning column storage)
 *                            base/infrastructure/block.h:412
 *                            cursor/infrastructure/table.h:49
 *   ssgpu_expr_bind        Expression::DoBind(schema, allocator, max_row_count)
 *                            supersonic/expression/base/expression.h:164-167
 *   ssgpu_expr_evaluate    BoundExpressionTree::Evaluate(const View&)
 *                            supersonic/expression/base/expression.h:116, expression.cc:57-76
 *   ssgpu_allocator_*      BufferAllocator (Allocate / BestEffortAllocate / Reallocate / Available, Buffer dtor) and
 *                          MemoryLimit (soft quota)    base/memory/memory.h:100-233,465-520
 *   ssgpu_plan_set_memory_limit   Operation::SetBufferAllocator(MemoryLimit*)   cursor/base/operation.h:66-76
 *   ssgpu_dict_*           STRING columns: StringPiece + Arena (base/memory/arena.h) become INT32 codes of an
 *                          order-preserving dictionary (see "STRING columns" below)
 *   ssgpu_host_alloc/free  BufferAllocator::Allocate / Buffer dtor
 *                            base/memory/memory.h:100-233
 *
 * Conventions
 *   - plain C, no exceptions cross this boundary, no torch types;
 *   - every function that can fail returns a reference ReturnCode integer
 *     (supersonic/proto/supersonic.proto:40-82): 0 = OK, 102 = MEMORY_EXCEEDED,
 *     103 = NOT_IMPLEMENTED, 104 = EVALUATION_ERROR, 4xx = schema/bind errors,
 *     1000 = INTERRUPTED.  ssgpu_last_error() returns the message;
 *   - DataType / Aggregation / ColumnOrder / OperatorId integers are the
 *     reference's proto enum values (supersonic.proto:15-36,86-101;
 *     expression/proto/operators.proto);
 *   - null masks are one byte per row (0 = value present, non-zero = NULL),
 *     exactly as the reference's bool* is_null (bit_pointers.h:529-533);
 *   - STRING columns cross this boundary as INT32 codes (4 bytes per row) of ONE order-
 *     preserving dictionary per plan, built by the caller over every STRING column it scans
 *     and every STRING constant of the plan (SSGPU_EXPR_CONST with dtype SSGPU_STRING carries
 *     the code in i64).  Code order == the reference's StringPiece order (memcmp, then
 *     length), so comparisons, IN / CASE, MIN / MAX / FIRST / LAST, group keys and sort order
 *     on the codes are bit-exact restatements of the same operations on the strings
 *     (types_infrastructure.h:238-246); the bytes themselves never reach the device.
 *     Arithmetic, casts and SUM on STRING are bind errors exactly as in the reference.  The
 *     dictionary is ssgpu_dict_* below (the host mirrors build it through the ABI);
 *   - floating aggregates, where results are not bit-defined by the reference's definition:
 *       SUM over FLOAT / DOUBLE: the reference folds sequentially in input order
 *       (aggregation_operators.h:173-186), so its result depends on the row order and is up to
 *       thousands of ULP from the exact sum on ill-conditioned data.  Here scalar and grouped sums
 *       are accumulated with compensation (double-double per lane, TwoSum-compensated atomics per
 *       group) and rounded once: <= 1 ULP from the exact sum (measured 0 ULP against math.fsum,
 *       tests/test_double_sum_gpu.py) and identical to the reference whenever every partial sum is
 *       exactly representable;
 *       SUM of a FLOAT / DOUBLE input into an INTEGER result (AddAggregationWithDefinedOutputType): `*result += val`
 *       adds in the floating type and truncates back after every row -- there the order IS the definition, and the
 *       rows are folded one after the other in input order (materialise -> one thread per group / cluster / the one
 *       wavefront of a ScalarAggregate): bit-identical to the reference, and slow by construction -- 10 - 20 ns per row of the
 *       longest segment, i.e. 1 - 2 s for a ScalarAggregate over 1e8 rows, where every other aggregate of the path takes a
 *       millisecond.  The one-segment fold runs in launches of 2^21 rows and looks at ssgpu_interrupt between them.  Under
 *       max_unique_keys_in_result the folded row's rows are folded in input order across its keys; next to DISTINCT
 *       aggregates (whose shape sorts a group's rows by the values) the rows are sorted back by their input row id first.
 *       Across shards the rows themselves travel (supersonic_amd.distributed.sharded_group_aggregate); the fused drivers refuse;
 *       MIN / MAX over FLOAT / DOUBLE: the reference's update is "if (val < result) result = val"
 *       (aggregation_operators.h:200,221) after ASSIGNING a group's first non-NULL value: a NaN that
 *       comes FIRST stays (nothing is less than NaN), a NaN that comes later is skipped.  Same here:
 *       the kernels skip NaNs and flag the run when one reaches a floating MIN / MAX; such a run is
 *       repeated ONCE with the plan in its NaN-exact form (a
 *       hidden FIRST of the column, result = IF(IS_NAN(first), first, min)), which the plan then
 *       keeps -- data without NaNs never pays for it.  (The repeat happens inside ssgpu_plan_run under the default
 *       "lazy_feedback" = 0, when the result is first touched under 1: see INPUT LIFETIME below.)  Not covered, and stated: aggregates next to a
 *       DISTINCT aggregate, under max_unique_keys_in_result and across shards keep the
 *       order-independent answer (NaNs skipped).  -0.0 and +0.0 compare equal in the reference, so
 *       which of the two a MIN / MAX returns is order-dependent there; here -0.0 < +0.0;
 *       MIN / MAX from one integer type into another (AddAggregationWithDefinedOutputType): the
 *       reference compares every value in its own type with the running result and stores the cast
 *       (aggregation_operators.h:187-228), so MAX of INT32 {0, 1, -1} into UINT32 is 1
 *       (aggregation_operators_test.cc:200-210) -- but a value the result type cannot hold makes its
 *       result depend on the row order ({-1, 0, 1} gives 4294967295).  Here the extremum is taken in
 *       the input's type and cast once: the same result whenever the values fit the result type, and
 *       an order-independent one where they do not;
 *   - every *_create has a *_destroy; one ctx/plan is driven by one host
 *     thread at a time (as the reference); only ssgpu_interrupt is callable
 *     concurrently;
 *   - there is NO CPU execution path behind this ABI: running a plan without
 *     a usable gfx950 device fails with SSGPU_ERROR_NO_DEVICE.
 */
#ifndef SSGPU_H_
#define SSGPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSGPU_ABI_VERSION 10

/* ---- reference enum values (supersonic/proto/supersonic.proto) ---------- */
enum {
  SSGPU_INT32 = 1, SSGPU_INT64 = 2, SSGPU_UINT64 = 3, SSGPU_DATETIME = 4,
  SSGPU_DOUBLE = 5, SSGPU_BOOL = 6, SSGPU_UINT32 = 8, SSGPU_FLOAT = 9,
  SSGPU_DATE = 10, SSGPU_STRING = 0, SSGPU_BINARY = 7
};
enum { SSGPU_NOT_NULLABLE = 0, SSGPU_NULLABLE = 1 };
enum {
  SSGPU_SUM = 0, SSGPU_MIN = 1, SSGPU_MAX = 2, SSGPU_COUNT = 3,
  SSGPU_CONCAT = 4, SSGPU_FIRST = 5, SSGPU_LAST = 6,
  /* not a reference aggregation: the exact residual of the DOUBLE SUM of the same input column in the same GroupAggregate /
   * AggregateClusters specification (which must precede it) -- the group's sum is accumulated in double-double, SUM emits
   * s = fl(hi + lo) and SUM_RESIDUAL emits e with s + e == hi + lo exactly.  What a sharded run sends next to its partial
   * sums so that the cross-shard total stays within 1 ULP of the exact sum (supersonic_amd/distributed.py). */
  SSGPU_SUM_RESIDUAL = 100
};
enum { SSGPU_ASCENDING = 0, SSGPU_DESCENDING = 1 };
enum {
  SSGPU_OK = 0,
  SSGPU_ERROR_UNKNOWN = 100,
  SSGPU_ERROR_GENERAL_IO_ERROR = 101,   /* file_io.cc: every FileInput / FileOutput failure */
  SSGPU_ERROR_MEMORY_EXCEEDED = 102,
  SSGPU_ERROR_NOT_IMPLEMENTED = 103,
  SSGPU_ERROR_EVALUATION_ERROR = 104,
  SSGPU_ERROR_TOO_MANY_ROWS = 302,
  SSGPU_ERROR_ATTRIBUTE_COUNT_MISMATCH = 401,
  SSGPU_ERROR_ATTRIBUTE_TYPE_MISMATCH = 402,
  SSGPU_ERROR_ATTRIBUTE_MISSING = 403,
  SSGPU_ERROR_ATTRIBUTE_EXISTS = 404,
  SSGPU_ERROR_INVALID_ARGUMENT_TYPE = 405,
  SSGPU_ERROR_INVALID_ARGUMENT_VALUE = 407,
  SSGPU_ERROR_ATTRIBUTE_AMBIGUOUS = 408,
  SSGPU_INTERRUPTED = 1000,
  /* not a reference code: no gfx950 device / HIP runtime failure */
  SSGPU_ERROR_NO_DEVICE = 2000,
  SSGPU_ERROR_HIP = 2001
};

/* ---- opaque handles ----------------------------------------------------- */
typedef struct ssgpu_ctx ssgpu_ctx;
typedef struct ssgpu_plan ssgpu_plan;
typedef struct ssgpu_block ssgpu_block;
typedef struct ssgpu_result ssgpu_result;

/* ---- schema (TupleSchema / Attribute, tuple_schema.h:77,126) ------------ */
typedef struct ssgpu_attr {
  const char* name;
  int32_t dtype;    /* DataType value */
  int32_t nullable; /* Nullability value */
} ssgpu_attr;

/* ---- symbolic expression tree (Expression, expression/base/expression.h) --
 * Nodes are stored in one array; children are referenced by index and must
 * precede their parent.  Variable-length child lists live in `expr_args`. */
enum {
  SSGPU_EXPR_ATTR_NAMED = 1, /* NamedAttribute(name)  projecting_expressions.h */
  SSGPU_EXPR_ATTR_AT = 2,    /* AttributeAt(i64)                                */
  SSGPU_EXPR_CONST = 3,      /* ConstInt64(..) etc: dtype + i64/f64 payload     */
  SSGPU_EXPR_NULL = 4,       /* Null(dtype)                                     */
  SSGPU_EXPR_OP = 5,         /* operator `op` (OperatorId) over children        */
  SSGPU_EXPR_ALIAS = 6,      /* Alias(name, child)                              */
  SSGPU_EXPR_COMPOUND = 7,   /* CompoundExpression: children (+ALIAS children)  */
  SSGPU_EXPR_CAST = 8        /* CastTo(dtype, child) (explicit, quiet)          */
};
/* Operators that are not a single OperatorId in the reference but are public
 * factory functions (comparison_expressions.h): expressed with these pseudo
 * ids; the binder rewrites them exactly as the reference does
 * (comparison_bound_expressions.cc:832-848: a > b  ==  Less(b, a)). */
enum { SSGPU_OP_GREATER = 100001, SSGPU_OP_GREATER_OR_EQUAL = 100002,
       /* NullingIf(cond, then, otherwise) (elementary_expressions.h:55-61): OPERATOR_IF whose NULL condition
        * gives a NULL result instead of taking the OTHERWISE branch */
       SSGPU_OP_NULLING_IF = 100003,
       /* RoundWithPrecision(x, precision) (math_expressions.h): binds as OPERATOR_ROUND_WITH_MULTIPLIER(x,
        * POW(10.0, precision)) after checking that the precision is an integer (math_bound_expressions.cc:341-382) */
       SSGPU_OP_ROUND_WITH_PRECISION = 100004 };

typedef struct ssgpu_expr {
  int32_t kind;      /* SSGPU_EXPR_* */
  int32_t op;        /* OperatorId for SSGPU_EXPR_OP */
  int32_t dtype;     /* CONST / NULL / CAST target type */
  int32_t first_arg; /* index into expr_args */
  int32_t nargs;
  int32_t reserved;
  int64_t i64;       /* integer / bool / date payload, or attribute position */
  double f64;        /* FLOAT / DOUBLE payload */
  const char* name;  /* attribute name or alias */
} ssgpu_expr;

/* ---- projector (SingleSourceProjector, base/infrastructure/projector.h) - */
enum {
  SSGPU_PROJ_ALL = 1,      /* ProjectAllAttributes()            */
  SSGPU_PROJ_NAMED = 2,    /* ProjectNamedAttribute(name)       */
  SSGPU_PROJ_AT = 3,       /* ProjectAttributeAt(position)      */
  SSGPU_PROJ_NAMED_AS = 4  /* ProjectNamedAttributeAs(name,alias) */
};
typedef struct ssgpu_proj {
  int32_t kind;
  int32_t position;
  const char* name;
  const char* alias;   /* NAMED_AS: new name; ALL: optional name prefix (ProjectAllAttributes("L.")) */
  int32_t source;      /* MultiSourceProjector entries (HASH_JOIN result): 0 = lhs, 1 = rhs */
  int32_t reserved;
} ssgpu_proj;

/* ---- AggregationSpecification::Element (cursor/core/aggregate.h:28-80) -- */
typedef struct ssgpu_agg {
  int32_t aggregation; /* Aggregation value */
  int32_t distinct;    /* 0/1 */
  int32_t output_type; /* DataType, or -1 when not specified */
  int32_t reserved;
  const char* input;   /* "" for COUNT(*) */
  const char* output;
} ssgpu_agg;

/* ---- SortOrder element (cursor/infrastructure/ordering.h:48-101) -------- */
typedef struct ssgpu_sortkey {
  const char* name;
  int32_t order; /* ColumnOrder value */
  int32_t reserved;
} ssgpu_sortkey;

/* ---- operation tree (Operation factories, supersonic/supersonic.h) ------ */
enum {
  SSGPU_OP_SCAN = 1,               /* ScanView(view)            scan_view.h   */
  SSGPU_OP_COMPUTE = 2,            /* Compute(expr, child)      compute.h     */
  SSGPU_OP_FILTER = 3,             /* Filter(pred, proj, child) filter.h      */
  SSGPU_OP_PROJECT = 4,            /* Project(proj, child)      project.h     */
  SSGPU_OP_SCALAR_AGGREGATE = 5,   /* ScalarAggregate(spec, child) aggregate.h:341 */
  SSGPU_OP_GROUP_AGGREGATE = 6,    /* GroupAggregate(keys, spec, opts, child) aggregate.h:224 */
  SSGPU_OP_AGGREGATE_CLUSTERS = 7, /* AggregateClusters(keys, spec, child) aggregate.h:285 */
  SSGPU_OP_SORT = 8,               /* Sort(order, proj, mem_limit, child) sort.h:83 */
  SSGPU_OP_BEST_EFFORT_GROUP_AGGREGATE = 10, /* BestEffortGroupAggregate(keys, spec, opts, child) aggregate.h:246-250; option0 =
                                      GroupAggregateOptions::memory_quota in bytes (0 = none): the result block holds quota / bytes of a
                                      result row groups; run with ssgpu_plan_run_best_effort (ABI 9) */
  SSGPU_OP_HASH_JOIN = 9           /* HashJoinOperation(type, lhs keys, rhs keys, result projector,
                                      uniqueness, lhs, rhs) hash_join.h:37-56.  `child` = lhs chain,
                                      `child2` = the rhs op, which must be a SCAN of the auxiliary
                                      input (a device-resident table).  UNIQUE rhs keys: the probe
                                      and the gathers of the rhs columns are fused into the lhs
                                      pipeline (a repeated key is reported at run time); NOT_UNIQUE:
                                      rows multiply -- the lhs pipeline materialises its side and an
                                      expand stage emits one row per match (lhs order, rhs order) */
};
enum { SSGPU_JOIN_INNER = 0, SSGPU_JOIN_LEFT_OUTER = 1 };           /* JoinType, supersonic.proto:108-113 */
enum { SSGPU_KEYS_NOT_UNIQUE = 0, SSGPU_KEYS_UNIQUE = 1 };            /* KeyUniqueness, :115-118 */
typedef struct ssgpu_op {
  int32_t kind;       /* SSGPU_OP_* */
  int32_t child;      /* index of the child op (must precede), -1 for SCAN */
  int32_t expr;       /* COMPUTE: expression root; FILTER: predicate; else -1 */
  int32_t proj_first; /* FILTER/PROJECT/SORT: result projector;               */
  int32_t proj_n;     /*   GROUP_AGGREGATE/AGGREGATE_CLUSTERS: key projector   */
  int32_t agg_first;
  int32_t agg_n;
  int32_t sort_first;
  int32_t sort_n;
  int32_t child2;     /* HASH_JOIN: index of the rhs op (must precede); else unused */
  int64_t option0;    /* GROUP: max_unique_keys_in_result (0 = no limit, n > 0 = limit n, -1 = limit 0; aggregate.h:160-205);
                         DISTINCT aggregates under it keep ONE seen-value set per result row, the folded last row included
                         (column_aggregator.cc:308-376), and CONCAT joins a result row's values in input order over all its keys;
                         SORT: memory limit (ignored: no spill path);
                         SCAN: input index (0 = the plan input, 1 = the auxiliary input);
                         HASH_JOIN: JoinType | KeyUniqueness << 8             */
  int32_t proj2_first; /* HASH_JOIN: rhs key selector (proj_first/proj_n = lhs key selector) */
  int32_t proj2_n;
  int32_t proj3_first; /* HASH_JOIN: result projector (entries tagged with `source`) */
  int32_t proj3_n;
} ssgpu_op;

typedef struct ssgpu_plan_desc {
  const ssgpu_attr* input_schema; /* schema of the scanned View/Block */
  int32_t n_attrs;
  const ssgpu_op* ops;
  int32_t n_ops;                  /* root = ops[n_ops-1] */
  const ssgpu_expr* exprs;
  int32_t n_exprs;
  const int32_t* expr_args;
  int32_t n_expr_args;
  const ssgpu_proj* projs;
  int32_t n_projs;
  const ssgpu_agg* aggs;
  int32_t n_aggs;
  const ssgpu_sortkey* sortkeys;
  int32_t n_sortkeys;
  const ssgpu_attr* aux_schema;   /* schema of the auxiliary input (rhs of a HASH_JOIN), or NULL */
  int32_t n_aux_attrs;
} ssgpu_plan_desc;

/* ---- a column of a View (base/infrastructure/block.h:55-192) ------------ */
typedef struct ssgpu_column {
  const void* data;       /* DEVICE pointer to rows * sizeof(type) bytes     */
  const uint8_t* is_null; /* DEVICE pointer to `rows` bytes, or NULL          */
} ssgpu_column;

/* ---- context ------------------------------------------------------------ */
/* device_id >= 0: bind that HIP device.  device_id == -1: bind-only context
 * (plans can be created/inspected, running fails with SSGPU_ERROR_NO_DEVICE);
 * this is what lets schema/bind errors be tested on a machine without a GPU. */
int ssgpu_ctx_create(int device_id, ssgpu_ctx** out);
void ssgpu_ctx_destroy(ssgpu_ctx* ctx);
const char* ssgpu_last_error(const ssgpu_ctx* ctx);
int ssgpu_ctx_has_device(const ssgpu_ctx* ctx);   /* 0 for a bind-only context */
int ssgpu_abi_version(void);
/* The HIP stream (hipStream_t) all kernels of this ctx are launched on, and
 * the side stream used for host<->device staging of blocks. */
void* ssgpu_ctx_stream(ssgpu_ctx* ctx);
void* ssgpu_ctx_copy_stream(ssgpu_ctx* ctx);
/* Use a caller-owned stream (e.g. torch's current stream) for kernels. */
int ssgpu_ctx_set_stream(ssgpu_ctx* ctx, void* hip_stream);
int ssgpu_ctx_synchronize(ssgpu_ctx* ctx);
/* Tuning knobs; an unknown key is ERROR_INVALID_ARGUMENT_VALUE.  None of them changes a result.
 *   shape of the tile VM:   tile_rows (0 = by the program's LDS need, else 512 / 1024 / 2048), lds_target_bytes, wgs_per_cu,
 *                           grid_limit
 *   runtime specialisation: specialize (1 = plans created afterwards run kernels compiled for them, see ssgpu_plan_specialize;
 *                           0 = only plans that call ssgpu_plan_specialize; default 0)
 *   Filter:                 filter_single_pass (1 = decoupled look-back instead of count + store passes)
 *   GroupAggregate:         group_capacity (initial table), group_local (0 = no LDS table in front of the global one),
 *                           group_partition (0 never / 1 by run feedback / 2 always hash-partitioned), group_slab (0 never /
 *                           1 by estimate / 2 always the one-table-per-CU form), group_resident (0 = the one-table-per-CU form always
 *                           through scatter + aggregation, never straight from the input columns), group_scout (0 = no scout run --
 *                           the direct shape over a 1/64 prefix, result discarded -- ahead of a plan's first run over >= group_scout_rows rows; group_scout_rows: 8 M by default, a plan that is run ONCE over fewer rows may lower it --
 *                           an eighth of a smaller input is scouted --, trading a faster first run for a coarser group-count estimate), part_plain (0 = the partition scatter always as
 *                           a VM program, never as its own kernel), part_n, part_wgs_per_cu, part_lds_target, part_agg_lds,
 *                           part_rec_align, lazy_feedback (0 = a GroupAggregate reads its overflow / feedback words at the end of EVERY run --
 *                           one stream synchronise per run -- instead of leaving them to the next touch of the result; see ssgpu_plan_run)
 *                           plain partition scatter (ABI 10): pscat_pipe (0 = the tile-after-tile loop instead of the software pipeline of the
 *                           specialised build), pscat_threads / pscat_rows / pscat_wgs (launch shape: 512 / 1024 threads, rows per thread,
 *                           workgroups per CU; 0 = 1024 x 2 x 1), part_split (1 = dense partitions' records as payload words + 16-bit table
 *                           entries), part_prefetch (0 = the partition aggregation loads its records in the trip that uses them),
 *                           part_overlap / part_overlap_rows (> 1: a dense run over at least that many rows takes its input in that many
 *                           row ranges, range k aggregated on a side stream beside the scatter of range k + 1; measured slower, default 1)
 *   ScalarAggregate:        fuse_emit (0 = the result row is emitted by a launch of its own instead of the finish launch)
 *   results:                out_arena (0 = one allocation per column of a stage's large result instead of one arena with skewed bases)
 *   stage hand-off:         async_handoff (0 = the row count of every intermediate result is read on the host before the next stage is launched;
 *                           default 1: a filter-less Compute / Project stage takes it from the device, no stream synchronise in between)
 *   Sort:                   sort_records (0 = gather payload column by column), sort_hybrid (0 = all eight digits),
 *                           sort_hi_digits (2..4, 0 = by row count), sort_compact (0 = (key, row id) pairs instead of one
 *                           (high half | row id) word)
 *   measurement:            profile, profile_total (HIP events around the stage kernels / the run: ssgpu_plan_counters,
 *                           ssgpu_plan_recent_kernel_ms), debug_timing
 *   development only (results may be WRONG): part_scatter_debug, part_agg_debug */
int ssgpu_ctx_set_option(ssgpu_ctx* ctx, const char* key, int64_t value);

/* ---- pinned host memory (BufferAllocator seam, memory.h:100-233) -------- */
int ssgpu_host_alloc(ssgpu_ctx* ctx, size_t bytes, void** out);
void ssgpu_host_free(ssgpu_ctx* ctx, void* p);

/* ---- BufferAllocator / MemoryLimit (base/memory/memory.h:100-233,465-520) -----------------------
 * An allocator of DMA-able (pinned, 256-byte aligned) host memory with an optional soft quota, shaped like the
 * reference's: Allocate(requested) = BestEffortAllocate(requested, requested); a request the quota cannot serve
 * returns SSGPU_ERROR_MEMORY_EXCEEDED (102) and *out = NULL (the reference's callers turn a NULL Buffer into
 * ERROR_MEMORY_EXCEEDED); zero-size requests succeed with a non-NULL pointer (memory.h:112-117).  quota < 0 =
 * unlimited.  On a bind-only context (no device) the memory is ordinary aligned host memory.  Not thread-safe (as
 * the reference's allocators other than ThreadSafe*). */
typedef struct ssgpu_allocator ssgpu_allocator;
int ssgpu_allocator_create(ssgpu_ctx* ctx, int64_t quota_bytes, ssgpu_allocator** out);
void ssgpu_allocator_destroy(ssgpu_allocator* a);
/* grants between `minimal` and `requested` bytes (as many as the quota leaves); *granted may be NULL */
int ssgpu_allocator_allocate(ssgpu_allocator* a, size_t requested, size_t minimal, void** out, size_t* granted);
/* BufferAllocator::Reallocate: contents preserved up to the smaller size; on failure the old buffer stays valid.  Like the
 * reference's mediator the quota is checked as if the new buffer had to exist next to the old one (memory_test.cc:183-198) */
int ssgpu_allocator_reallocate(ssgpu_allocator* a, void* p, size_t requested, size_t minimal, void** out, size_t* granted);
void ssgpu_allocator_free(ssgpu_allocator* a, void* p);
int64_t ssgpu_allocator_available(const ssgpu_allocator* a); /* bytes the quota still allows; INT64_MAX if unlimited */
int64_t ssgpu_allocator_allocated(const ssgpu_allocator* a);

/* ---- STRING columns: order-preserving dictionary ----------------------------------------------------
 * STRING values cross the ABI as INT32 codes (see Conventions).  The dictionary is built here, behind the ABI, so
 * that every host mirror (and every shard of a multi-GPU run that builds it over the same strings) gets the same
 * codes: code = rank of the byte string in the reference's StringPiece order (memcmp, then length,
 * types_infrastructure.h:238-246) among the DISTINCT strings given.  Strings are (pointer, length) pairs, as
 * StringPiece; the dictionary copies the bytes (the Arena deep-copy rule, cursor/core/filter.cc:205-230). */
typedef struct ssgpu_dict ssgpu_dict;
int ssgpu_dict_create(const char* const* strings, const int32_t* lengths, int64_t n, ssgpu_dict** out);
void ssgpu_dict_destroy(ssgpu_dict* d);
int32_t ssgpu_dict_size(const ssgpu_dict* d);
/* codes[i] = code of strings[i], or -1 if it is not in the dictionary (is_null rows, if given, get code 0) */
int ssgpu_dict_encode(const ssgpu_dict* d, const char* const* strings, const int32_t* lengths, const uint8_t* is_null,
                      int64_t n, int32_t* codes);
int ssgpu_dict_decode(const ssgpu_dict* d, int32_t code, const char** bytes, int32_t* length);

/* CONCAT aggregates (column_aggregator.cc:496-505) produce strings that are in no dictionary yet.  The device orders the
 * values (materialise, stable sort by the group keys) and counts them; the strings are printed on the host -- PrintTyped
 * forms: integers in decimal, BOOL as TRUE / FALSE, FLOAT / DOUBLE as SimpleFtoa / SimpleDtoa, STRING as is -- when the
 * column is fetched, joined with ',' in input order, and become a dictionary OWNED BY THE RESULT:
 * ssgpu_result_column_dict(result, col) is that dictionary for a CONCAT column (its INT32 cells are codes of it; valid
 * until the plan runs again) and NULL for every other column (STRING cells of those are codes of the plan's dictionary).
 * ssgpu_plan_set_dict hands the plan the dictionary its STRING columns were encoded with (borrowed; CONCAT of a STRING
 * column prints through it).  DISTINCT CONCAT prints a value once per result row, at its first
 * occurrence.  DATE / DATETIME print as the reference's strftime forms ("%Y/%m/%d", "%Y/%m/%d-%H:%M:%S" of gmtime,
 * types_infrastructure.cc:92-114).  Limits, refused at bind: BINARY inputs, a CONCAT result
 * that feeds another operation.  (Across shards the rows travel: sharded_group_aggregate.  Next to DISTINCT aggregates the rows are sorted back into input order first.) */
int ssgpu_plan_set_dict(ssgpu_plan* plan, const ssgpu_dict* dict);
const ssgpu_dict* ssgpu_result_column_dict(ssgpu_result* r, int32_t col);

/* ---- device-resident Block ----------------------------------------------- */
/* Layout (base/infrastructure/block.cc:20-36 allocates a buffer per column; so did this library until ABI 9): a block of 32 MiB or
 * more is ONE device allocation, column i's data starting i x (its 2 MiB-rounded size + 512 bytes) into it and the NULL masks
 * behind the data.  A pipeline reads the same rows of all its columns at once; columns that are allocated one by one start at
 * bases congruent modulo every power of two the allocator aligns to, and those reads meet on the same HBM channels.  Measured on
 * the 8-column headline query, same box, alternating: one allocation per column 0.786 of 8 TB/s, the arena 0.816 - 0.820
 * (profiles/r06_layout_ab.txt).  Callers that bring their own device columns
 * (ssgpu_plan_run) get what their layout gives; ssgpu_block_create + ssgpu_block_column is how to get this one. */
int ssgpu_block_create(ssgpu_ctx* ctx, const ssgpu_attr* schema, int32_t n_attrs,
                       int64_t row_capacity, ssgpu_block** out);
void ssgpu_block_destroy(ssgpu_block* b);
/* Async H2D on the copy stream; `is_null` may be NULL for NOT_NULLABLE columns.
 * Source should be pinned (ssgpu_host_alloc) for the copy to overlap.  The next run of ANY plan of the context (ssgpu_plan_run with
 * ssgpu_block_column's pointers, ssgpu_expr_evaluate, ssgpu_plan_run_block ...) is ordered behind the uploads issued before it: the
 * caller needs no synchronise between upload and run (ABI 10; until then only ssgpu_plan_run_block waited for them). */
int ssgpu_block_upload(ssgpu_block* b, int32_t col, const void* host_data,
                       const uint8_t* host_is_null, int64_t row_offset, int64_t rows);
int ssgpu_block_set_row_count(ssgpu_block* b, int64_t rows);
int64_t ssgpu_block_row_count(const ssgpu_block* b);
/* Device pointers of column `col` (valid while the block lives). */
int ssgpu_block_column(const ssgpu_block* b, int32_t col, ssgpu_column* out);

/* ---- View file format (cursor/infrastructure/file_io.cc:176-193,377-440) ------
 * The reference's only on-disk block format: a sequence of chunks (<= 8192 rows, :70), each
 *   uint64 row_count, then per column  [row_count bytes of bool is_null, if the attribute is
 *   NULLABLE]  [row_count * sizeof(type) bytes of data]   (fixed-width types; the schema is
 * not stored).  ssgpu_block_create_from_file = FileInput(schema, file, ...) drained into a
 * device Block: chunks are read into pinned staging buffers and copied on the copy stream
 * while the next chunk is being read.  ssgpu_result_write_file = FileOutput(file)->Write(view)
 * for a finished result (or any block via ssgpu_block_write_file). */
int ssgpu_block_create_from_file(ssgpu_ctx* ctx, const ssgpu_attr* schema, int32_t n_attrs,
                                 const char* path, ssgpu_block** out);
int ssgpu_block_write_file(ssgpu_block* b, const char* path);
int ssgpu_result_write_file(ssgpu_result* r, const char* path);

/* ---- plan: bind + lower --------------------------------------------------- */
int ssgpu_plan_create(ssgpu_ctx* ctx, const ssgpu_plan_desc* desc, ssgpu_plan** out);
void ssgpu_plan_destroy(ssgpu_plan* plan);
int32_t ssgpu_plan_attr_count(const ssgpu_plan* plan);
int ssgpu_plan_attr(const ssgpu_plan* plan, int32_t i, ssgpu_attr* out);
/* Human-readable bound tree + lowered pipeline stages + VM disassembly. */
const char* ssgpu_plan_describe(ssgpu_plan* plan);
/* Debug/test hook: raw VM program of pipeline stage `stage` (see csrc/vm.h). */
int ssgpu_plan_program(const ssgpu_plan* plan, int32_t stage, const void** instrs,
                       int32_t* n_instrs, int32_t* instr_bytes);

/* Device memory a plan may hold (its stages' output, scratch and table buffers): a soft quota in the sense of
 * MemoryLimit (memory.h:465).  A run that would need more fails with SSGPU_ERROR_MEMORY_EXCEEDED -- what a cursor
 * of the reference does when its Operation was given a MemoryLimit allocator (operation.h:66-76,
 * aggregate_groups.cc:372-402).  bytes < 0 = unlimited (default). */
int ssgpu_plan_set_memory_limit(ssgpu_plan* plan, int64_t bytes);
/* Per-plan kernel specialisation by runtime compilation (hiprtc).  ssgpu_plan_specialize(plan) -- or the context option
 * "specialize" = 1 at the time the plan is created -- makes the plan run kernels compiled for it: the same handlers as
 * the interpreting kernel with the opcode dispatch and operand offsets folded away, cached by program and shared by all
 * plans with the same program; results are identical either way.  WHEN code is compiled is explicit: inside
 * ssgpu_plan_specialize for every stage whose launch shape is known without a run (scalar aggregates, materialising
 * stages, clustered aggregation), otherwise at the first launch of a kernel shape (the plan's first run; a partitioned
 * GroupAggregate has up to three kernels -- the stage's program, its partition-scatter program and the
 * partition-aggregation kernel -- whose shape its first runs settle).  A plan that never asked never compiles: no run
 * of it blocks on a compiler, and no code is loaded behind its back.  A compilation (~2 s + 0.4 s per VM instruction)
 * does not hold any library lock: other plans run and compile meanwhile; two plans wanting the same kernel share one
 * compilation.  The modules are reference-counted: ssgpu_plan_destroy drops the plan's references; a kernel without a user
 * stays loaded while it is among the 8 most recently released ones (the next plan with the same program finds it instead
 * of compiling for seconds) and is unloaded beyond that; ssgpu_specialized_kernels_trim(keep) unloads idle kernels down to
 * `keep` at once (ssgpu_memory_stats shows modules and code bytes currently loaded).
 * Compiled code objects are also kept ON DISK (ABI 6), named by a hash of everything the compilation depended on (the
 * library's kernel sources, the plan's program / descriptors, the compiler options, the architecture, the HIP runtime
 * version): a second process loads them in milliseconds instead of compiling for seconds (ssgpu_memory_stats:
 * rtc_disk_hits vs rtc_compilations).  Directory: $SSGPU_RTC_CACHE_DIR (empty string = no disk cache), else
 * $XDG_CACHE_HOME/ssgpu/rtc, else $HOME/.cache/ssgpu/rtc; files are written atomically and checksummed.  The directory is bounded:
 * after a store the least recently used code objects go until it is below 3/4 of $SSGPU_RTC_CACHE_MAX_MB (default 512; <= 0: no limit).
 * "specialize" = 2: a plan runs the compiled kernel of a stage where one EXISTS -- loaded in this process, or in the on-disk
 * cache from any earlier process -- and the interpreting kernel where none does; it never compiles.
 * THE DEFAULT POLICY ("specialize" = 3; process-wide default: environment SSGPU_SPECIALIZE): like 2, and a kernel that is
 * MISSING when a run over at least "specialize_min_rows" (default 2^22) input rows wants it is compiled IN THE BACKGROUND by the
 * library's one worker thread.  No run ever waits for a compiler: that run and the next ones use the interpreting kernel; the
 * plan asks again (at most every 20 ms) and switches when the kernel is there -- ssgpu_plan_specialized counts it from then on,
 * ssgpu_plan_specialize_reason says "being compiled ..." until then -- and every later plan and, through the disk cache, every
 * later process finds it at once.  A cold process therefore runs its first seconds interpreted (the headline query: 0.70 of the
 * roofline instead of 0.80) and a service reaches the compiled kernels without anybody calling ssgpu_plan_specialize.  Small
 * runs never start a compilation.  The worker compiles one kernel at a time, oldest request first; at process exit queued
 * requests are dropped and a compilation in flight is waited for (seconds).  ssgpu_plan_specialize on a plan of policy 2 / 3
 * compiles what it did not find -- now, waiting (for the worker too, if it is at that kernel).  "specialize" = 0: the
 * interpreting kernels only (unless the plan asks).
 * A stage whose specialisation is not possible (no libhiprtc on the host, a compilation failure) keeps the
 * interpreting kernel -- still the HIP path -- and ssgpu_plan_specialize_reason says why ("" if nothing was refused).
 * ssgpu_plan_specialized: how many specialised kernels the plan currently holds. */
int ssgpu_plan_specialize(ssgpu_plan* plan);
void ssgpu_specialized_kernels_trim(int32_t keep);
int32_t ssgpu_plan_specialized(const ssgpu_plan* plan);
const char* ssgpu_plan_specialize_reason(const ssgpu_plan* plan);
/* What the library holds in this process, right now: device and pinned-host bytes of every live buffer (blocks, plans'
 * scratch / tables / outputs, results), live plans / blocks / HIP events, and the specialised kernels' modules.  The
 * per-process measure behind "repeated runs of a plan do not grow memory" (the reference watches its allocator the
 * same way, testing/expression_test_helper.cc:213-245). */
typedef struct ssgpu_memory_stats_t {
  int64_t device_bytes, pinned_bytes;
  int64_t live_plans, live_blocks, events;
  int64_t rtc_modules, rtc_code_bytes, rtc_compilations;   /* loaded now / loaded now / hiprtc compilations so far */
  int64_t rtc_disk_hits;   /* specialised kernels loaded from the on-disk cache of code objects instead of being compiled (ABI 6) */
} ssgpu_memory_stats_t;
int ssgpu_memory_stats(ssgpu_memory_stats_t* out);
/* Device blocks of destroyed plans wait in a bounded per-process pool (SSGPU_POOL_MB, default 4 GiB; still counted in
 * device_bytes) for the next plan of the same device.  The library gives them back to the driver by itself when an allocation
 * fails for lack of memory and when the process's last device context is destroyed; ssgpu_pool_trim does it on request
 * (device < 0: every device) and returns the bytes freed (ABI 9).  The reference's counterpart is MemoryLimit / the soft
 * quota's release on cursor destruction (base/memory/memory.h:420-520). */
int64_t ssgpu_pool_trim(int32_t device);
int64_t ssgpu_plan_memory_in_use(const ssgpu_plan* plan);

/* What the last run of a plan's stage did -- the execution shape run feedback chose.  For tests and tuning: a parity
 * test that claims to cover the partitioned GroupAggregate or the hybrid Sort asserts here that it ran. */
typedef struct ssgpu_stage_info {
  int32_t kind;             /* 1 scalar aggregate, 2 materialise, 3 group aggregate, 4 sort, 5 clustered aggregate, 6 join expansion */
  int32_t group_shape;      /* group aggregate: 0 direct (LDS table + global table), 1 hash partitions, 2 slab (one LDS table of all groups per
                               aggregation workgroup, records through memory), 3 resident (the same tables fed straight from the input
                               columns of a plain stage: no scatter, no records) */
  int32_t part_n;           /* hash partitions of the partitioned shape */
  int32_t part_seg_growth;  /* x4 per segment overflow (skewed keys) */
  int32_t group_wgs_per_cu; /* direct shape: resident workgroups per CU the LDS table is sized for */
  int32_t reruns;           /* attempts the last run needed beyond the first (regrown table, segments or partitions) */
  int32_t sort_passes;      /* radix passes of the last run */
  int32_t sort_mode;        /* 0 LSD over the varying digits, 1 high digits + tie fix-up, 2 one-word (high half | row id) keys;
                               +16: tie runs were too long and all digits were sorted after all */
  int32_t specialized;      /* bit 0 the stage's program, bit 1 the partition-scatter program, bit 2 the partition aggregation,
                               bit 3 the plain partition scatter, bit 4 the resident group aggregation */
  int32_t plain_scatter;    /* the partition scatter ran as its own kernel over (partition, XCD) segments, not as a VM program */
  int32_t hot_keys;         /* heavy-hitter keys the stage aggregates apart from the hash partitions (found when a segment overflowed; ABI 6) */
  int32_t dense_slots;      /* > 0: the group tables are indexed by the key columns' value ranges (dense slots, ABI 7) -- this many slots; group_shape
                               then says how they are filled: 1 partitions of slot ranges, 3 one table fed from the input columns */
  int32_t split_records;    /* dense partitions: the last run's records left the scatter as payload words + 16-bit table entries (no index word) */
  int32_t row_ranges;       /* dense partitions: row ranges the last run took its input in -- range k is aggregated on a side stream while range k + 1 is scattered (1: one range, no overlap) */
  int32_t reserved[2];
} ssgpu_stage_info;
int32_t ssgpu_plan_stage_count(const ssgpu_plan* plan);
int ssgpu_plan_stage_info(const ssgpu_plan* plan, int32_t stage, ssgpu_stage_info* out);

/* ---- standalone expression seam --------------------------------------------------------------------------
 * Expression::Bind(schema, ...) -> BoundExpressionTree, BoundExpressionTree::Evaluate(view) -> result View
 * (expression/base/expression.h:46-167, :116): the expression tree `root` (nodes as in ssgpu_plan_desc) is bound
 * against `schema` exactly like a Compute over a scan (type promotion, names, nullability, bind errors); the bound
 * tree is an ssgpu_plan whose schema (ssgpu_plan_attr) is the tree's result_schema.  max_row_count is the tree's
 * row_capacity(): Evaluate over more rows fails with SSGPU_ERROR_TOO_MANY_ROWS (expression.cc:57-66); <= 0 means no
 * bound (device buffers grow to the View).  Evaluate returns one result row per input row, in order. */
int ssgpu_expr_bind(ssgpu_ctx* ctx, const ssgpu_attr* schema, int32_t n_attrs, const ssgpu_expr* exprs, int32_t n_exprs,
                    const int32_t* expr_args, int32_t n_expr_args, int32_t root, int64_t max_row_count, ssgpu_plan** out);
int64_t ssgpu_expr_row_capacity(const ssgpu_plan* bound);
int ssgpu_expr_evaluate(ssgpu_plan* bound, const ssgpu_column* cols, int32_t n_cols, int64_t rows, ssgpu_result** out);
/* The node-level form, BoundExpression::DoEvaluate(const View& input, const BoolView& skip_vectors) (expression.h:46-92): `skip` holds
 * one DEVICE byte vector of `rows` bytes per result attribute (n_skip == the bound tree's attribute count; a NULL entry = nothing is
 * skipped), in and out as in the reference: a row whose byte is set on entry is not evaluated -- its result is NULL and a signalling
 * operator (DivideSignaling, SqrtSignaling ...) does not fail on it -- and on return the vector holds the result column's NULLs,
 * entry skips included.  The tree is re-bound once, on first use, as IF($skip_i, NULL, e_i) per attribute (guarded evaluation IS the
 * skip-vector contract).  Nested compound expressions and variable-length results: SSGPU_ERROR_NOT_IMPLEMENTED (ABI 10). */
int ssgpu_expr_evaluate_skip(ssgpu_plan* bound, const ssgpu_column* cols, int32_t n_cols, int64_t rows, uint8_t* const* skip, int32_t n_skip,
                             ssgpu_result** out);

/* ---- run ------------------------------------------------------------------ */
/* cols: one entry per attribute of the plan's input schema, DEVICE pointers.
 * INPUT LIFETIME.  By default (ABI 7; context option "lazy_feedback" = 0) ssgpu_plan_run returns with every decision made that
 * could send the library back to the input columns: a GroupAggregate's overflow / feedback words have been read, a floating
 * MIN / MAX that met a NaN has been repeated in the plan's NaN-exact form.  The run itself is still asynchronous work on the
 * context's stream in general, so the rule is the reference's for ScanView ("the view must outlive the operation",
 * cursor/core/scan_view.h) shortened to the run: the columns passed to ssgpu_plan_run (the block passed to
 * ssgpu_plan_run_block, the auxiliary input) must stay alive and unmodified until ssgpu_plan_run has returned AND the stream
 * has been synchronised (ssgpu_ctx_synchronize, or any result access) -- after that they may be freed or overwritten, and
 * the result is still fetched correctly (tests/test_cursor_contract_gpu.py::test_input_may_be_overwritten_after_a_synchronised_run).
 * "lazy_feedback" = 1 is the opt-in for callers that step a plan without ever waiting for the host (a sharded job's
 * steps: supersonic_amd/distributed.py, include/supersonic_amd/sharded.h, bench.py): a GroupAggregate in its steady state
 * then leaves its feedback words on the stream, and a run that overflowed after all -- or met a NaN -- is repeated FROM THE
 * INPUT COLUMNS when its result is first touched (ssgpu_result_row_count / _column / _device_column / _write_file) or when
 * the plan runs again.  Such a caller keeps the columns of ssgpu_plan_run / _run_partial alive and unmodified until the
 * result has been fetched or the plan has been run again or destroyed.  The option is a property of the PLAN: a plan takes the
 * context's value when it is created, and ssgpu_plan_set_option(plan, "lazy_feedback", v) changes it for that plan alone (ABI 9) --
 * which is what the sharded drivers do, so that other plans of a shared context keep the default contract. */
int ssgpu_plan_set_option(ssgpu_plan* plan, const char* key, int64_t value);
/* BestEffortGroupAggregate (cursor/core/aggregate.h:230-250; GroupAggregateCursor::Next / ProcessInput with best_effort_,
 * aggregate_groups.cc:211-222,332-433), for a plan whose root is SSGPU_OP_BEST_EFFORT_GROUP_AGGREGATE (ABI 9).  ONE call is one
 * ProcessInput: the GroupAggregate over the LONGEST run of input rows that starts at start_row and holds at most `capacity`
 * distinct keys (capacity = option0 bytes / bytes of one result row -- widths + 1 per NULLABLE column, block.cc:20-36 --, at least 1;
 * option0 = 0: no bound) -- the reference's verdict "the result block is full" is its allocator's, deterministic under
 * GuaranteeMemory (:160-163); here it is the block's row capacity.  The rows of the result are key-unique; *next_row is where
 * the next call starts, == rows when the input is exhausted.  The cursor contract on top (both host mirrors): every Next()
 * returns rows of ONE such result, so "rows are key-unique on the group-by key within each returned view" holds, and an input
 * of any size never raises ERROR_MEMORY_EXCEEDED -- a run that does not fit the plan's memory limit is repeated over half the
 * input window with the capacity lowered to match; only a single row that does not fit fails (:405-412).
 * Not available (bind-time ERROR_NOT_IMPLEMENTED): DISTINCT / CONCAT aggregates, floating SUMs into integers, and a
 * BestEffortGroupAggregate below another operation. */
int ssgpu_plan_run_best_effort(ssgpu_plan* plan, const ssgpu_column* cols, int32_t n_cols, int64_t rows, int64_t start_row,
                               int64_t* next_row, ssgpu_result** out);
int ssgpu_plan_run(ssgpu_plan* plan, const ssgpu_column* cols, int32_t n_cols,
                   int64_t rows, ssgpu_result** out);
int ssgpu_plan_run_block(ssgpu_plan* plan, const ssgpu_block* block, ssgpu_result** out);
/* CHUNKED STAGING of a host-resident input (ABI 8).  ssgpu_plan_run / _run_block want the whole input in device memory, and a
 * block's uploads are waited for before the first kernel starts: copy and kernel run one after the other, and an input larger than
 * the device's memory cannot run at all -- where the reference drains its child 1024 rows at a time (aggregate_scalar.cc:53-68).
 * ssgpu_plan_run_host takes HOST columns (`data` / `is_null` are host pointers; pinned memory -- ssgpu_host_alloc -- for the
 * copies to be asynchronous) of any length and moves them through two alternating sets of device columns of `chunk_rows` rows
 * (<= 0: 2^24): chunk k + 1 is copied on the copy stream while the plan reads chunk k; every chunk leaves the partial state of
 * the multi-GPU form, and ONE launch folds the states in row order and emits the row.  Device memory: 2 * chunk_rows rows.
 * Same result as ssgpu_plan_run over the whole input (FIRST / LAST follow the global row order; floating sums are folded chunk
 * by chunk in double-double like the shards of a multi-GPU job; a leading NaN of a floating MIN / MAX is skipped as across
 * shards).  The host columns must stay alive and unmodified until the call has returned AND the context's streams have drained
 * (ssgpu_ctx_synchronize, or fetching the result).
 * Which plans (ABI 10; ssgpu_plan_chunked_form tells, also on a bind-only context):
 *   1  ONE ScalarAggregate stage (over Filter / Compute / Project) -- the path's headline shape: the state fold described above;
 *   2  plans of row-local operations only (Filter / Compute / Project / HashJoin against the auxiliary input; filter.cc:96-128 pulls
 *      its child the same way): every chunk runs the plan, its result rows are appended on the device; the rows keep the input's order;
 *   3  plans whose first blocking operation on the input's path is a GroupAggregate without a key limit, over row-local operations
 *      (aggregate_groups.cc:212-282 ProcessInput): every chunk runs the plan UP TO AND INCLUDING the GroupAggregate -- in whatever
 *      shape its run feedback picks, DOUBLE sums with their SSGPU_SUM_RESIDUAL -- and leaves a partial table; the partial tables are
 *      appended, and at the end ONE second plan runs over them: GroupAggregate of the merge functions (COUNT merges as SUM, FIRST /
 *      LAST in chunk = row order) -> Compute restoring the first one's schema -> the operations above it (Sort, Compute, ...).
 *      A floating MIN / MAX keeps a NaN that is the group's FIRST value (aggregation_operators.h:189-228): the first value travels as a
 *      hidden FIRST next to the NaN-skipping partial result and the merged one is IF(IS_NAN(first), first, min).  DISTINCT / CONCAT
 *      aggregates and the row-after-row SUM of a floating column into an integer are not partial results: SSGPU_ERROR_NOT_IMPLEMENTED
 *      at bind time -- those run over device columns.  Device memory: the staging sets + (groups met per chunk) x chunks partial rows.
 * Everything else (Sort / AggregateClusters / key-limited GroupAggregate as the first blocking operation): SSGPU_ERROR_NOT_IMPLEMENTED. */
int ssgpu_plan_run_host(ssgpu_plan* plan, const ssgpu_column* host_cols, int32_t n_cols, int64_t rows, int64_t chunk_rows, ssgpu_result** out);
/* The PUSH form of the same, for a caller that meets its input the way the reference's cursors do -- a child's Next() handing out
 * Views of <= 1024 rows that are only valid until the next Next() (cursor.h:131-148, aggregate_scalar.cc:53-68):
 *     ssgpu_plan_stream_begin(plan, chunk_rows)            (<= 0: 2^22 rows per staging set)
 *     ssgpu_plan_stream_push(plan, host_cols, n, rows)     any number of times, any number of rows; the rows are COPIED into a pinned
 *                                                          staging set before the call returns (the caller's memory is free again);
 *                                                          a full set is sent to the device and the plan runs over it while the
 *                                                          caller fills the other one
 *     ssgpu_plan_stream_finish(plan, &result)              the partly filled set, the fold of the chunks' states, the result row
 * Same plans, same result as ssgpu_plan_run_host.  A failed push / finish ends the stream; begin on an open stream drops it. */
int ssgpu_plan_stream_begin(ssgpu_plan* plan, int64_t chunk_rows);
int ssgpu_plan_stream_push(ssgpu_plan* plan, const ssgpu_column* host_cols, int32_t n_cols, int64_t rows);
int ssgpu_plan_stream_finish(ssgpu_plan* plan, ssgpu_result** out);
/* The chunked form (1 / 2 / 3 above) ssgpu_plan_run_host / _stream_* would take for this plan, or its refusal's return code; for form 3
 * `head` / `tail` (may be NULL) receive the descriptions (as ssgpu_plan_describe gives them) of the per-chunk plan and of the merging
 * plan, valid while the plan lives.  Needs no device. */
int ssgpu_plan_chunked_form(ssgpu_plan* plan, int32_t* kind, const char** head, const char** tail);
/* The auxiliary input of a HASH_JOIN plan (rhs: DEVICE columns of the dimension table).  Stays
 * bound until replaced; the join index is rebuilt from it at the start of every run. */
int ssgpu_plan_set_aux_input(ssgpu_plan* plan, const ssgpu_column* cols, int32_t n_cols, int64_t rows);
/* Thread-safe, non-blocking; the running/next run returns SSGPU_INTERRUPTED. */
void ssgpu_interrupt(ssgpu_plan* plan);

/* Multi-GPU partial aggregates (row-range shards).  For plans
 * whose root is a ScalarAggregate the run can stop before finalisation and expose
 * element-wise reducible partial buffers (a GroupAggregate with dense slots has
 * its own, table-shaped form: ssgpu_plan_run_dense below).
 * Segment `i` is `count` elements of `dtype` to be combined across ranks with
 * `reduce` (0 = sum, 1 = min, 2 = max); the caller all-reduces each segment in
 * place (RCCL) and then calls ssgpu_plan_finalize.
 * STRING results (MIN / MAX / FIRST / LAST of a STRING column) are codes of the plan's dictionary: states -- and result
 * images, below -- of different plans may only be combined when the plans were created with the SAME dictionary (every
 * rank passing the job's strings: ssgpu_dict_*; distributed.py: job_strings).  The host drivers that do not build one
 * refuse STRING results (include/supersonic_amd/sharded.h). */
typedef struct ssgpu_partial_segment {
  void* device_ptr;
  int64_t count;
  int32_t dtype;  /* SSGPU_INT64 / SSGPU_UINT64 / SSGPU_DOUBLE */
  int32_t reduce; /* 0 sum, 1 min, 2 max */
} ssgpu_partial_segment;
int ssgpu_plan_run_partial(ssgpu_plan* plan, const ssgpu_column* cols, int32_t n_cols,
                           int64_t rows, int64_t global_row_offset);
int32_t ssgpu_plan_partial_segments(ssgpu_plan* plan, ssgpu_partial_segment* out,
                                    int32_t max_segments);
/* Folds `n_images` images of the partial state into the plan's own state with ONE kernel on the
 * plan's stream: `images` is device memory holding n_images consecutive copies of the buffer the
 * segments describe (all segments are contiguous, in segment order) -- exactly what an all-gather
 * of that buffer across ranks produces.  Equivalent to reducing each segment with its `reduce`
 * operator; the alternative to one all-reduce per segment, and the only way to merge FIRST / LAST
 * aggregates (their value travels with the smallest / largest contributing global row id, which
 * an element-wise all-reduce cannot express). */
int ssgpu_plan_fold_partials(ssgpu_plan* plan, const void* images, int32_t n_images);

/* ---- dense-slot GroupAggregate across ranks (ABI 7; SURVEY 8(e): "slot = dense key index") ----------------------------------
 * A GroupAggregate whose group keys are input columns with small value ranges (integers, BOOL, DATE / DATETIME, STRING codes)
 * and whose aggregate inputs are input columns (under Filters of the form `column CMP constant`) can index its table by
 * the keys' mixed-radix number instead of a hash: slot = sum_k (value_k - lo_k) * stride_k.  A single process does this on
 * its own (context option "group_dense", default 1: one pass over the key columns finds the ranges on a plan's first large
 * run; ssgpu_stage_info.dense_slots says it happened).  Across ranks it makes every shard's partial table THE SAME ARRAY, so
 * the exchange is one all-to-all of contiguous slot slices and one element-wise fold -- no merge plan, no routing, no
 * packed images, and DOUBLE sums cross as their raw (hi, lo) accumulators (rounded once, after the fold):
 *
 *   set-up   every rank: ssgpu_plan_key_ranges over its shard; the ranks agree on the union (min of lo, max of hi: any
 *            host-side exchange) and each calls ssgpu_plan_set_dense with the SAME ranges and n_chunks = number of ranks.
 *   step     ssgpu_plan_run_dense(shard columns, table)   table: n_chunks * chunk_bytes of device memory; chunk r = the slots
 *                                                         rank r owns (64-byte header + keys + accumulators + counts)
 *            ONE all-to-all of the chunks (chunk r of every rank -> rank r)
 *            ssgpu_plan_fold_dense(received chunks, n)    folds the n images of the owned slot range and extracts them: the
 *                                                         result holds the groups this rank owns (every group on one rank)
 *   check    ssgpu_plan_dense_flags (after any number of steps; it synchronises): what the headers of the LAST fold carried
 *            -- bit 1 a record segment ran full on some rank (ssgpu_plan_dense_grow on every rank, repeat the step), bit 2 a
 *            key outside the ranges (agree on new ranges, ssgpu_plan_set_dense again, repeat), error: the OR of the ranks'
 *            evaluation-error words.  Every rank sees the same flags: a rank whose ssgpu_plan_run_dense FAILED (memory quota,
 *            interrupt, ...) calls ssgpu_plan_dense_fail(table, code) instead and still takes part in the all-to-all -- its
 *            chunks arrive flagged (bit 3, the return code in bits 8..23 of `flags`), every rank raises the same error, and
 *            nobody waits in the collective for a rank that has already returned.
 *
 * ssgpu_plan_key_ranges / _set_dense return SSGPU_ERROR_NOT_IMPLEMENTED for a plan this form cannot take (not a single
 * plain GroupAggregate stage; floating keys; FIRST / LAST aggregates, whose values live in the shard that saw the row) and
 * ssgpu_plan_set_dense returns SSGPU_ERROR_INVALID_ARGUMENT_VALUE for ranges whose table would be too large
 * (> 2^21 slots): the caller then uses the image exchange (result images, below).  lo / hi are in an order-preserving
 * unsigned domain (signed key columns: value XOR 2^63 after sign extension); lo[k] > hi[k] = "no value of key k seen". */
typedef struct ssgpu_dense_layout {
  int64_t slots;        /* product of the key spans */
  int32_t n_parts;      /* partitions a shard aggregates in (a multiple of n_chunks) */
  int32_t part_cap;     /* table entries per partition */
  int64_t chunk_slots;  /* regular slots of a chunk = (n_parts / n_chunks) * part_cap; one special slot follows them */
  int64_t chunk_bytes;  /* 64-byte header + (chunk_slots + 1) * (8 + 8 n_gaggs [+ 4 n_gaggs]) bytes, rounded to 64 */
  int32_t n_gaggs;      /* accumulator words per group */
  int32_t has_counts;   /* contribution counts are kept (some aggregate input is nullable) */
} ssgpu_dense_layout;
int ssgpu_plan_key_ranges(ssgpu_plan* plan, const ssgpu_column* cols, int32_t n_cols, int64_t rows, int32_t* n_keys, uint64_t* lo, uint64_t* hi);
int ssgpu_plan_set_dense(ssgpu_plan* plan, int32_t n_keys, const uint64_t* lo, const uint64_t* hi, int32_t n_chunks, ssgpu_dense_layout* out);
int ssgpu_plan_run_dense(ssgpu_plan* plan, const ssgpu_column* cols, int32_t n_cols, int64_t rows, void* table);
int ssgpu_plan_fold_dense(ssgpu_plan* plan, const void* chunks, int32_t n_chunks, ssgpu_result** out);
int ssgpu_plan_dense_flags(ssgpu_plan* plan, uint32_t* flags, uint32_t* error);
int ssgpu_plan_dense_grow(ssgpu_plan* plan);
int ssgpu_plan_dense_fail(ssgpu_plan* plan, void* table, int32_t code);
int ssgpu_plan_finalize(ssgpu_plan* plan, ssgpu_result** out);
/* ssgpu_plan_fold_partials + ssgpu_plan_finalize as ONE kernel launch (ABI 6): what follows the collective of a sharded scalar
 * aggregate is a few hundred bytes of work -- three dependent launches of it cost a measurable part of a step at the shard
 * sizes of an 8-GPU job (12.5 M rows: 0.13 ms of scan).  Same result as the two calls. */
int ssgpu_plan_fold_finalize(ssgpu_plan* plan, const void* images, int32_t n_images, ssgpu_result** out);

/* ---- result images: the ONE-collective exchange of materialised results ----------------------
 * Multi-GPU GroupAggregate over row-range shards (the reference documents the same
 * external pattern -- aggregate per shard, shuffle, final aggregate -- at cursor/core/aggregate.h:236-242):
 * every rank packs its partial group table into ONE contiguous device buffer of a size all ranks
 * agree on (an "image": 64-byte header with the row count, then every column's data and NULL mask
 * padded to `capacity_rows`), ONE all-gather moves the images, and ssgpu_images_unpack lays the
 * n_images tables out as the contiguous columns of a (n_images * capacity_rows)-row View plus a
 * trailing BOOL validity column (1 for real rows, 0 for padding) that the merge plan filters on.
 * Row counts travel inside the images: nothing between the per-shard run and the merge run reads
 * a device value on the host.  All three calls are asynchronous on the context's stream.
 *   image_bytes / unpacked_bytes: sizes of one image and of the unpacked table;
 *   offsets[4 * i + 0..3] for attribute i: data / NULL-mask offset inside an image, data / NULL-mask
 *     offset inside the unpacked buffer (-1: not nullable); entry n_attrs (the validity column)
 *     has only the unpacked data offset. `offsets` may be NULL.
 * A table with more rows than capacity_rows is truncated and flagged in the header (word 2), and a
 * run that hit an evaluation error (signaling division, SQRT ...) carries its error word (word 4);
 * ssgpu_images_unpack leaves {largest row count seen, sum of row counts, any overflow, OR of the
 * error words} as four int64 at the end of the unpacked buffer (unpacked_bytes - 32) for the
 * caller's final check. */
int ssgpu_plan_image_layout(const ssgpu_plan* plan, int64_t capacity_rows, int32_t n_images,
                            int64_t* image_bytes, int64_t* unpacked_bytes, int64_t* offsets);
int ssgpu_result_pack_image(ssgpu_result* r, int64_t capacity_rows, void* image);
/* Key-range exchange (the form that scales): instead of sending its whole partial table to every rank, a shard routes
 * every row to ONE of n_dest images by a hash of the row's first n_keys columns (the group keys; NULL keys hash as a flag)
 * -- image d of `images` (n_dest consecutive images of ssgpu_plan_image_layout(plan, capacity_rows, 1) bytes) holds the
 * rows whose key belongs to rank d.  One all-to-all of the equally sized images then gives every rank all partial rows
 * of the groups it owns; it merges 1 / n_dest of the key space instead of everything (ssgpu_images_unpack + the merge
 * plan as before).  The hash depends on the key bytes only: every rank agrees on a key's owner.  Headers as for
 * ssgpu_result_pack_image (an image that is full is flagged; a set flag makes the receiver repeat the step). */
int ssgpu_result_route_images(ssgpu_result* r, int32_t n_keys, int32_t n_dest, int64_t capacity_rows, void* images);
int ssgpu_images_unpack(ssgpu_plan* plan, const void* images, int32_t n_images, int64_t capacity_rows,
                        void* unpacked, ssgpu_column* cols /* attr_count + 1 */);

/* ---- result (ResultView / View) ------------------------------------------- */
void ssgpu_result_destroy(ssgpu_result* r);
int64_t ssgpu_result_row_count(ssgpu_result* r);
int32_t ssgpu_result_column_count(const ssgpu_result* r);
/* Host view of column i (pinned memory owned by the result; D2H on first use). */
int ssgpu_result_column(ssgpu_result* r, int32_t i, const void** data,
                        const uint8_t** is_null);
/* Device view of column i (no copy). */
int ssgpu_result_device_column(ssgpu_result* r, int32_t i, ssgpu_column* out);

/* ---- counters (profiling seam: CursorStatistics, benchmark/) -------------- */
typedef struct ssgpu_counters {
  double kernel_ms;        /* HIP-event time of all kernels of the last run   */
  double dominant_ms;      /* HIP-event time of the dominant (pipeline) kernel */
  int64_t rows_in;
  int64_t rows_out;
  int64_t algorithmic_bytes; /* bytes the plan must move through HBM once     */
  int32_t n_launches;
  int32_t tile_rows;
  int32_t grid;
  int32_t lds_bytes;
} ssgpu_counters;
/* Durations (ms) of the dominant kernel of the most recent runs of the plan, oldest first (HIP events
 * recorded on the plan's stream around that kernel, kept for the last 256 runs): lets a caller time many
 * asynchronous runs and read every kernel duration afterwards.  Waits for the stream; returns the
 * number of values written (<= max). */
int32_t ssgpu_plan_recent_kernel_ms(ssgpu_plan* plan, double* out_ms, int32_t max);
int ssgpu_plan_counters(ssgpu_plan* plan, ssgpu_counters* out);

#ifdef __cplusplus
}
#endif
#endif /* SSGPU_H_ */
