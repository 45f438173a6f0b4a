"""ctypes binding of the C ABI in include/ssgpu.h (libssgpu.so).

This is the reference-side stub a maintainer would write for a Python host: plain
pointers and sizes, no torch types.  The library is built in-tree by
``__graft_entry__.build()`` (``make -C supersonic_amd/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SSGPU_LIB") or os.path.join(_HERE, "lib", "libssgpu.so")   # SSGPU_LIB: build-variant experiments (tools/)

# reference enum values (supersonic/proto/supersonic.proto:15-36,86-101)
INT32, INT64, UINT64, DATETIME, DOUBLE, BOOL, UINT32, FLOAT, DATE, STRING, BINARY = 1, 2, 3, 4, 5, 6, 8, 9, 10, 0, 7
NOT_NULLABLE, NULLABLE = 0, 1
SUM, MIN, MAX, COUNT, CONCAT, FIRST, LAST = 0, 1, 2, 3, 4, 5, 6
SUM_RESIDUAL = 100   # extension, see include/ssgpu.h
ASCENDING, DESCENDING = 0, 1

OK = 0
ERROR_UNKNOWN = 100
ERROR_GENERAL_IO_ERROR = 101
ERROR_MEMORY_EXCEEDED = 102
ERROR_NOT_IMPLEMENTED = 103
ERROR_EVALUATION_ERROR = 104
ERROR_TOO_MANY_ROWS = 302
ERROR_ATTRIBUTE_COUNT_MISMATCH = 401
ERROR_ATTRIBUTE_TYPE_MISMATCH = 402
ERROR_ATTRIBUTE_MISSING = 403
ERROR_ATTRIBUTE_EXISTS = 404
ERROR_INVALID_ARGUMENT_TYPE = 405
ERROR_INVALID_ARGUMENT_VALUE = 407
INTERRUPTED = 1000
ERROR_NO_DEVICE = 2000
ERROR_HIP = 2001

EXPR_ATTR_NAMED, EXPR_ATTR_AT, EXPR_CONST, EXPR_NULL, EXPR_OP, EXPR_ALIAS, EXPR_COMPOUND, EXPR_CAST = 1, 2, 3, 4, 5, 6, 7, 8
OP_GREATER, OP_GREATER_OR_EQUAL, OP_NULLING_IF, OP_ROUND_WITH_PRECISION = 100001, 100002, 100003, 100004
PROJ_ALL, PROJ_NAMED, PROJ_AT, PROJ_NAMED_AS = 1, 2, 3, 4
JOIN_INNER, JOIN_LEFT_OUTER = 0, 1
KEYS_NOT_UNIQUE, KEYS_UNIQUE = 0, 1
OP_SCAN, OP_COMPUTE, OP_FILTER, OP_PROJECT, OP_SCALAR_AGGREGATE, OP_GROUP_AGGREGATE, OP_AGGREGATE_CLUSTERS, OP_SORT = 1, 2, 3, 4, 5, 6, 7, 8
OP_BEST_EFFORT_GROUP_AGGREGATE = 10
OP_HASH_JOIN = 9


class Attr(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_int32), ("nullable", C.c_int32)]


class Expr(C.Structure):
    _fields_ = [("kind", C.c_int32), ("op", C.c_int32), ("dtype", C.c_int32), ("first_arg", C.c_int32),
                ("nargs", C.c_int32), ("reserved", C.c_int32), ("i64", C.c_int64), ("f64", C.c_double),
                ("name", C.c_char_p)]


class Proj(C.Structure):
    _fields_ = [("kind", C.c_int32), ("position", C.c_int32), ("name", C.c_char_p), ("alias", C.c_char_p),
                ("source", C.c_int32), ("reserved", C.c_int32)]


class Agg(C.Structure):
    _fields_ = [("aggregation", C.c_int32), ("distinct", C.c_int32), ("output_type", C.c_int32),
                ("reserved", C.c_int32), ("input", C.c_char_p), ("output", C.c_char_p)]


class SortKey(C.Structure):
    _fields_ = [("name", C.c_char_p), ("order", C.c_int32), ("reserved", C.c_int32)]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("child", C.c_int32), ("expr", C.c_int32), ("proj_first", C.c_int32),
                ("proj_n", C.c_int32), ("agg_first", C.c_int32), ("agg_n", C.c_int32), ("sort_first", C.c_int32),
                ("sort_n", C.c_int32), ("child2", C.c_int32), ("option0", C.c_int64),
                ("proj2_first", C.c_int32), ("proj2_n", C.c_int32), ("proj3_first", C.c_int32), ("proj3_n", C.c_int32)]


class MemoryStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("device_bytes", "pinned_bytes", "live_plans", "live_blocks", "events",
                                          "rtc_modules", "rtc_code_bytes", "rtc_compilations", "rtc_disk_hits")]


class StageInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "group_shape", "part_n", "part_seg_growth", "group_wgs_per_cu", "reruns",
                                          "sort_passes", "sort_mode", "specialized", "plain_scatter", "hot_keys", "dense_slots", "split_records", "row_ranges")] + [("reserved", C.c_int32 * 2)]


class PlanDesc(C.Structure):
    _fields_ = [("input_schema", C.POINTER(Attr)), ("n_attrs", C.c_int32),
                ("ops", C.POINTER(Op)), ("n_ops", C.c_int32),
                ("exprs", C.POINTER(Expr)), ("n_exprs", C.c_int32),
                ("expr_args", C.POINTER(C.c_int32)), ("n_expr_args", C.c_int32),
                ("projs", C.POINTER(Proj)), ("n_projs", C.c_int32),
                ("aggs", C.POINTER(Agg)), ("n_aggs", C.c_int32),
                ("sortkeys", C.POINTER(SortKey)), ("n_sortkeys", C.c_int32),
                ("aux_schema", C.POINTER(Attr)), ("n_aux_attrs", C.c_int32)]


class Column(C.Structure):
    _fields_ = [("data", C.c_void_p), ("is_null", C.c_void_p)]


class PartialSegment(C.Structure):
    _fields_ = [("device_ptr", C.c_void_p), ("count", C.c_int64), ("dtype", C.c_int32), ("reduce", C.c_int32)]


class DenseLayout(C.Structure):
    _fields_ = [("slots", C.c_int64), ("n_parts", C.c_int32), ("part_cap", C.c_int32), ("chunk_slots", C.c_int64), ("chunk_bytes", C.c_int64),
                ("n_gaggs", C.c_int32), ("has_counts", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("dominant_ms", C.c_double), ("rows_in", C.c_int64),
                ("rows_out", C.c_int64), ("algorithmic_bytes", C.c_int64), ("n_launches", C.c_int32),
                ("tile_rows", C.c_int32), ("grid", C.c_int32), ("lds_bytes", C.c_int32)]


# every symbol include/ssgpu.h declares: (name, restype, argtypes)
P = C.c_void_p
SYMBOLS = [
    ("ssgpu_abi_version", C.c_int, []),
    ("ssgpu_ctx_create", C.c_int, [C.c_int, C.POINTER(P)]),
    ("ssgpu_ctx_destroy", None, [P]),
    ("ssgpu_last_error", C.c_char_p, [P]),
    ("ssgpu_ctx_has_device", C.c_int, [P]),
    ("ssgpu_allocator_create", C.c_int, [P, C.c_int64, C.POINTER(P)]),
    ("ssgpu_allocator_destroy", None, [P]),
    ("ssgpu_allocator_allocate", C.c_int, [P, C.c_size_t, C.c_size_t, C.POINTER(P), C.POINTER(C.c_size_t)]),
    ("ssgpu_allocator_reallocate", C.c_int, [P, P, C.c_size_t, C.c_size_t, C.POINTER(P), C.POINTER(C.c_size_t)]),
    ("ssgpu_allocator_free", None, [P, P]),
    ("ssgpu_allocator_available", C.c_int64, [P]),
    ("ssgpu_allocator_allocated", C.c_int64, [P]),
    ("ssgpu_dict_create", C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.c_int64, C.POINTER(P)]),
    ("ssgpu_dict_destroy", None, [P]),
    ("ssgpu_dict_size", C.c_int32, [P]),
    ("ssgpu_dict_encode", C.c_int, [P, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), P, C.c_int64, C.POINTER(C.c_int32)]),
    ("ssgpu_dict_decode", C.c_int, [P, C.c_int32, C.POINTER(P), C.POINTER(C.c_int32)]),
    ("ssgpu_plan_set_memory_limit", C.c_int, [P, C.c_int64]),
    ("ssgpu_plan_set_dict", C.c_int, [P, P]),
    ("ssgpu_result_column_dict", P, [P, C.c_int32]),
    ("ssgpu_plan_specialized", C.c_int32, [P]),
    ("ssgpu_plan_specialize", C.c_int, [P]),
    ("ssgpu_specialized_kernels_trim", None, [C.c_int32]),
    ("ssgpu_plan_stage_count", C.c_int32, [P]),
    ("ssgpu_plan_stage_info", C.c_int, [P, C.c_int32, C.POINTER(StageInfo)]),
    ("ssgpu_plan_specialize_reason", C.c_char_p, [P]),
    ("ssgpu_memory_stats", C.c_int, [C.POINTER(MemoryStats)]),
    ("ssgpu_pool_trim", C.c_int64, [C.c_int32]),
    ("ssgpu_plan_set_option", C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    ("ssgpu_plan_run_best_effort", C.c_int, [C.c_void_p, C.POINTER(Column), C.c_int32, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_void_p)]),
    ("ssgpu_plan_memory_in_use", C.c_int64, [P]),
    ("ssgpu_expr_bind", C.c_int, [P, C.POINTER(Attr), C.c_int32, C.POINTER(Expr), C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int64, C.POINTER(P)]),
    ("ssgpu_expr_row_capacity", C.c_int64, [P]),
    ("ssgpu_expr_evaluate", C.c_int, [P, C.POINTER(Column), C.c_int32, C.c_int64, C.POINTER(P)]),
    ("ssgpu_expr_evaluate_skip", C.c_int, [P, C.POINTER(Column), C.c_int32, C.c_int64, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(P)]),
    ("ssgpu_ctx_stream", P, [P]),
    ("ssgpu_ctx_copy_stream", P, [P]),
    ("ssgpu_ctx_set_stream", C.c_int, [P, P]),
    ("ssgpu_ctx_synchronize", C.c_int, [P]),
    ("ssgpu_ctx_set_option", C.c_int, [P, C.c_char_p, C.c_int64]),
    ("ssgpu_host_alloc", C.c_int, [P, C.c_size_t, C.POINTER(P)]),
    ("ssgpu_host_free", None, [P, P]),
    ("ssgpu_block_create", C.c_int, [P, C.POINTER(Attr), C.c_int32, C.c_int64, C.POINTER(P)]),
    ("ssgpu_block_destroy", None, [P]),
    ("ssgpu_block_upload", C.c_int, [P, C.c_int32, P, P, C.c_int64, C.c_int64]),
    ("ssgpu_block_set_row_count", C.c_int, [P, C.c_int64]),
    ("ssgpu_block_row_count", C.c_int64, [P]),
    ("ssgpu_block_column", C.c_int, [P, C.c_int32, C.POINTER(Column)]),
    ("ssgpu_block_create_from_file", C.c_int, [P, C.POINTER(Attr), C.c_int32, C.c_char_p, C.POINTER(P)]),
    ("ssgpu_block_write_file", C.c_int, [P, C.c_char_p]),
    ("ssgpu_result_write_file", C.c_int, [P, C.c_char_p]),
    ("ssgpu_plan_create", C.c_int, [P, C.POINTER(PlanDesc), C.POINTER(P)]),
    ("ssgpu_plan_destroy", None, [P]),
    ("ssgpu_plan_attr_count", C.c_int32, [P]),
    ("ssgpu_plan_attr", C.c_int, [P, C.c_int32, C.POINTER(Attr)]),
    ("ssgpu_plan_describe", C.c_char_p, [P]),
    ("ssgpu_plan_program", C.c_int, [P, C.c_int32, C.POINTER(P), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("ssgpu_plan_run", C.c_int, [P, C.POINTER(Column), C.c_int32, C.c_int64, C.POINTER(P)]),
    ("ssgpu_plan_run_block", C.c_int, [P, P, C.POINTER(P)]),
    ("ssgpu_plan_run_host", C.c_int, [P, C.POINTER(Column), C.c_int32, C.c_int64, C.c_int64, C.POINTER(P)]),
    ("ssgpu_plan_stream_begin", C.c_int, [P, C.c_int64]),
    ("ssgpu_plan_stream_push", C.c_int, [P, C.POINTER(Column), C.c_int32, C.c_int64]),
    ("ssgpu_plan_stream_finish", C.c_int, [P, C.POINTER(P)]),
    ("ssgpu_plan_chunked_form", C.c_int, [P, C.POINTER(C.c_int32), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]),
    ("ssgpu_plan_set_aux_input", C.c_int, [P, C.POINTER(Column), C.c_int32, C.c_int64]),
    ("ssgpu_interrupt", None, [P]),
    ("ssgpu_plan_run_partial", C.c_int, [P, C.POINTER(Column), C.c_int32, C.c_int64, C.c_int64]),
    ("ssgpu_plan_partial_segments", C.c_int32, [P, C.POINTER(PartialSegment), C.c_int32]),
    ("ssgpu_plan_fold_partials", C.c_int, [P, P, C.c_int32]),
    ("ssgpu_plan_key_ranges", C.c_int, [P, C.POINTER(Column), C.c_int32, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("ssgpu_plan_set_dense", C.c_int, [P, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32, C.POINTER(DenseLayout)]),
    ("ssgpu_plan_run_dense", C.c_int, [P, C.POINTER(Column), C.c_int32, C.c_int64, P]),
    ("ssgpu_plan_fold_dense", C.c_int, [P, P, C.c_int32, C.POINTER(P)]),
    ("ssgpu_plan_dense_flags", C.c_int, [P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("ssgpu_plan_dense_grow", C.c_int, [P]),
    ("ssgpu_plan_dense_fail", C.c_int, [P, P, C.c_int32]),
    ("ssgpu_plan_finalize", C.c_int, [P, C.POINTER(P)]),
    ("ssgpu_plan_fold_finalize", C.c_int, [P, P, C.c_int32, C.POINTER(P)]),
    ("ssgpu_plan_image_layout", C.c_int, [P, C.c_int64, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("ssgpu_result_pack_image", C.c_int, [P, C.c_int64, P]),
    ("ssgpu_result_route_images", C.c_int, [P, C.c_int32, C.c_int32, C.c_int64, P]),
    ("ssgpu_images_unpack", C.c_int, [P, P, C.c_int32, C.c_int64, P, C.POINTER(Column)]),
    ("ssgpu_result_destroy", None, [P]),
    ("ssgpu_result_row_count", C.c_int64, [P]),
    ("ssgpu_result_column_count", C.c_int32, [P]),
    ("ssgpu_result_column", C.c_int, [P, C.c_int32, C.POINTER(P), C.POINTER(P)]),
    ("ssgpu_result_device_column", C.c_int, [P, C.c_int32, C.POINTER(Column)]),
    ("ssgpu_plan_counters", C.c_int, [P, C.POINTER(Counters)]),
    ("ssgpu_plan_recent_kernel_ms", C.c_int32, [P, C.POINTER(C.c_double), C.c_int32]),
]

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm ships its own libamdhip64.so (SONAME libamdhip64.so.7) and finds it by file name,
    so a process that loads libssgpu.so first (which binds /opt/rocm's copy by SONAME) and torch later
    ends up with TWO HIP runtimes, and the second one sees no GPU.  Loading torch's copy first makes
    both resolve to the same runtime in either import order (torch itself is not imported here)."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libssgpu.so; fails loudly if the HIP extension was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libssgpu.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` -- "
                "there is no CPU fallback for the product path" % LIB_PATH)
        _share_hip_runtime_with_torch()
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            if os.environ.get("SSGPU_LIB") and not hasattr(lib, name):
                continue             # (a build-variant experiment with an OLDER library, tools/ab/: entry points it lacks are not bound)
            fn = getattr(lib, name)  # AttributeError if the library does not export the ABI
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
