"""ctypes driver of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() as the checker.  It duck-types the plain description
objects of supersonic_amd.api (Expression / Operation trees) so one tree can be
evaluated by the HIP path and by the oracle; it shares no execution code with
the product.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_NP = {1: np.int32, 2: np.int64, 8: np.uint32, 3: np.uint64, 9: np.float32, 5: np.float64, 6: np.bool_,
       10: np.int32, 4: np.int64, 0: np.int32}   # 0 = STRING: dictionary codes inside the C restatement
T_STRING = 0

KIND = {"ScanView": 1, "Compute": 2, "Filter": 3, "Project": 4, "ScalarAggregate": 5, "GroupAggregate": 6, "BestEffortGroupAggregate": 6,
        "AggregateClusters": 7, "Sort": 8, "HashJoinOperation": 9}


class OracleError(Exception):
    def __init__(self, return_code, message):
        Exception.__init__(self, "[%d] %s" % (return_code, message))
        self.return_code = return_code
        self.message = message


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        P = C.c_void_p
        L.orc_expr_new.restype = P
        L.orc_expr_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_double, C.c_char_p]
        L.orc_expr_add_arg.argtypes = [P, P]
        L.orc_op_new.restype = P
        L.orc_op_new.argtypes = [C.c_int, P, P]
        L.orc_op_add_proj.argtypes = [P, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
        L.orc_op_add_agg.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
        L.orc_op_add_sortkey.argtypes = [P, C.c_char_p, C.c_int]
        L.orc_op_set_max_unique_keys.argtypes = [P, C.c_int64]
        L.orc_op_set_best_effort_quota.argtypes = [P, C.c_int64]
        L.orc_op_set_join.argtypes = [P, P, C.c_int]
        L.orc_op_add_proj_to.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
        L.orc_scan_add_column.argtypes = [P, C.c_char_p, C.c_int, C.c_int, P, P]
        L.orc_scan_set_rows.argtypes = [P, C.c_int64]
        L.orc_create_cursor.restype = P
        L.orc_create_cursor.argtypes = [P]
        L.orc_cursor_error.restype = C.c_int
        L.orc_cursor_error.argtypes = [P, C.c_char_p, C.c_int]
        L.orc_cursor_ncols.restype = C.c_int
        L.orc_cursor_ncols.argtypes = [P]
        L.orc_cursor_col_name.restype = C.c_char_p
        L.orc_cursor_col_name.argtypes = [P, C.c_int]
        L.orc_cursor_col_type.restype = C.c_int
        L.orc_cursor_col_type.argtypes = [P, C.c_int]
        L.orc_cursor_col_nullable.restype = C.c_int
        L.orc_cursor_col_nullable.argtypes = [P, C.c_int]
        L.orc_next.restype = C.c_int
        L.orc_next.argtypes = [P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(P), C.POINTER(P)]
        L.orc_drain.restype = C.c_int64
        L.orc_drain.argtypes = [P]
        L.orc_bench_threads.restype = C.c_int
        L.orc_bench_threads.argtypes = [P, P, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double)]
        L.orc_bench_stream_read.restype = C.c_double
        L.orc_bench_stream_read.argtypes = [P, C.c_int, C.c_int, C.POINTER(C.c_int)]
        _LIB = L
    return _LIB


def _enc(s):
    return None if s is None else (s.encode() if isinstance(s, str) else s)


def _bytes(v):
    return v.encode() if isinstance(v, str) else bytes(v)


def _collect_strings(o, found):
    """STRING constants and STRING column values of a tree (own walk: no code shared with the product)."""
    def walk(e):
        if e is None:
            return
        if getattr(e, "sval", None) is not None:
            found.add(_bytes(e.sval))
        for a in getattr(e, "args", ()):
            walk(a)
    while o is not None:
        if getattr(o, "rhs_child", None) is not None:
            _collect_strings(o.rhs_child, found)
        walk(getattr(o, "expression", None))
        walk(getattr(o, "predicate", None))
        v = getattr(o, "view", None)
        if v is not None:
            schema = v.schema()
            for i in range(schema.attribute_count()):
                if schema.attribute(i).type() == T_STRING:
                    col = v.column(i)
                    for j, s in enumerate(col.data):
                        if col.is_null is None or not col.is_null[j]:
                            found.add(_bytes(s))
        o = getattr(o, "child", None)


class _Tree(object):
    def __init__(self):
        self.keep = []
        self.values = []     # order-preserving dictionary: sorted byte strings; code = index
        self.code = {}

    def expr(self, e):
        L = lib()
        i64 = int(e.i64)
        if e.kind == 3 and e.dtype == T_STRING:      # ConstString -> its code
            i64 = self.code[_bytes(e.sval)]
        h = L.orc_expr_new(e.kind, e.op, e.dtype, i64, float(e.f64), _enc(e.name))
        for a in e.args:
            L.orc_expr_add_arg(h, self.expr(a))
        return h

    def op(self, o):
        L = lib()
        kind = KIND[type(o).__name__]
        if kind == 1:
            h = L.orc_op_new(1, None, None)
            v = o.view
            schema = v.schema()
            for i in range(schema.attribute_count()):
                a = schema.attribute(i)
                col = v.column(i)
                if a.type() == T_STRING:
                    data = np.array([0 if (col.is_null is not None and col.is_null[j]) else self.code[_bytes(sv)]
                                     for j, sv in enumerate(col.data)], dtype=np.int32)
                else:
                    data = np.ascontiguousarray(col.data)
                nulls = None if col.is_null is None else np.ascontiguousarray(col.is_null).view(np.uint8)
                self.keep += [data, nulls]
                L.orc_scan_add_column(h, _enc(a.name()), a.type(), a.nullability(), data.ctypes.data_as(C.c_void_p),
                                      None if nulls is None else nulls.ctypes.data_as(C.c_void_p))
            L.orc_scan_set_rows(h, v.row_count())
            return h
        child = self.op(o.child)
        if kind == 9:   # HashJoinOperation(type, lhs keys, rhs keys, result projector, uniqueness, lhs, rhs)
            rhs = self.op(o.rhs_child)
            h = L.orc_op_new(9, child, None)
            L.orc_op_set_join(h, rhs, int(o.join_type) | (int(o.uniqueness) << 8))
            for (k, pos, name, alias) in o.lhs_keys.entries:
                L.orc_op_add_proj(h, k, pos, _enc(name), _enc(alias))
            for (k, pos, name, alias) in o.rhs_keys.entries:
                L.orc_op_add_proj_to(h, 2, 0, k, pos, _enc(name), _enc(alias))
            for (source, k, pos, name, alias) in o.result_projector.entries:
                L.orc_op_add_proj_to(h, 3, source, k, pos, _enc(name), _enc(alias))
            return h
        expr = None
        if kind == 2:
            expr = self.expr(o.expression)
        elif kind == 3:
            expr = self.expr(o.predicate)
        h = L.orc_op_new(kind, child, expr)
        proj = getattr(o, "projector", None) if kind in (3, 4, 8) else getattr(o, "group_by", None)
        if kind == 8 and proj is None:
            L.orc_op_add_proj(h, 1, 0, None, None)
        elif proj is not None:
            for (k, pos, name, alias) in proj.entries:
                L.orc_op_add_proj(h, k, pos, _enc(name), _enc(alias))
        spec = getattr(o, "spec", None)
        if spec is not None:
            for (agg, distinct, otype, inp, outp) in spec.elements:
                L.orc_op_add_agg(h, agg, distinct, otype, _enc(inp), _enc(outp))
        if kind == 8:
            for (name, order) in o.order.keys:
                L.orc_op_add_sortkey(h, _enc(name), order)
        if type(o).__name__ == "BestEffortGroupAggregate":      # aggregate.h:230-250; options.memory_quota bounds the result block
            quota = getattr(getattr(o, "options", None), "memory_quota", None)
            L.orc_op_set_best_effort_quota(h, int(quota) if quota is not None else 0)
        elif kind == 6 and getattr(o, "options", None) is not None:
            limit = o.options.max_unique_keys_in_result        # GroupAggregateOptions (aggregate.h:160-205): kint64max = no limit
            if limit < (1 << 63) - 1:
                L.orc_op_set_max_unique_keys(h, int(limit))
        return h


class Cursor(object):
    def __init__(self, operation):
        self.tree = _Tree()
        found = set()
        _collect_strings(operation, found)
        self.tree.values = sorted(found)
        self.tree.code = {v: i for i, v in enumerate(self.tree.values)}
        self.handle = lib().orc_create_cursor(self.tree.op(operation))
        buf = C.create_string_buffer(600)
        code = lib().orc_cursor_error(self.handle, buf, 600)
        if code:
            raise OracleError(code, buf.value.decode())
        L = lib()
        n = L.orc_cursor_ncols(self.handle)
        self.schema = [(L.orc_cursor_col_name(self.handle, i).decode(), L.orc_cursor_col_type(self.handle, i),
                        L.orc_cursor_col_nullable(self.handle, i)) for i in range(n)]

    def next(self, max_rows=1024):
        """-> list of (data, is_null|None) per column, or None at end of stream."""
        L = lib()
        n = len(self.schema)
        rows = C.c_int64()
        data = (C.c_void_p * max(n, 1))()
        nulls = (C.c_void_p * max(n, 1))()
        r = L.orc_next(self.handle, max_rows, C.byref(rows), data, nulls)
        if r < 0:
            buf = C.create_string_buffer(600)
            code = L.orc_cursor_error(self.handle, buf, 600)
            raise OracleError(code, buf.value.decode())
        if r == 0:
            return None
        out = []
        for i, (_name, t, _nullable) in enumerate(self.schema):
            dt = np.dtype(_NP[t])
            d = np.frombuffer(C.string_at(data[i], rows.value * dt.itemsize), dtype=dt).copy()
            z = None
            if nulls[i]:
                z = np.frombuffer(C.string_at(nulls[i], rows.value), dtype=np.uint8).copy() != 0
            out.append((d, z))
        return out

    def drain_discard(self):
        return lib().orc_drain(self.handle)


class ThreadedBench(object):
    """bench.py's N-thread CPU baseline (ss_oracle.c "bench harness"): `nthreads` pthreads, pinned to `cpus` (one id per thread,
    or None), each draining a fresh cursor of `operation` over its own contiguous row range of the plan's ScanView `passes`
    times; for GroupAggregate plans the partial tables are merged afterwards (every thread owns a hash range of the keys).
    `sample`: a plan of the same shape over a small view -- every thread fills ITS rows of the big view with copies of it
    before the clock starts (first touch by the thread that will read them: NUMA-local pages)."""

    def __init__(self, operation, sample=None):
        self.tree, self.sample_tree = _Tree(), _Tree()
        self.handle = self.tree.op(operation)
        self.sample = self.sample_tree.op(sample) if sample is not None else None

    @staticmethod
    def _cpus(cpus, nthreads):
        if cpus is None:
            return None
        assert len(cpus) >= nthreads
        return (C.c_int * nthreads)(*[int(c) for c in cpus[:nthreads]])

    def run(self, nthreads, passes, cpus=None, merge=True):
        """-> dict(seconds, merge_seconds (None if this plan has no merge step / it is not supported), merged_groups,
        merged_checksum, result_rows)"""
        out = (C.c_double * 5)()
        rc = lib().orc_bench_threads(self.handle, self.sample, int(nthreads), int(passes), self._cpus(cpus, nthreads), 1 if merge else 0, out)
        if rc == -1:
            raise OracleError(-1, "a cursor of the threaded CPU baseline failed")
        return {"seconds": out[0], "merge_seconds": out[1] if out[1] > 0 else None, "merged_groups": int(out[2]),
                "merged_checksum": out[3], "result_rows": int(out[4])}

    def stream_read(self, nthreads, passes, cpus=None):
        """bytes/s of a plain summing read of the ScanView's columns by the same threads"""
        return lib().orc_bench_stream_read(self.handle, int(nthreads), int(passes), self._cpus(cpus, nthreads))


A_CONCAT, T_BOOL, T_FLOAT, T_DOUBLE, T_DATE, T_DATETIME = 4, 6, 9, 5, 10, 4


def _civil(days):
    """(year, month, day) of a day count since 1970-01-01 in the proleptic Gregorian calendar (what gmtime gives, any year)."""
    z = days + 719468                                   # days since 0000-03-01; 400 years = 146097 days
    era, doe = divmod(z, 146097)
    yoe = (doe - doe // 1460 + doe // 36524 - doe // 146096) // 365
    doy = doe - (365 * yoe + yoe // 4 - yoe // 100)
    mp = (5 * doy + 2) // 153
    month = mp + 3 if mp < 10 else mp - 9
    return yoe + era * 400 + (1 if month <= 2 else 0), month, doy - (153 * mp + 2) // 5 + 1


def _print_typed(t, v):
    """PrintTyped (base/infrastructure/types_infrastructure.cc:45-80): decimal integers, TRUE / FALSE, SimpleFtoa / SimpleDtoa
    (the shortest of %.6g / %.9g resp. %.15g / %.17g that reads back as the same value), STRING as is."""
    if t == T_STRING:
        return _bytes(v)
    if t == T_BOOL:
        return b"TRUE" if v else b"FALSE"
    if t in (T_DATE, T_DATETIME):
        # PrintTyped<DATE> / <DATETIME> (types_infrastructure.cc:36-39,92-114): strftime("%Y/%m/%d" / "%Y/%m/%d-%H:%M:%S") of gmtime;
        # DATE's `value * (24 * 3600)` is an int32 product (wraps beyond +-24855 days: undefined in the reference), DATETIME drops
        # the microseconds toward zero; glibc's %Y prints the year unpadded
        if t == T_DATE:
            secs = ((int(v) * 86400 + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)
        else:
            secs = abs(int(v)) // 1000000 * (1 if int(v) >= 0 else -1)
        days, sod = divmod(secs, 86400)
        y, m, d = _civil(days)
        text = "%d/%02d/%02d" % (y, m, d)
        if t == T_DATETIME:
            text += "-%02d:%02d:%02d" % (sod // 3600, sod // 60 % 60, sod % 60)
        return text.encode()
    if t in (T_FLOAT, T_DOUBLE):
        f = float(v)
        if f != f:
            return b"nan"
        if f in (float("inf"), float("-inf")):
            return b"inf" if f > 0 else b"-inf"
        short, long_ = (6, 9) if t == T_FLOAT else (15, 17)
        text = "%.*g" % (short, f)
        back = np.float32(text) if t == T_FLOAT else float(text)
        if back != (np.float32(f) if t == T_FLOAT else f):
            text = "%.*g" % (long_, f)
        return text.encode()
    return str(int(v)).encode()


def _run_with_concat(operation, max_rows):
    """CONCAT aggregates (aggregation_operators.h:236-283, column_aggregator.cc:108-124,496-505): every non-NULL value of a
    group, printed, in input order, joined with ','; a group without one is NULL.  The C restatement keeps STRINGs as
    dictionary codes and cannot make new ones, so this part of the fold is restated here: the child's rows come from the C
    restatement, the groups are formed in first-seen order (GroupAggregate), as key runs (AggregateClusters) or as the one
    group of a ScalarAggregate -- the same orders the C restatement gives the other aggregates of the specification, which
    it still computes."""
    import copy
    kind = KIND[type(operation).__name__]
    cschema, crows = run(operation.child, max_rows)
    names = [c[0] for c in cschema]
    n = len(crows[0][0]) if crows else 0
    keys = []
    proj = getattr(operation, "group_by", None)
    if proj is not None:
        for (k, pos, name, _alias) in proj.entries:
            if k == 1:              # ProjectAllAttributes
                keys += list(range(len(names)))
            else:
                keys.append(pos if k == 3 else names.index(name))
    def key_of(i):
        return tuple((True, None) if (crows[k][1] is not None and crows[k][1][i]) else (False, crows[k][0][i].tobytes() if hasattr(crows[k][0][i], "tobytes") else crows[k][0][i]) for k in keys)
    group_of, n_groups = np.zeros(n, dtype=np.int64), (1 if kind == 5 else 0)
    if kind == 6:        # GroupAggregate: first-seen order (row_hash_set.cc:458-517)
        seen = {}
        limit = getattr(getattr(operation, "options", None), "max_unique_keys_in_result", None)
        for i in range(n):
            k = key_of(i)
            if k not in seen:
                # GroupAggregateOptions::max_unique_keys_in_result (row_hash_set.cc:500-511): an unseen key is appended while the index
                # holds <= limit rows; later unseen keys are answered with the index's last row and NOT remembered
                if limit is not None and len(seen) > limit:
                    group_of[i] = len(seen) - 1
                    continue
                seen[k] = len(seen)
            group_of[i] = seen[k]
        n_groups = len(seen)
    elif kind == 7:      # AggregateClusters: a new group whenever the key changes (aggregate_clusters.cc:338-520)
        prev = None
        for i in range(n):
            k = key_of(i)
            if i == 0 or k != prev:
                n_groups += 1
            prev = k
            group_of[i] = n_groups - 1
    rest = copy.copy(operation)
    rest.spec = copy.copy(operation.spec)
    rest.spec.elements = [e for e in operation.spec.elements if e[0] != A_CONCAT]
    rschema, rcols = run(rest, max_rows)
    n_keys = len(rschema) - len(rest.spec.elements)
    if kind != 5 and rcols:
        assert len(rcols[0][0]) == n_groups
    schema, cols, ri = list(rschema[:n_keys]), list(rcols[:n_keys]), n_keys
    for (agg, distinct, _otype, inp, outp) in operation.spec.elements:
        if agg != A_CONCAT:
            schema.append(rschema[ri]); cols.append(rcols[ri]); ri += 1
            continue
        c = names.index(inp)
        text, has = [b""] * n_groups, np.zeros(n_groups, dtype=bool)
        seen = set()
        for i in range(n):
            if crows[c][1] is not None and crows[c][1][i]:
                continue
            g = group_of[i]
            if distinct:
                # DISTINCT CONCAT: the DistinctAggregator in front of the CONCAT (column_aggregator.cc:308-376) keeps one set of seen
                # values per result row and passes a value on at its first occurrence only; values compare as in the C restatement's
                # distinct_seen(): by their bits, -0.0 as +0.0
                v = crows[c][0][i]
                if cschema[c][1] in (T_FLOAT, T_DOUBLE) and v == 0:
                    v = type(v)(0.0)
                k = (int(g), v.tobytes() if hasattr(v, "tobytes") else v)
                if k in seen:
                    continue
                seen.add(k)
            text[g] = (text[g] + b"," if has[g] else b"") + _print_typed(cschema[c][1], crows[c][0][i])
            has[g] = True
        data = np.empty(n_groups, dtype=object)
        data[:] = text
        schema.append((outp, T_STRING, 1)); cols.append((data, ~has))
    return schema, cols


def run(operation, max_rows=1024):
    """Evaluate an operation tree on the CPU: (schema, [(data, is_null|None), ...])."""
    spec = getattr(operation, "spec", None)
    if spec is not None and any(e[0] == A_CONCAT for e in spec.elements):
        return _run_with_concat(operation, max_rows)
    cur = Cursor(operation)
    parts = []
    while True:
        p = cur.next(max_rows)
        if p is None:
            break
        parts.append(p)
    cols = []
    for i, (_n, t, nullable) in enumerate(cur.schema):
        dt = np.dtype(_NP[t])
        d = np.concatenate([p[i][0] for p in parts]) if parts else np.zeros(0, dt)
        z = None
        if nullable:
            z = np.concatenate([(p[i][1] if p[i][1] is not None else np.zeros(len(p[i][0]), bool)) for p in parts]) \
                if parts else np.zeros(0, bool)
        if t == T_STRING:       # codes -> byte strings
            dec = np.empty(len(d), dtype=object)
            for j, c in enumerate(d):
                dec[j] = b"" if (z is not None and z[j]) else cur.tree.values[int(c)]
            d = dec
        cols.append((d, z))
    return cur.schema, cols
