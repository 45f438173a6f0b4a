"""Builds operation trees from the golden-vector DSL (tests/golden/reference_tests.json) and
compares an engine's output with the transcribed expectations."""
import json
import math
import os

import numpy as np

import supersonic_amd as ss

HERE = os.path.dirname(os.path.abspath(__file__))
TYPES = {"INT32": ss.INT32, "INT64": ss.INT64, "UINT32": ss.UINT32, "UINT64": ss.UINT64, "FLOAT": ss.FLOAT,
         "DOUBLE": ss.DOUBLE, "BOOL": ss.BOOL, "DATE": ss.DATE, "DATETIME": ss.DATETIME, "STRING": ss.STRING}
AGGS = {"SUM": ss.SUM, "MIN": ss.MIN, "MAX": ss.MAX, "COUNT": ss.COUNT, "FIRST": ss.FIRST, "LAST": ss.LAST, "CONCAT": ss.CONCAT}
ORDERS = {"ASCENDING": ss.ASCENDING, "DESCENDING": ss.DESCENDING}


def load_cases():
    with open(os.path.join(HERE, "golden", "reference_tests.json")) as f:
        return json.load(f)


def _value(v):
    if v == "inf":
        return math.inf
    if v == "nan":
        return math.nan
    return v


def build_view(inp):
    schema = ss.TupleSchema([ss.Attribute(n, TYPES[t], ss.NULLABLE if nullable else ss.NOT_NULLABLE)
                             for n, t, nullable in inp["schema"]])
    cols = []
    for i, (_n, t, nullable) in enumerate(inp["schema"]):
        vals = [r[i] for r in inp["rows"]]
        nulls = np.array([v is None for v in vals], dtype=bool)
        dt = ss.numpy_dtype(TYPES[t])
        if t == "STRING":
            data = ["" if v is None else v for v in vals]
        else:
            py = [0 if v is None else _value(v) for v in vals]
            if np.issubdtype(dt, np.integer) and all(isinstance(v, int) and not isinstance(v, bool) for v in py):
                data = np.array(py, dtype=dt)      # exact for UINT64 values above 2^63
            else:
                data = np.array(py).astype(dt) if vals else np.zeros(0, dt)
        cols.append(ss.Column(data, nulls if nullable else None))
        if not nullable:
            assert not nulls.any()
    return ss.View(schema, cols, len(inp["rows"]))


def build_expr(e):
    if not isinstance(e, list):
        return e
    head, args = e[0], e[1:]
    if head == "CompoundExpression":
        c = ss.CompoundExpression()
        for a in args:
            if a[0] == "AddAs":
                c.AddAs(a[1], build_expr(a[2]))
            else:
                c.Add(build_expr(a[1] if a[0] == "Add" else a))
        return c
    if head == "CaseList":     # CASE arg0 WHEN arg2 THEN arg3 ... ELSE arg1
        return ss.Case([build_expr(a) for a in args])
    if head == "InList":       # needle IN (rest...)
        return ss.In(build_expr(args[0]), [build_expr(a) for a in args[1:]])
    if head == "NullOf":
        return ss.Null(TYPES[args[0]])
    if head == "CastToType":   # CastTo(type, expr)
        return ss.CastTo(TYPES[args[0]], build_expr(args[1]))
    if head in ("AttributeAt", "NamedAttribute") or head.startswith("Const"):
        return getattr(ss, head)(*args)
    return getattr(ss, head)(*[build_expr(a) for a in args])


def build_projector(p):
    if p is None:
        return None
    head, args = p[0], p[1:]
    if head == "Compound":      # (new CompoundSingleSourceProjector)->add(p1)->add(p2)...
        c = ss.CompoundSingleSourceProjector()
        for a in args:
            c.add(build_projector(a))
        return c
    return getattr(ss, head)(*args)


def build_spec(spec):
    s = ss.AggregationSpecification()
    for item in spec:
        distinct = item[0].endswith("_DISTINCT")          # AddDistinctAggregation(...)
        agg = AGGS[item[0][: -len("_DISTINCT")] if distinct else item[0]]
        if len(item) == 4 and distinct:
            s.AddDistinctAggregationWithDefinedOutputType(agg, item[1], item[2], TYPES[item[3]])
        elif len(item) == 4:
            s.AddAggregationWithDefinedOutputType(agg, item[1], item[2], TYPES[item[3]])
        elif distinct:
            s.AddDistinctAggregation(agg, item[1], item[2])
        else:
            s.AddAggregation(agg, item[1], item[2])
    return s


def build_plan(plan, view):
    """view: the case's input View, or a (view, second_view) pair for two-input plans."""
    if plan == "INPUT":
        return ss.ScanView(view[0] if isinstance(view, tuple) else view)
    if plan == "INPUT2":
        return ss.ScanView(view[1])
    head = plan[0]
    if head == "HashJoin":   # [HashJoin, type, lhs key positions, rhs key positions, [[source, projector], ...], uniqueness, lhs, rhs]
        def selector(positions):
            c = ss.CompoundSingleSourceProjector()
            for q in positions:
                c.add(ss.ProjectAttributeAt(q))
            return c
        mp = ss.CompoundMultiSourceProjector()
        for source, pr in plan[4]:
            mp.add(source, build_projector(pr))
        return ss.HashJoin({"INNER": ss.INNER, "LEFT_OUTER": ss.LEFT_OUTER}[plan[1]], selector(plan[2]), selector(plan[3]), mp,
                           {"UNIQUE": ss.UNIQUE, "NOT_UNIQUE": ss.NOT_UNIQUE}[plan[5]], build_plan(plan[6], view), build_plan(plan[7], view))
    if head == "Compute":
        return ss.Compute(build_expr(plan[1]), build_plan(plan[2], view))
    if head == "Filter":
        return ss.Filter(build_expr(plan[1]), build_projector(plan[2]), build_plan(plan[3], view))
    if head == "Project":
        return ss.Project(build_projector(plan[1]), build_plan(plan[2], view))
    if head == "ScalarAggregate":
        return ss.ScalarAggregate(build_spec(plan[1]), build_plan(plan[2], view))
    if head == "BestEffortGroupAggregate":
        options = None
        if len(plan) > 4 and plan[4]:
            options = ss.GroupAggregateOptions()
            if "memory_quota" in plan[4]:
                options.set_memory_quota(plan[4]["memory_quota"])
            if "estimated_result_row_count" in plan[4]:
                options.set_estimated_result_row_count(plan[4]["estimated_result_row_count"])
        return ss.BestEffortGroupAggregate(build_projector(plan[1]), build_spec(plan[2]), options, build_plan(plan[3], view))
    if head == "GroupAggregate":
        options = None
        if len(plan) > 4 and plan[4]:
            options = ss.GroupAggregateOptions().set_max_unique_keys_in_result_(plan[4]["max_unique_keys_in_result"])
        return ss.GroupAggregate(build_projector(plan[1]), build_spec(plan[2]), options, build_plan(plan[3], view))
    if head == "AggregateClusters":
        return ss.AggregateClusters(build_projector(plan[1]), build_spec(plan[2]), build_plan(plan[3], view))
    if head == "Sort":
        order = ss.SortOrder()
        for name, o in plan[1]:
            order.add(name, ORDERS[o])
        return ss.Sort(order, build_projector(plan[2]), 0, build_plan(plan[3], view))
    raise ValueError(head)


def _row_key(row):
    return tuple((0, 0) if v is None else (1, v) for v in row)


def check(case, schema, cols):
    """schema: [(name, type, nullable)], cols: [(data, is_null|None)] -- the engine's output."""
    exp = case["expected"]
    if exp.get("names"):
        assert [s[0] for s in schema] == exp["names"], ([s[0] for s in schema], exp["names"])
    if exp.get("types"):
        assert [s[1] for s in schema] == [TYPES[t] for t in exp["types"]], (schema, exp["types"])
    if exp.get("nullable"):
        assert [bool(s[2]) for s in schema] == exp["nullable"], (schema, exp["nullable"])
    if case["kind"] == "binding":
        return
    n = len(cols[0][0]) if cols else 0
    got = []
    for r in range(n):
        row = []
        for (d, z) in cols:
            if z is not None and z[r]:
                row.append(None)
            else:
                v = d[r].decode() if isinstance(d[r], bytes) else d[r].item()
                row.append(v)
        got.append(row)
    want = [[_value(v) for v in r] for r in exp["rows"]]
    assert len(got) == len(want), "row count %d != %d" % (len(got), len(want))
    if case.get("sorted_on"):   # the guide's sort fixtures: non-decreasing on the key columns, ties in any order (sort.h:42)
        for r in range(1, len(got)):
            assert tuple(got[r - 1][c] for c in case["sorted_on"]) <= tuple(got[r][c] for c in case["sorted_on"]), "row %d out of order" % r
    if not case["ordered"]:
        got, want = sorted(got, key=_row_key), sorted(want, key=_row_key)
    for r, (g, w) in enumerate(zip(got, want)):
        for c, (gv, wv) in enumerate(zip(g, w)):
            if wv is None or gv is None:
                assert gv is None and wv is None, "row %d col %d: got %r want %r" % (r, c, gv, wv)
            elif isinstance(wv, float) and math.isnan(wv):
                assert isinstance(gv, float) and math.isnan(gv), "row %d col %d: got %r want nan" % (r, c, gv)
            elif case.get("max_ulp") and isinstance(gv, float):
                # libm family: the reference's expectations are libm calls; the device libm agrees within a few ULP
                from helpers import ulp_distance
                d = float(ulp_distance(np.array([gv], dtype=np.float64), np.array([float(wv)], dtype=np.float64))[0])
                assert d <= case["max_ulp"], "row %d col %d: got %r want %r (%g ULP)" % (r, c, gv, wv, d)
            elif isinstance(gv, float) or isinstance(wv, float):
                # expected literals are written as in the reference's tests; compare in the column's type
                t = cols[c][0].dtype.type
                assert t(gv) == t(wv), "row %d col %d: got %r want %r" % (r, c, gv, wv)
            else:
                assert gv == wv, "row %d col %d: got %r want %r" % (r, c, gv, wv)
