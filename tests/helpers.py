"""Shared helpers for the parity tests: run one operation tree through the HIP path
(C ABI) and through the CPU oracle and compare, masking values at NULL rows (values at
NULL result rows are unspecified in the reference: binary_column_computers.h:204-218)."""
import numpy as np

import supersonic_amd as ss
from oracle import oracle


def to_cols(view):
    return [(view.column(i).data, view.column(i).is_null) for i in range(view.column_count())]


def schema_list(schema):
    return [(schema.attribute(i).name(), schema.attribute(i).type(), schema.attribute(i).nullability())
            for i in range(schema.attribute_count())]


def sort_rows(cols):
    """Order-insensitive comparison support: sort rows lexicographically (NULLs first)."""
    if not cols or len(cols[0][0]) == 0:
        return cols
    keys = []
    for d, z in reversed(cols):
        if d.dtype == object:   # STRING column: rank of the byte strings
            uniq = {v: i for i, v in enumerate(sorted(set(d.tolist())))}
            d = np.array([uniq[v] for v in d.tolist()], dtype=np.int64)
        keys.append(np.where(z, 0, d) if z is not None else d)
        if z is not None:
            keys.append(~z)
    order = np.lexsort(keys)
    return [(d[order], None if z is None else z[order]) for d, z in cols]


def ulp_distance(g, w):
    """Distance in units in the last place between two float64 arrays (0 for two NaNs or two equal infinities)."""
    def key(a):
        b = np.ascontiguousarray(a, dtype=np.float64).view(np.int64)
        return np.where(b < 0, np.int64(-(2 ** 63)) - b, b)       # monotone integer image of the doubles
    with np.errstate(over="ignore"):
        d = np.abs(key(g) - key(w)).astype(np.float64)               # exact in int64 (values of like sign and magnitude)
    both_nan = np.isnan(g) & np.isnan(w)
    return np.where(both_nan, 0.0, np.where(np.isnan(g) != np.isnan(w), np.inf, d))


def assert_cols_equal(got, want, float_exact=True, context="", max_ulp=0):
    assert len(got) == len(want), (context, len(got), len(want))
    for i, ((gd, gz), (wd, wz)) in enumerate(zip(got, want)):
        assert len(gd) == len(wd), "%s column %d: %d rows vs %d" % (context, i, len(gd), len(wd))
        gz_ = np.zeros(len(gd), bool) if gz is None else gz
        wz_ = np.zeros(len(wd), bool) if wz is None else wz
        assert np.array_equal(gz_, wz_), "%s column %d: NULL masks differ at %s" % (
            context, i, np.nonzero(gz_ != wz_)[0][:10])
        live = ~wz_
        g, w = gd[live], wd[live]
        if g.dtype == object or w.dtype == object:
            bad = [j for j in range(len(g)) if g[j] != w[j]]
            assert not bad, "%s column %d: %d rows differ, first %s: got %s want %s" % (
                context, i, len(bad), bad[:5], [g[j] for j in bad[:5]], [w[j] for j in bad[:5]])
        elif max_ulp and g.dtype == np.float64:
            d = ulp_distance(g, w)
            bad = np.nonzero(d > max_ulp)[0]
            assert len(bad) == 0, "%s column %d: %d rows beyond %d ULP, first %s: got %s want %s (ULP %s)" % (
                context, i, len(bad), max_ulp, bad[:5], g[bad[:5]], w[bad[:5]], d[bad[:5]])
        elif float_exact or g.dtype.kind != "f":
            same = g.view(np.uint8).reshape(len(g), -1) == w.view(np.uint8).reshape(len(w), -1) if len(g) else np.ones((0, 1), bool)
            bad = np.nonzero(~same.all(axis=1))[0] if len(g) else []
            assert len(bad) == 0, "%s column %d: %d rows differ, first %s: got %s want %s" % (
                context, i, len(bad), bad[:5], g[bad[:5]], w[bad[:5]])
        else:
            assert np.allclose(g, w, rtol=0, atol=0, equal_nan=True)


def run_both(op, ctx, ignore_order=False, max_rows=1024, max_ulp=0, stats=None):
    """stats: a dict that collects which kernels the device run used (plans that held a specialised kernel, GroupAggregate
    stages that ran in a dense-slot shape) -- how tests/test_fuzz_shipped_gpu.py shows what it tested."""
    cur = op.CreateCursor(ctx)
    got_view = ss.drain(cur, max_rows)
    if stats is not None:
        stats["plans"] = stats.get("plans", 0) + 1
        stats["specialized_plans"] = stats.get("specialized_plans", 0) + (1 if cur.plan.specialized() > 0 else 0)
        info = cur.plan.stage_info()
        groups = [st for st in info if st["kind"] == 3]
        stats["group_stages"] = stats.get("group_stages", 0) + len(groups)
        stats["dense_stages"] = stats.get("dense_stages", 0) + sum(1 for st in groups if st["dense_slots"] > 0)
    oschema, want = oracle.run(op, max_rows)
    assert schema_list(cur.schema()) == oschema, (schema_list(cur.schema()), oschema)
    got = to_cols(got_view)
    if ignore_order:
        got, want = sort_rows(got), sort_rows(want)
    assert_cols_equal(got, want, context=str(cur.schema()), max_ulp=max_ulp)
    return got_view
