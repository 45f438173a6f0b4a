"""GPU tests of the seams around the hot path: Expression::Bind -> BoundExpressionTree::Evaluate
(expression/base/expression.h:96-167, expression.cc:57-76), the MemoryLimit allocator behind
Operation::SetBufferAllocator (cursor/base/operation.h:66-76 -> ERROR_MEMORY_EXCEEDED) and STRING columns through the
ABI dictionary.  Results are compared with the CPU oracle evaluating Compute(expression, ScanView(view))."""
import numpy as np
import pytest

import supersonic_amd as ss
from supersonic_amd import _lib as L
from helpers import to_cols, schema_list, assert_cols_equal
from oracle import oracle

pytestmark = pytest.mark.gpu
NA = ss.NamedAttribute


def table(n, seed=3, nullable=True):
    rng = np.random.default_rng(seed)
    N = ss.NULLABLE if nullable else ss.NOT_NULLABLE
    schema = ss.TupleSchema([ss.Attribute("a", ss.INT64, N), ss.Attribute("b", ss.INT32), ss.Attribute("x", ss.DOUBLE, N),
                             ss.Attribute("t", ss.BOOL)])

    def nl():
        return (rng.random(n) < 0.15) if nullable else None
    return ss.View(schema, [ss.Column(rng.integers(-1000, 1000, n), nl()), rng.integers(-50, 50, n).astype(np.int32),
                            ss.Column(rng.normal(size=n) * 100, nl()), rng.integers(0, 2, n).astype(bool)])


EXPRESSIONS = {
    "plus": lambda: ss.Plus(NA("a"), NA("b")),
    "compound": lambda: ss.CompoundExpression().AddAs("s", ss.Multiply(NA("x"), NA("x"))).Add(NA("a"))
                          .AddAs("q", ss.DivideNulling(NA("a"), NA("b"))),
    "if": lambda: ss.If(NA("t"), ss.Negate(NA("a")), ss.CastTo(ss.INT64, NA("b"))),
    "guarded": lambda: ss.If(ss.Equal(NA("b"), ss.ConstInt32(0)), ss.ConstInt32(0), ss.DivideSignaling(ss.ConstInt32(1000), NA("b"))),
    "logic": lambda: ss.Or(ss.IsNull(NA("x")), ss.And(NA("t"), ss.Less(NA("x"), ss.ConstDouble(10.0)))),
}


@pytest.mark.parametrize("name", sorted(EXPRESSIONS))
@pytest.mark.parametrize("n", [0, 1, 65, 1025, 50001])
def test_evaluate_matches_oracle(gpu_ctx, name, n):
    view = table(n)
    bound = EXPRESSIONS[name]().Bind(view.schema(), ss.HeapBufferAllocator(gpu_ctx), max(n, 1), gpu_ctx)
    r = bound.Evaluate(view)
    assert r.has_data(), r.exception()
    oschema, want = oracle.run(ss.Compute(EXPRESSIONS[name](), ss.ScanView(view)), 1 << 20)
    assert schema_list(bound.result_schema) == oschema
    assert r.view().row_count() == n
    assert_cols_equal(to_cols(r.view()), want, context=name)


# ---- the node-level form: DoEvaluate(view, skip_vectors) (expression.h:46-92) --------------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 65, 1025, 50001])
def test_do_evaluate_honours_and_returns_skip_vectors(gpu_ctx, n):
    """One in / out skip vector per result attribute: a skipped row is NULL in the result whatever its inputs; on return the vector holds
    the result's NULLs (the skips of the call included).  Checked against the oracle's Compute over the same View with the skipped rows'
    results nulled; an attribute without a vector (None) is evaluated everywhere."""
    view = table(n)
    rng = np.random.default_rng(n + 7)
    expr = EXPRESSIONS["compound"]
    bound = expr().Bind(view.schema(), None, max(n, 1), gpu_ctx)
    skip = [rng.random(n) < 0.3, None, rng.random(n) < 0.5]
    given = [None if v is None else v.copy() for v in skip]
    r = bound.DoEvaluate(view, skip)
    assert r.has_data(), r.exception()
    _schema, want = oracle.run(ss.Compute(expr(), ss.ScanView(view)), 1 << 20)
    got = to_cols(r.view())
    for i, (data, nulls) in enumerate(want):
        wn = np.zeros(n, dtype=bool) if nulls is None else np.asarray(nulls, dtype=bool).copy()
        if given[i] is not None:
            wn |= given[i]
        gn = np.zeros(n, dtype=bool) if got[i][1] is None else np.asarray(got[i][1], dtype=bool)
        assert np.array_equal(gn, wn), "attribute %d: NULLs" % i
        keep = ~wn
        assert np.array_equal(np.asarray(got[i][0])[keep], np.asarray(data)[keep]), "attribute %d: values" % i
        if skip[i] is not None:
            assert np.array_equal(skip[i], wn), "attribute %d: the vector on return" % i


def test_do_evaluate_does_not_fail_on_skipped_rows(gpu_ctx):
    """A signalling division by zero fails the evaluation (expression.cc:57-76) -- unless every offending row is skipped."""
    n = 5000
    schema = ss.TupleSchema([ss.Attribute("a", ss.INT64), ss.Attribute("b", ss.INT64)])
    b = np.ones(n, dtype=np.int64)
    bad = np.zeros(n, dtype=bool)
    bad[[7, 1234, 4999]] = True
    b[bad] = 0
    view = ss.View(schema, [np.arange(n, dtype=np.int64), b])
    bound = ss.DivideSignaling(NA("a"), NA("b")).Bind(schema, None, n, gpu_ctx)
    r = bound.Evaluate(view)
    assert r.is_failure() and r.exception().return_code == L.ERROR_EVALUATION_ERROR
    skip = [bad.copy()]
    r = bound.DoEvaluate(view, skip)
    assert r.has_data(), r.exception()
    data, nulls = to_cols(r.view())[0]
    assert np.array_equal(np.asarray(nulls, dtype=bool), bad)
    assert np.array_equal(np.asarray(data)[~bad], np.arange(n)[~bad])
    r = bound.DoEvaluate(view, [np.zeros(n, dtype=bool)])         # nothing skipped: the failure is back
    assert r.is_failure() and r.exception().return_code == L.ERROR_EVALUATION_ERROR
    r = bound.DoEvaluate(view, [None, None])                       # one vector per attribute
    assert r.is_failure() and r.exception().return_code == L.ERROR_ATTRIBUTE_COUNT_MISMATCH


def test_evaluate_reuses_the_bound_tree_over_many_views(gpu_ctx):
    # a bound tree is evaluated block after block (what ComputeCursor::Next does, compute.cc:74-90)
    bound = EXPRESSIONS["compound"]().Bind(table(1).schema(), None, 4096, gpu_ctx)
    for seed, n in [(1, 4096), (2, 17), (3, 4096), (4, 0), (5, 1000)]:
        view = table(n, seed)
        r = bound.Evaluate(view)
        assert r.has_data(), r.exception()
        _, want = oracle.run(ss.Compute(EXPRESSIONS["compound"](), ss.ScanView(view)), 1 << 20)
        assert_cols_equal(to_cols(r.view()), want, context="seed %d" % seed)


def test_evaluate_beyond_row_capacity_is_too_many_rows(gpu_ctx):
    view = table(100)
    bound = EXPRESSIONS["plus"]().Bind(view.schema(), None, 64, gpu_ctx)
    assert bound.row_capacity() == 64
    r = bound.Evaluate(view)
    assert r.is_failure() and r.exception().return_code == L.ERROR_TOO_MANY_ROWS
    small = table(64)
    assert bound.Evaluate(small).has_data()              # the tree stays usable


def test_evaluate_reports_evaluation_errors(gpu_ctx):
    schema = ss.TupleSchema([ss.Attribute("b", ss.INT32)])
    view = ss.View(schema, [np.array([4, 2, 0, 1], np.int32)])
    bound = ss.DivideSignaling(ss.ConstInt32(8), NA("b")).Bind(schema, None, 16, gpu_ctx)
    r = bound.Evaluate(view)
    assert r.is_failure() and r.exception().return_code == L.ERROR_EVALUATION_ERROR
    ok = ss.View(schema, [np.array([4, 2, 8, 1], np.int32)])
    r = bound.Evaluate(ok)
    assert r.has_data() and list(r.view().column(0).data) == [2.0, 4.0, 1.0, 8.0]


def test_evaluate_string_columns_through_the_dictionary(gpu_ctx):
    schema = ss.TupleSchema([ss.Attribute("s", ss.STRING, ss.NULLABLE), ss.Attribute("k", ss.INT32)])
    bound = (ss.CompoundExpression().AddAs("lt", ss.Less(NA("s"), ss.ConstString("m")))
             .AddAs("pick", ss.If(ss.Greater(NA("k"), ss.ConstInt32(1)), NA("s"), ss.ConstString("zz")))).Bind(schema, None, 0, gpu_ctx)
    for words in (["apple", "m", "zebra", "ma", ""], ["n", "l", "m", "m\x00", "k"]):
        nulls = np.array([False, False, True, False, False])
        view = ss.View(schema, [ss.Column(np.array([w.encode() for w in words], dtype=object), nulls),
                                np.array([0, 1, 2, 3, 4], np.int32)])
        r = bound.Evaluate(view)
        assert r.has_data(), r.exception()
        _, want = oracle.run(ss.Compute(
            ss.CompoundExpression().AddAs("lt", ss.Less(NA("s"), ss.ConstString("m")))
            .AddAs("pick", ss.If(ss.Greater(NA("k"), ss.ConstInt32(1)), NA("s"), ss.ConstString("zz"))), ss.ScanView(view)), 1 << 20)
        assert_cols_equal(to_cols(r.view()), want, context=str(words))


def group_op(view):
    spec = ss.AggregationSpecification().AddAggregation(ss.SUM, "a", "sa").AddAggregation(ss.COUNT, "", "n")
    return ss.GroupAggregate(ss.ProjectNamedAttribute("b"), spec, None, ss.ScanView(view))


def test_memory_limit_surfaces_memory_exceeded(gpu_ctx):
    view = table(200000, nullable=False)
    # generous quota: runs, and the plan reports what it holds
    op = group_op(view)
    op.SetBufferAllocator(ss.MemoryLimit(1 << 30, gpu_ctx), True)
    cur = op.CreateCursor(gpu_ctx)
    got = ss.drain(cur)
    assert got.row_count() == 100
    held = cur.plan.memory_in_use()
    assert 0 < held <= (1 << 30)
    # a quota below what the plan needs: ERROR_MEMORY_EXCEEDED from Next(), not a crash
    # (aggregate_groups.cc:372-402 returns it when the allocator refuses the block)
    op = group_op(view)
    op.SetBufferAllocator(ss.MemoryLimit(4096, gpu_ctx), True)
    r = op.CreateCursor(gpu_ctx).Next()
    assert r.is_failure() and r.exception().return_code == L.ERROR_MEMORY_EXCEEDED
    # Sort and a materialising Filter allocate their outputs under the quota too
    for mk in (lambda: ss.Sort(ss.SortOrder().add("a", ss.ASCENDING), None, 0, ss.ScanView(view)),
               lambda: ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(0)), ss.ProjectAllAttributes(), ss.ScanView(view))):
        op = mk()
        op.SetBufferAllocator(ss.MemoryLimit(65536, gpu_ctx), True)
        r = op.CreateCursor(gpu_ctx).Next()
        assert r.is_failure() and r.exception().return_code == L.ERROR_MEMORY_EXCEEDED, mk
        op = mk()
        op.SetBufferAllocator(ss.HeapBufferAllocator(gpu_ctx), True)
        assert op.CreateCursor(gpu_ctx).Next().has_data()


def test_memory_limit_can_be_raised_after_a_failure(gpu_ctx):
    view = table(100000, nullable=False)
    plan = ss.Plan(group_op(view), gpu_ctx)
    plan.set_memory_limit(1024)
    with pytest.raises(ss.SupersonicException) as e:
        plan.run()
    assert e.value.return_code == L.ERROR_MEMORY_EXCEEDED
    assert plan.memory_in_use() <= 1024
    plan.set_memory_limit(None)
    plan.run()
    got = plan.fetch()
    _, want = oracle.run(group_op(view), 1 << 20)
    from helpers import sort_rows
    assert_cols_equal(sort_rows(to_cols(got)), sort_rows(want))


def test_pinned_allocator_buffers_feed_uploads(gpu_ctx):
    import ctypes as C
    a = ss.MemoryLimit(1 << 20, gpu_ctx)
    n = 4096
    p, g = a.Allocate(n * 8)
    arr = np.ctypeslib.as_array((C.c_int64 * n).from_address(p))
    arr[:] = np.arange(n)
    schema = ss.TupleSchema([ss.Attribute("v", ss.INT64)])
    view = ss.View(schema, [arr])
    spec = ss.AggregationSpecification().AddAggregation(ss.SUM, "v", "s")
    got = ss.drain(ss.ScalarAggregate(spec, ss.ScanView(view)).CreateCursor(gpu_ctx))
    assert int(got.column(0).data[0]) == n * (n - 1) // 2
    del view, arr
    a.Free(p)
    assert a.GetUsage() == 0
