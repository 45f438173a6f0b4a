"""Chunked execution of a HOST input beyond the ScalarAggregate shape (include/ssgpu.h "CHUNKED STAGING", forms 2 and 3;
csrc/runtime.cpp stream_job_*).  The reference pulls <= 1024-row blocks from any child for every operation (filter.cc:96-128,
aggregate_groups.cc:212-282 ProcessInput); here the rows cross PCIe in chunks while the device works on the chunk before:
 * row-local plans (Filter / Compute / Project): every chunk's result rows are appended, in input order;
 * plans whose first blocking operation is a GroupAggregate: every chunk leaves a partial table, ONE merging plan runs at the end
   (and the operations above the GroupAggregate after it).
Both through ssgpu_plan_run_host (host columns in place) and ssgpu_plan_stream_* (a child cursor's blocks), against the oracle over the
whole input; the shapes that have no chunked form are refused at begin / run_host with the reference's ERROR_NOT_IMPLEMENTED."""
import numpy as np
import pytest

import supersonic_amd as ss
from helpers import assert_cols_equal, sort_rows, to_cols
from oracle import oracle
from test_parity_gpu import group_query, make_view

NA = ss.NamedAttribute


def blocks_of(view, block):
    """The child's ONE output block, overwritten by every Next() (cursor.h:131-148)."""
    n, schema = view.row_count(), view.schema()
    buf = [(np.zeros(block, dtype=view.column(i).data.dtype), None if view.column(i).is_null is None else np.zeros(block, dtype=bool)) for i in range(view.column_count())]
    for lo in range(0, n, block):
        m = min(block, n - lo)
        for i, (d, z) in enumerate(buf):
            d[:m] = view.column(i).data[lo:lo + m]
            if z is not None:
                z[:m] = view.column(i).is_null[lo:lo + m]
        yield ss.View(schema, [ss.Column(d[:m], None if z is None else z[:m]) for (d, z) in buf], m)
        for d, _z in buf:
            d[:m] = 0


def filter_query(view, with_compute):
    child = ss.ScanView(view)
    if with_compute:
        e = (ss.CompoundExpression().Add(NA("a")).Add(NA("b")).AddAs("s", ss.Plus(NA("a"), NA("b"))).Add(NA("d0")).AddAs("p", ss.Multiply(NA("d2"), NA("d3")))
             .Add(NA("t")).Add(NA("u")))
        child = ss.Compute(e, child)
    return ss.Filter(ss.Greater(NA("b"), ss.ConstInt64(499)), ss.ProjectAllAttributes(), child)


# ---- form 2: row-local plans ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("nullable", [False, True])
@pytest.mark.parametrize("with_compute", [False, True])
@pytest.mark.parametrize("n,chunk", [(0, 1000), (1, 1000), (1000, 1000), (1001, 1000), (70001, 4096), (300007, 50000), (300007, 0)])
def test_a_materialising_filter_appends_the_result_rows_of_every_chunk(gpu_ctx, n, chunk, nullable, with_compute):
    view = make_view(n, nullable=nullable)
    op = filter_query(view, with_compute)
    _schema, want = oracle.run(op)
    plan = ss.Plan(op, gpu_ctx)
    assert plan.chunked_form()[0] == 2
    for _ in range(2):                       # (staging sets and the accumulation are reused)
        plan.run_host(chunk_rows=chunk)
        assert_cols_equal(to_cols(plan.fetch()), want, context="filter, chunked n=%d chunk=%d" % (n, chunk))     # rows in input order
    plan.run()                               # the ordinary form still runs on the same plan, and after it the chunked one again
    assert_cols_equal(to_cols(plan.fetch()), want)
    plan.stream(blocks_of(view, 1024), chunk_rows=chunk)
    assert_cols_equal(to_cols(plan.fetch()), want, context="filter, pushed blocks n=%d chunk=%d" % (n, chunk))


@pytest.mark.gpu
def test_a_compute_without_a_filter_is_row_local_too(gpu_ctx):
    view = make_view(100003, nullable=True)
    e = ss.CompoundExpression().AddAs("q", ss.Plus(NA("a"), NA("b"))).AddAs("r", ss.Multiply(NA("d0"), NA("d1"))).Add(NA("k1"))
    op = ss.Compute(e, ss.ScanView(view))
    _schema, want = oracle.run(op)
    plan = ss.Plan(op, gpu_ctx)
    plan.run_host(chunk_rows=7777)
    assert_cols_equal(to_cols(plan.fetch()), want)


@pytest.mark.gpu
def test_an_evaluation_error_of_any_chunk_fails_a_row_local_stream(gpu_ctx):
    n = 10000
    schema = ss.TupleSchema([ss.Attribute("a", ss.INT64), ss.Attribute("b", ss.INT64)])
    b = np.ones(n, dtype=np.int64)
    b[4137] = 0
    view = ss.View(schema, [np.arange(n), b])
    op = ss.Compute(ss.CompoundExpression().AddAs("q", ss.DivideSignaling(NA("a"), NA("b"))), ss.ScanView(view))
    plan = ss.Plan(op, gpu_ctx)
    with pytest.raises(ss.SupersonicException) as e:
        plan.run_host(chunk_rows=1000)
        plan.fetch()
    assert e.value.return_code == ss.ERROR_EVALUATION_ERROR


# ---- form 3: GroupAggregate ------------------------------------------------------------------------------------------------------------
def group_all_functions(view, with_filter, keys):
    spec = ss.AggregationSpecification()
    for col in ["d0", "d1"]:
        spec.AddAggregation(ss.SUM, col, "sum_" + col).AddAggregation(ss.MIN, col, "min_" + col).AddAggregation(ss.MAX, col, "max_" + col)
    (spec.AddAggregation(ss.COUNT, "", "n").AddAggregation(ss.COUNT, "d0", "n_d0").AddAggregation(ss.SUM, "a", "sum_a").AddAggregation(ss.SUM, "u", "sum_u")
         .AddAggregation(ss.MIN, "d", "min_d").AddAggregation(ss.MAX, "t", "max_t").AddAggregation(ss.FIRST, "c", "first_c").AddAggregation(ss.LAST, "c", "last_c")
         .AddAggregation(ss.FIRST, "d0", "first_d0").AddAggregation(ss.SUM, "f", "sum_f"))
    child = ss.ScanView(view)
    if with_filter:
        child = ss.Filter(ss.Greater(NA("b"), ss.ConstInt64(499)), ss.ProjectAllAttributes(), child)
    return ss.GroupAggregate(ss.ProjectNamedAttributes(list(keys)), spec, None, child)


@pytest.mark.gpu
@pytest.mark.parametrize("dense", [0, 1])
@pytest.mark.parametrize("nullable", [False, True])
@pytest.mark.parametrize("with_filter", [False, True])
@pytest.mark.parametrize("n,chunk", [(0, 1000), (1, 1000), (1001, 1000), (70001, 4096), (300007, 50000), (300007, 0)])
def test_group_aggregate_over_chunks_merges_partial_tables(n, chunk, with_filter, nullable, dense):
    """SUM (integer, DOUBLE with residuals, FLOAT), MIN / MAX, COUNT(*) and COUNT(column), FIRST / LAST by global row order, NULL keys and NULL
    inputs, under a Filter: partial tables of every chunk, merged once -- the oracle's rows over the whole input, whatever the chunking and
    whichever shape (hashed / dense slots) the chunks' runs take."""
    ctx = ss.Context(0)
    ctx.set_option("group_dense", dense)
    ctx.set_option("dense_min_rows", 1)
    view = make_view(n, nullable=nullable)
    keys = ("k1",) if nullable else ("k1", "k2")
    op = group_all_functions(view, with_filter, keys)
    _schema, want = oracle.run(op)
    plan = ss.Plan(op, ctx)
    kind, head, tail = plan.chunked_form()
    assert kind == 3 and "$res" in head and "$res" in tail, (kind, head, tail)
    for _ in range(2):
        plan.run_host(chunk_rows=chunk)
        assert_cols_equal(sort_rows(to_cols(plan.fetch())), sort_rows(want), context="group, chunked n=%d chunk=%d" % (n, chunk))
    plan.stream(blocks_of(view, 1024), chunk_rows=chunk)
    assert_cols_equal(sort_rows(to_cols(plan.fetch())), sort_rows(want), context="group, pushed blocks n=%d chunk=%d" % (n, chunk))
    plan.run()
    assert_cols_equal(sort_rows(to_cols(plan.fetch())), sort_rows(want))


@pytest.mark.gpu
def test_the_operations_above_the_group_aggregate_run_after_the_merge(gpu_ctx):
    """Sort(Compute(GroupAggregate(Filter(...)))): the GroupAggregate is chunked, Compute and Sort run once over the merged table -- rows in
    the Sort's order."""
    view = make_view(200003)
    g = group_query(view, True, ("k1", "k2"))
    e = (ss.CompoundExpression().Add(NA("k1")).Add(NA("k2")).AddAs("spread", ss.Minus(NA("max_d0"), NA("min_d0"))).Add(NA("n")).Add(NA("sum_d1")).Add(NA("sum_a")))
    op = ss.Sort(ss.SortOrder().add("k2", ss.DESCENDING).add("k1", ss.ASCENDING), ss.ProjectAllAttributes(), 0, ss.Compute(e, g))
    _schema, want = oracle.run(op)
    plan = ss.Plan(op, gpu_ctx)
    assert plan.chunked_form()[0] == 3
    plan.run_host(chunk_rows=30000)
    assert_cols_equal(to_cols(plan.fetch()), want, context="sorted merge")
    plan.stream(blocks_of(view, 1000), chunk_rows=1 << 15)
    assert_cols_equal(to_cols(plan.fetch()), want, context="sorted merge, pushed")


@pytest.mark.gpu
def test_config3_plan_over_host_chunks(gpu_ctx):
    """BASELINE configs[2]'s plan (2 x INT32 keys, 1e5 groups, 12 DOUBLE aggregates) over a host input in chunks."""
    import bench
    n = 1_000_000
    view = ss.View(bench.group_schema(ss), bench.host_columns(np, "group", n))
    saved = bench.GROUP_FILTER
    try:
        for with_filter in (False, True):
            bench.GROUP_FILTER = with_filter
            op = bench.build_group_plan(ss, view)
            _schema, want = oracle.run(op)
            plan = ss.Plan(op, gpu_ctx)
            plan.run_host(chunk_rows=1 << 18)
            assert_cols_equal(sort_rows(to_cols(plan.fetch())), sort_rows(want), context="config #%d chunked" % (4 if with_filter else 3))
    finally:
        bench.GROUP_FILTER = saved


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["double", "float"])
@pytest.mark.parametrize("chunk", [1000, 777, 5000, 0])
def test_a_nan_that_is_a_groups_first_value_stays_its_min_and_max_across_chunks(gpu_ctx, dtype, chunk):
    """aggregation_operators.h:189-228: the first non-NULL value is assigned, and `val < result` never replaces a NaN -- so a group whose
    FIRST value is a NaN keeps it, a NaN met later is skipped.  Per-chunk partial results skip NaNs; the group's first value travels next
    to them (a hidden FIRST) and the merged result is IF(IS_NAN(first), first, min): the oracle's rows whatever the chunking -- groups
    that start with a NaN (in the first chunk and in a later one), meet one later, hold nothing but NaNs, or nothing but NULLs."""
    n, groups = 5000, 9
    np_t, ss_t = (np.float64, ss.DOUBLE) if dtype == "double" else (np.float32, ss.FLOAT)
    rng = np.random.default_rng(17)
    k = (np.arange(n) % groups).astype(np.int32)
    x = (rng.integers(-1000, 1000, n) * 0.5).astype(np_t)
    nulls = rng.random(n) < 0.1
    first_row = {g: int(np.flatnonzero(k == g)[0]) for g in range(groups)}
    x[first_row[0]] = np.nan; nulls[first_row[0]] = False                    # group 0: starts with a NaN (chunk 0)
    x[first_row[1]] = np.nan; nulls[first_row[1]] = True                     # group 1: its first row is NULL (the NaN under it does not count) ...
    x[np.flatnonzero(k == 1)[1]] = np.nan; nulls[np.flatnonzero(k == 1)[1]] = False   # ... and its first non-NULL value is a NaN
    x[np.flatnonzero(k == 2)[300]] = np.nan; nulls[np.flatnonzero(k == 2)[300]] = False   # group 2: a NaN in a later chunk, not first
    x[k == 3] = np.nan; nulls[k == 3] = False                                # group 3: nothing but NaNs
    nulls[k == 4] = True                                                     # group 4: nothing but NULLs
    g5 = np.flatnonzero(k == 5); nulls[g5[:200]] = True; x[g5[200]] = np.nan; nulls[g5[200]] = False   # group 5: NULLs through the first chunks, then a NaN first
    schema = ss.TupleSchema([ss.Attribute("k", ss.INT32), ss.Attribute("x", ss_t, ss.NULLABLE)])
    view = ss.View(schema, [k, ss.Column(x, nulls)])
    spec = (ss.AggregationSpecification().AddAggregation(ss.MIN, "x", "mn").AddAggregation(ss.MAX, "x", "mx").AddAggregation(ss.COUNT, "x", "c")
            .AddAggregation(ss.FIRST, "x", "f"))
    op = ss.GroupAggregate(ss.ProjectNamedAttribute("k"), spec, None, ss.ScanView(view))
    _schema, want = oracle.run(op)
    plan = ss.Plan(op, gpu_ctx)
    kind, head, tail = plan.chunked_form()
    assert kind == 3 and "mn$first" in head and "mx$first" in head, (head, tail)
    plan.run_host(chunk_rows=chunk)
    assert_cols_equal(sort_rows(to_cols(plan.fetch())), sort_rows(want), context="NaN-first groups, chunk %d" % chunk)
    plan.run()
    assert_cols_equal(sort_rows(to_cols(plan.fetch())), sort_rows(want), context="NaN-first groups, one run")


# ---- which plans (no device needed) ------------------------------------------------------------------------------------------------------
def test_chunked_form_of_a_plan_is_decided_at_bind_time():
    ctx = ss.Context(-1)                     # bind-only
    view = make_view(10)
    spec1 = ss.AggregationSpecification().AddAggregation(ss.SUM, "a", "s")
    assert ss.Plan(ss.ScalarAggregate(spec1, ss.ScanView(view)), ctx).chunked_form()[0] == 1
    assert ss.Plan(filter_query(view, True), ctx).chunked_form()[0] == 2
    kind, head, tail = ss.Plan(group_query(view, True), ctx).chunked_form()
    assert kind == 3
    assert "sum_d0$res" in head and "GroupAggregate" in tail or "group" in tail.lower(), (head, tail)

    def refused(op):
        with pytest.raises(ss.SupersonicException) as e:
            ss.Plan(op, ctx).chunked_form()
        return e.value.return_code
    assert refused(ss.Sort(ss.SortOrder().add("a", ss.ASCENDING), ss.ProjectAllAttributes(), 0, ss.ScanView(view))) == ss.ERROR_NOT_IMPLEMENTED
    distinct = ss.AggregationSpecification().AddDistinctAggregation(ss.COUNT, "a", "n")
    assert refused(ss.GroupAggregate(ss.ProjectNamedAttribute("k1"), distinct, None, ss.ScanView(view))) == ss.ERROR_NOT_IMPLEMENTED
    limited = ss.GroupAggregateOptions().set_max_unique_keys_in_result_(3)
    assert refused(ss.GroupAggregate(ss.ProjectNamedAttribute("k1"), spec1, limited, ss.ScanView(view))) == ss.ERROR_NOT_IMPLEMENTED
    truncating = ss.AggregationSpecification().AddAggregationWithDefinedOutputType(ss.SUM, "d0", "s", ss.INT64)
    assert refused(ss.GroupAggregate(ss.ProjectNamedAttribute("k1"), truncating, None, ss.ScanView(view))) == ss.ERROR_NOT_IMPLEMENTED
    assert refused(ss.AggregateClusters(ss.ProjectNamedAttribute("k1"), spec1, ss.ScanView(view))) == ss.ERROR_NOT_IMPLEMENTED
