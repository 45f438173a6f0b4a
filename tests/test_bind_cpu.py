"""Host logic on CPU: the product's binder (libssgpu.so, bind-only context -- no device, no data
path) must give every golden case the reference's result names, types, nullability and bind
errors, and must agree with the oracle's independent binder on the parity-test plans."""
import pytest

import supersonic_amd as ss
from golden_runner import TYPES, build_plan, build_view, load_cases
from helpers import schema_list
from oracle import oracle

CASES = load_cases()
EVAL_ERRORS = (104,)


@pytest.fixture(scope="module")
def bind_ctx():
    return ss.Context(-1)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_binder_matches_reference(bind_ctx, case):
    view = build_view(case["input"])
    if case.get("input2"):
        view = (view, build_view(case["input2"]))
    op = build_plan(case["plan"], view)
    if case["expect_error"] and case["expect_error"] not in EVAL_ERRORS:
        with pytest.raises(ss.SupersonicException) as e:
            ss.Plan(op, bind_ctx)
        # -1: the reference's test only requires the bind to FAIL (TestBoundFactoryFailure): any bind-time code (4xx)
        assert (400 <= e.value.return_code < 500) if case["expect_error"] == -1 else e.value.return_code == case["expect_error"]
        return
    plan = ss.Plan(op, bind_ctx)
    schema = schema_list(plan.result_schema)
    exp = case["expected"]
    if exp.get("names"):
        assert [s[0] for s in schema] == exp["names"]
    if exp.get("types"):
        assert [s[1] for s in schema] == [TYPES[t] for t in exp["types"]]
    if exp.get("nullable"):
        assert [bool(s[2]) for s in schema] == exp["nullable"]
    # and the oracle's independent binder agrees on the whole schema
    assert schema == oracle.Cursor(op).schema


def test_running_without_a_device_fails_loudly(bind_ctx):
    import numpy as np
    view = ss.View(ss.TupleSchema([ss.Attribute("a", ss.INT64)]), [np.arange(4)])
    cur = ss.Compute(ss.Plus(ss.NamedAttribute("a"), ss.ConstInt64(1)), ss.ScanView(view)).CreateCursor(bind_ctx)
    assert cur.schema().attribute(0).name() == "(a + CONST_INT64)"
    r = cur.Next(1024)
    assert r.is_failure() and r.exception().return_code == 2000   # no CPU fallback behind the ABI


def test_bind_errors(bind_ctx):
    import numpy as np
    view = ss.View(ss.TupleSchema([ss.Attribute("a", ss.INT64), ss.Attribute("t", ss.BOOL)]), [np.arange(4), np.zeros(4, bool)])
    NA = ss.NamedAttribute

    def code(op):
        with pytest.raises(ss.SupersonicException) as e:
            ss.Plan(op, bind_ctx)
        return e.value.return_code

    scan = lambda: ss.ScanView(view)  # noqa: E731
    assert code(ss.Compute(NA("zz"), scan())) == 403                                   # ERROR_ATTRIBUTE_MISSING
    assert code(ss.Compute(ss.AttributeAt(7), scan())) == 401                          # ERROR_ATTRIBUTE_COUNT_MISMATCH
    assert code(ss.Compute(ss.And(NA("a"), NA("t")), scan())) == 402                   # ERROR_ATTRIBUTE_TYPE_MISMATCH
    assert code(ss.Filter(NA("a"), ss.ProjectAllAttributes(), scan())) == 402          # predicate must be BOOL (filter.cc:84-92)
    assert code(ss.Filter(ss.CompoundExpression().Add(NA("t")).Add(NA("a")), ss.ProjectAllAttributes(), scan())) == 401
    assert code(ss.Compute(ss.CompoundExpression().Add(NA("a")).AddAs("a", NA("t")), scan())) == 404   # duplicate name
    spec = ss.AggregationSpecification().AddAggregation(ss.SUM, "nope", "s")
    assert code(ss.ScalarAggregate(spec, scan())) == 403                                # aggregator.cc:134-140
    spec = ss.AggregationSpecification().AddAggregation(ss.SUM, "a", "s").AddAggregation(ss.MAX, "a", "s")
    assert code(ss.ScalarAggregate(spec, scan())) == 404                                # aggregator.cc:153-158
    spec = ss.AggregationSpecification().AddAggregation(ss.SUM, "t", "s")
    assert code(ss.ScalarAggregate(spec, scan())) == 405                                # column_aggregator.cc:549-556
    assert code(ss.Compute(ss.CastTo(ss.INT32, ss.Plus(NA("a"), ss.ConstDouble(1.0))), scan())) == 402   # float -> int cast


def test_max_unique_keys_in_result_binds_as_aggregate_sort_fold():
    # GroupAggregateOptions::max_unique_keys_in_result (aggregate.h:160-205, row_hash_set.cc:500-511): hash aggregate with a
    # hidden first-seen row id, sort by it, fold of the rows beyond the limit -- the hidden column is not in the result schema;
    # FIRST / LAST under a limit fold by a hidden row-id twin each (also not in the result schema); DISTINCT and CONCAT under a limit re-key the rows by their result row; a CONCAT result below another operation is refused
    import numpy as np
    schema = ss.TupleSchema([ss.Attribute("k", ss.INT64), ss.Attribute("v", ss.INT64)])
    view = ss.View(schema, [np.arange(4), np.arange(4)])
    opts = ss.GroupAggregateOptions().set_max_unique_keys_in_result_(2)
    op = ss.GroupAggregate(ss.ProjectNamedAttributes(["k"]), ss.AggregationSpecification().AddAggregation(ss.SUM, "v", "s"), opts, ss.ScanView(view))
    plan = ss.Plan(op, ss.Context(-1))
    rs = plan.result_schema
    assert [rs.attribute(i).name() for i in range(rs.attribute_count())] == ["k", "s"]
    assert "fold beyond 2 keys" in plan.describe()
    fl = ss.GroupAggregate(ss.ProjectNamedAttributes(["k"]), ss.AggregationSpecification().AddAggregation(ss.FIRST, "v", "f").AddAggregation(ss.LAST, "v", "l"),
                           ss.GroupAggregateOptions().set_max_unique_keys_in_result_(2), ss.ScanView(view))
    rs = ss.Plan(fl, ss.Context(-1)).result_schema
    assert [rs.attribute(i).name() for i in range(rs.attribute_count())] == ["k", "f", "l"]
    # DISTINCT under a limit: the rows are re-keyed by their result row ($rank, hidden) -- one seen-value set per result row
    ds = ss.GroupAggregate(ss.ProjectNamedAttributes(["k"]), ss.AggregationSpecification().AddDistinctAggregation(ss.COUNT, "v", "c").AddAggregation(ss.LAST, "v", "l"),
                           ss.GroupAggregateOptions().set_max_unique_keys_in_result_(2), ss.ScanView(view))
    dplan = ss.Plan(ds, ss.Context(-1))
    rs = dplan.result_schema
    assert [(rs.attribute(i).name(), rs.attribute(i).is_nullable()) for i in range(rs.attribute_count())] == [("k", False), ("c", False), ("l", True)]
    assert "result row of every input row under the limit 2" in dplan.describe()
    cc = ss.GroupAggregate(ss.ProjectNamedAttributes(["k"]), ss.AggregationSpecification().AddAggregation(ss.CONCAT, "v", "f"),
                           ss.GroupAggregateOptions().set_max_unique_keys_in_result_(2), ss.ScanView(view))
    cplan = ss.Plan(cc, ss.Context(-1))
    rs = cplan.result_schema
    assert [(rs.attribute(i).name(), rs.attribute(i).type()) for i in range(rs.attribute_count())] == [("k", ss.INT64), ("f", ss.STRING)]
    assert "sort by (result row, row id)" in cplan.describe()
    bad = ss.Sort(ss.SortOrder().add("k", ss.ASCENDING), None, 0,
                  ss.GroupAggregate(ss.ProjectNamedAttributes(["k"]), ss.AggregationSpecification().AddAggregation(ss.CONCAT, "v", "f").AddDistinctAggregation(ss.SUM, "v", "s"),
                                    ss.GroupAggregateOptions().set_max_unique_keys_in_result_(2), ss.ScanView(view)))
    with pytest.raises(ss.SupersonicException) as e:
        ss.Plan(bad, ss.Context(-1))
    assert e.value.return_code == ss.ERROR_NOT_IMPLEMENTED


def test_first_last_next_to_distinct_pick_by_the_stored_row_id():
    # FIRST / LAST follow the input order (aggregation_operators.h:290-320); the DISTINCT shape aggregates rows sorted by
    # (keys, distinct column): the input row id rides along as a column and the aggregates pick by it
    import numpy as np
    schema = ss.TupleSchema([ss.Attribute("k", ss.INT32), ss.Attribute("a", ss.INT64), ss.Attribute("b", ss.INT64)])
    view = ss.View(schema, [np.zeros(4, np.int32), np.arange(4), np.arange(4)])
    spec = ss.AggregationSpecification().AddDistinctAggregation(ss.SUM, "a", "s").AddAggregation(ss.FIRST, "b", "f")
    plan = ss.Plan(ss.GroupAggregate(ss.ProjectNamedAttributes(["k"]), spec, None, ss.ScanView(view)), ss.Context(-1))
    rs = plan.result_schema
    assert [rs.attribute(i).name() for i in range(rs.attribute_count())] == ["k", "s", "f"]
    assert "ROWID_64" in plan.describe() and "KEY_APPEND_64" in plan.describe()
    scalar = ss.Plan(ss.ScalarAggregate(spec, ss.ScanView(view)), ss.Context(-1))
    first = [ln for ln in scalar.describe().splitlines() if "AGG_FIRST_64" in ln]
    assert "ROWID_64" in scalar.describe() and len(first) == 1 and "d=r" in first[0], scalar.describe()


def test_distinct_inside_aggregate_clusters_binds_with_a_segment_id_column():
    # materialise (keys, inputs, the cluster's number from the boundary scan) -> sort by (number, DISTINCT column) -> flags ->
    # clusters of (number, keys); the number is projected away behind the aggregate
    import numpy as np
    schema = ss.TupleSchema([ss.Attribute("k", ss.INT32), ss.Attribute("a", ss.INT64), ss.Attribute("b", ss.INT64)])
    view = ss.View(schema, [np.zeros(4, np.int32), np.arange(4), np.arange(4)])
    spec = ss.AggregationSpecification().AddDistinctAggregation(ss.SUM, "a", "s").AddAggregation(ss.MAX, "b", "m")
    plan = ss.Plan(ss.AggregateClusters(ss.ProjectNamedAttributes(["k"]), spec, ss.ScanView(view)), ss.Context(-1))
    rs = plan.result_schema
    assert [rs.attribute(i).name() for i in range(rs.attribute_count())] == ["k", "s", "m"]
    assert "segment ids" in plan.describe()
    # CONCAT next to it: the rows are sorted back into input order, the strings come from the aggregation's stage behind the projection
    both = ss.AggregateClusters(ss.ProjectNamedAttributes(["k"]), ss.AggregationSpecification().AddDistinctAggregation(ss.SUM, "a", "s").AddAggregation(ss.CONCAT, "b", "c"),
                                ss.ScanView(view))
    rs = ss.Plan(both, ss.Context(-1)).result_schema
    assert [(rs.attribute(i).name(), rs.attribute(i).type()) for i in range(rs.attribute_count())] == [("k", ss.INT32), ("s", ss.INT64), ("c", ss.STRING)]


def test_a_pipeline_over_more_input_arrays_than_the_kernel_takes_is_refused():
    """VM_MAX_STAGED (csrc/vm.h) input arrays -- columns and NULL masks -- per pipeline; beyond them ssgpu_plan_create says so instead of
    writing past the kernel's argument block."""
    import numpy as np
    ctx = ss.Context(-1)

    def compute_over(n_in):
        schema = ss.TupleSchema([ss.Attribute("c%d" % i, ss.INT32, ss.NULLABLE) for i in range(n_in)])     # (narrow columns: the LDS bound comes later)
        view = ss.View(schema, [ss.Column(np.zeros(3, dtype=np.int32), np.zeros(3, dtype=bool)) for _ in range(n_in)])
        e = ss.CompoundExpression()
        for i in range(0, n_in - 1, 2):
            e.AddAs("s%d" % i, ss.Plus(ss.NamedAttribute("c%d" % i), ss.NamedAttribute("c%d" % (i + 1))))
        return ss.Plan(ss.Compute(e, ss.ScanView(view)), ctx)
    compute_over(40)                           # 80 arrays: fits
    with pytest.raises(ss.SupersonicException) as err:
        compute_over(42)                       # 84
    assert err.value.return_code == ss.ERROR_NOT_IMPLEMENTED and "input arrays" in str(err.value)
