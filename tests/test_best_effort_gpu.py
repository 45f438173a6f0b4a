"""BestEffortGroupAggregate on the device (cursor/core/aggregate.h:230-250; GroupAggregateCursor with best_effort_,
aggregate_groups.cc:211-222,332-433) against the oracle's restatement: a view aggregates the longest run of input rows, starting
where the last view stopped, whose keys fit the result block (GroupAggregateOptions::memory_quota / bytes of a result row groups).
The reference's own two vectors (aggregate_groups_test.cc:601-647) run through tests/test_golden_gpu.py; here: many views,
exact order, the per-view key-uniqueness contract, and "an input of any size never raises ERROR_MEMORY_EXCEEDED"."""
import numpy as np
import pytest

import supersonic_amd as ss
from supersonic_amd import _lib as L
from helpers import assert_cols_equal, run_both, to_cols
from oracle import oracle

pytestmark = pytest.mark.gpu
NA = ss.NamedAttribute


def table(n, groups, seed, nullable_key=False):
    rng = np.random.default_rng(seed)
    schema = ss.TupleSchema([ss.Attribute("k", ss.INT32, ss.NULLABLE if nullable_key else ss.NOT_NULLABLE), ss.Attribute("j", ss.INT64),
                             ss.Attribute("v", ss.INT64, ss.NULLABLE), ss.Attribute("d", ss.DOUBLE)])
    return ss.View(schema, [ss.Column(rng.integers(0, groups, n).astype(np.int32), (rng.random(n) < 0.1) if nullable_key else None),
                            ss.Column(rng.integers(0, 3, n)), ss.Column(rng.integers(-1000, 1000, n), rng.random(n) < 0.2),
                            ss.Column(rng.integers(-4000, 4000, n) * 0.25)])


def spec():
    return (ss.AggregationSpecification().AddAggregation(ss.SUM, "v", "sum_v").AddAggregation(ss.COUNT, "", "rows")
            .AddAggregation(ss.MIN, "d", "min_d").AddAggregation(ss.MAX, "v", "max_v").AddAggregation(ss.FIRST, "v", "first_v")
            .AddAggregation(ss.LAST, "d", "last_d"))


def best_effort(view, keys, quota, child=None):
    options = ss.GroupAggregateOptions().set_memory_quota(quota) if quota is not None else None
    return ss.BestEffortGroupAggregate(ss.ProjectNamedAttributes(keys), spec(), options, child if child is not None else ss.ScanView(view))


# a result row of (k, sum_v, rows, min_d, max_v, first_v, last_d): 4 + 8+1 + 8 + 8+1 + 8+1 + 8+1 + 8+1 = 57 bytes (+1 if k is NULLABLE)
@pytest.mark.parametrize("n,groups,capacity", [(1537, 40, 1), (1537, 40, 7), (20011, 300, 64), (20011, 5000, 1000), (70001, 90, 89), (70001, 90, 90),
                                                (200003, 30000, 4096)])
@pytest.mark.parametrize("nullable_key", [False, True])
def test_views_match_the_oracle_row_for_row(gpu_ctx, n, groups, capacity, nullable_key):
    view = table(n, groups, n + capacity, nullable_key)
    op = best_effort(view, ["k"], capacity * (58 if nullable_key else 57))
    got = run_both(op, gpu_ctx)          # first-seen order inside a view, views in input order: compared in order
    assert got.row_count() >= min(groups, n) * (1 if capacity >= groups else 0)


def test_two_keys_below_a_filter_and_a_compute(gpu_ctx):
    view = table(50021, 200, 5)
    child = ss.Filter(ss.Greater(NA("v"), ss.ConstInt64(-500)), ss.ProjectAllAttributes(),
                      ss.Compute(ss.CompoundExpression().Add(NA("k")).Add(NA("j")).AddAs("v", ss.Plus(NA("v"), NA("j"))).Add(NA("d")), ss.ScanView(view)))
    run_both(best_effort(view, ["k", "j"], 300 * 65, child), gpu_ctx)


def test_rows_are_key_unique_within_each_returned_view(gpu_ctx):
    view = table(30011, 500, 11)
    cur = best_effort(view, ["k"], 57 * 100).CreateCursor(gpu_ctx)
    views, total = 0, 0
    while True:
        r = cur.Next(1 << 40)
        assert not r.is_failure(), r.exception()
        if r.is_eos():
            break
        keys = r.view().column(0).data
        assert len(np.unique(keys)) == len(keys) <= 100
        views += 1
        total += int(r.view().column(2).data.sum())      # COUNT(*)
    assert views > 3 and total == 30011


def test_without_a_quota_it_is_the_group_aggregate(gpu_ctx):
    view = table(40009, 700, 13)
    got = ss.drain(best_effort(view, ["k"], None).CreateCursor(gpu_ctx))
    _schema, want = oracle.run(ss.GroupAggregate(ss.ProjectNamedAttributes(["k"]), spec(), None, ss.ScanView(view)))
    assert_cols_equal(to_cols(got), want)               # (both in first-seen order)


def test_empty_input_and_no_group_by_columns(gpu_ctx):
    run_both(best_effort(table(0, 5, 1), ["k"], 57 * 4), gpu_ctx)
    run_both(best_effort(table(3000, 5, 2), [], 1000), gpu_ctx)


def test_distinct_is_refused_at_bind_time(gpu_ctx):
    view = table(100, 5, 3)
    op = ss.BestEffortGroupAggregate(ss.ProjectNamedAttributes(["k"]), ss.AggregationSpecification().AddDistinctAggregation(ss.COUNT, "v", "c"),
                                     None, ss.ScanView(view))
    with pytest.raises(ss.SupersonicException) as e:
        ss.Plan(op, gpu_ctx)
    assert e.value.return_code == L.ERROR_NOT_IMPLEMENTED


def test_an_input_the_group_aggregate_cannot_hold_never_raises_memory_exceeded(gpu_ctx):
    """aggregate.h:243-245: "ERROR_MEMORY_EXCEEDED is not returned when the input is too large".  Under a MemoryLimit the plain
    GroupAggregate fails; the best-effort form halves its input window until a view fits, and the views' partial rows -- shuffled by
    key and aggregated once more, the use the reference names -- give the full result."""
    n, groups = 400003, 100000
    view = table(n, groups, 17)
    limit = 8 << 20
    plain = ss.GroupAggregate(ss.ProjectNamedAttributes(["k"]), spec(), None, ss.ScanView(view))
    plain.SetBufferAllocator(ss.MemoryLimit(limit, gpu_ctx), True)
    r = plain.CreateCursor(gpu_ctx).Next()
    assert r.is_failure() and r.exception().return_code == L.ERROR_MEMORY_EXCEEDED
    op = best_effort(view, ["k"], None)
    op.SetBufferAllocator(ss.MemoryLimit(limit, gpu_ctx), True)
    parts = ss.drain(op.CreateCursor(gpu_ctx))
    assert parts.row_count() > groups * 0.9
    k, s, c = parts.column(0).data, parts.column(1), parts.column(2).data
    want_rows = np.bincount(view.column(0).data, minlength=groups)
    assert np.array_equal(np.bincount(k, weights=c, minlength=groups).astype(np.int64), want_rows)
    v = view.column(2)
    want_sum = np.bincount(view.column(0).data, weights=np.where(v.is_null, 0, v.data), minlength=groups).astype(np.int64)
    got_sum = np.bincount(k, weights=np.where(s.is_null, 0, s.data), minlength=groups).astype(np.int64)
    assert np.array_equal(got_sum, want_sum)
