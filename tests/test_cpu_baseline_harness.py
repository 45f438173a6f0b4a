"""bench.py's N-thread CPU baseline harness (oracle/ss_oracle.c "bench harness", oracle.ThreadedBench): row-range shards on
pthreads + the merge of the partial GroupAggregate tables must give what ONE oracle cursor over all rows gives -- the figure
bench.py prints is only a baseline if the threaded job computes the same result."""
import numpy as np
import pytest

import bench
import supersonic_amd as ss
from oracle import oracle


@pytest.mark.parametrize("threads", [1, 3, 8])
@pytest.mark.parametrize("with_filter", [False, True])
def test_sharded_group_aggregate_merges_to_the_one_cursor_result(threads, with_filter):
    n = 150001
    cols = bench.host_columns(np, "group", n, seed=7)
    saved = bench.GROUP_FILTER
    bench.GROUP_FILTER = with_filter
    try:
        op = bench.build_group_plan(ss, ss.View(bench.group_schema(ss), cols))
    finally:
        bench.GROUP_FILTER = saved
    _schema, want = oracle.run(op)
    r = oracle.ThreadedBench(op).run(threads, 2, cpus=None, merge=True)
    assert r["merged_groups"] == len(want[0][0])
    assert r["merge_seconds"] is not None
    assert r["merged_checksum"] == float(want[2][0].sum())          # SUM(d0): integers, exact in any order


def test_shards_filled_from_a_sample_and_scalar_plans_have_no_merge():
    n = 40000
    cols = bench.host_columns(np, "wide", n, seed=3)
    sample = bench.build_plan(ss, ss.View(bench.bench_schema(ss), cols))
    big = [np.empty(4 * n, dtype=c.dtype) for c in cols]
    op = bench.build_plan(ss, ss.View(bench.bench_schema(ss), big))
    tb = oracle.ThreadedBench(op, sample=sample)
    r = tb.run(4, 3, cpus=None)
    assert r["merge_seconds"] is None and r["result_rows"] == 4          # one row per shard
    for b, c in zip(big, cols):                                          # every thread wrote its own quarter: 4 copies of the sample
        assert np.array_equal(b, np.tile(c, 4))
    assert tb.stream_read(4, 2) > 0


@pytest.mark.parametrize("query", ["sort", "filter_mat"])
def test_row_producing_plans_run_sharded_and_free_their_cursors(query):
    n = 30000
    cols = bench.host_columns(np, query, n, seed=5)
    view = ss.View(bench.bench_schema(ss), cols)
    op = bench.build_sort_plan(ss, view) if query == "sort" else bench.build_filter_mat_plan(ss, view)
    r = oracle.ThreadedBench(op).run(4, 5)
    want = n if query == "sort" else int((cols[0] > bench.K_FILTER).sum())
    assert r["result_rows"] == want
