"""Parity fuzz (GPU): seeded random plans -- expression trees over every operator family on the device
path, under Compute / Filter / ScalarAggregate / GroupAggregate -- must produce the oracle's result
bit for bit (fuzz_plans.py keeps to operators whose results are bit-defined).  1537 rows = three full
512-row tiles and a partial one."""
import os

import pytest

import supersonic_amd as ss
from oracle import oracle
from helpers import run_both
from fuzz_plans import Gen, make_view

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(int(os.environ.get("SS_FUZZ_SEEDS", "2000"))))   # SS_FUZZ_SEEDS=20000 for a longer hunt
def test_random_plan_matches_oracle(gpu_ctx, seed):
    view = make_view(1537, 1000 + seed)
    op, ordered = Gen(seed).plan(view)
    try:
        oracle.run(op)
    except oracle.OracleError:
        with pytest.raises(ss.SupersonicException):
            ss.Plan(op, gpu_ctx)
        return
    run_both(op, gpu_ctx, ignore_order=not ordered)


@pytest.mark.parametrize("seed", range(2000, 2200))
def test_random_plan_matches_oracle_many_tiles(gpu_ctx, seed):
    # 70001 rows: more tiles than one workgroup round, several workgroups per CU, a ragged last tile
    view = make_view(70001, seed)
    op, ordered = Gen(seed).plan(view)
    try:
        oracle.run(op)
    except oracle.OracleError:
        return
    run_both(op, gpu_ctx, ignore_order=not ordered)


@pytest.mark.parametrize("seed", range(3000, 3080))
@pytest.mark.parametrize("partition", [0, 2])
def test_random_high_cardinality_group_aggregate(seed, partition):
    # ~50 k groups out of 70001 rows: the group table outgrows its first capacity (regrow + rerun) on the direct
    # path, and the hash-partitioned execution (forced with group_partition = 2) runs with random aggregates
    ctx = ss.Context(0)
    ctx.set_option("group_partition", partition)
    view = make_view(70001, seed)
    g = Gen(seed)
    op = g.aggregate_plan(view, True)
    op = ss.GroupAggregate(ss.ProjectNamedAttributes(g.pick([["a", "b"], ["a", "k1"], ["w"]])), op.spec, None, op.child)
    try:
        oracle.run(op)
    except oracle.OracleError:
        return
    run_both(op, ctx, ignore_order=True)


@pytest.mark.parametrize("n", [1537, 20011])
@pytest.mark.parametrize("seed", range(4000, 4000 + int(os.environ.get("SS_FUZZ_ORDERED_SEEDS", "250"))))
def test_random_ordered_aggregates(gpu_ctx, seed, n):
    # the round-4 shapes: DISTINCT next to FIRST / LAST (scalar, grouped, clustered), key limits with FIRST / LAST and keys of any
    # width, DISTINCT inside AggregateClusters -- every one carries the input order along as a stored column
    view = make_view(n, seed)
    op, ordered = Gen(seed).ordered_aggregate_plan(view)
    try:
        oracle.run(op)
    except oracle.OracleError:
        with pytest.raises(ss.SupersonicException):
            ss.Plan(op, gpu_ctx)
        return
    run_both(op, gpu_ctx, ignore_order=not ordered)


@pytest.mark.parametrize("n", [1537, 20011])
@pytest.mark.parametrize("seed", range(6000, 6000 + int(os.environ.get("SS_FUZZ_LIMIT_SEEDS", "200"))))   # SS_FUZZ_LIMIT_SEEDS=3000 for a longer hunt
def test_random_distinct_aggregates_under_a_key_limit(gpu_ctx, seed, n):
    # DISTINCT under max_unique_keys_in_result: one seen-value set per RESULT row (column_aggregator.cc:308-376 over the row index
    # row_hash_set.cc:500-511 answers) -- the device stores every input row's result row and aggregates by it
    view = make_view(n, seed)
    op, ordered = Gen(seed).distinct_limit_plan(view)
    try:
        oracle.run(op)
    except oracle.OracleError:
        with pytest.raises(ss.SupersonicException):
            ss.Plan(op, gpu_ctx)
        return
    run_both(op, gpu_ctx, ignore_order=not ordered)


@pytest.mark.parametrize("n", [1537, 20011])
@pytest.mark.parametrize("seed", range(5000, 5000 + int(os.environ.get("SS_FUZZ_SEQ_SEEDS", "150"))))
def test_random_sequential_sums(gpu_ctx, seed, n):
    # SUM of floating inputs into integer results: the reference's row-after-row arithmetic, bit for bit
    view = make_view(n, seed)
    op, ordered = Gen(seed).sequential_sum_plan(view)
    try:
        oracle.run(op)
    except oracle.OracleError:
        with pytest.raises(ss.SupersonicException):
            ss.Plan(op, gpu_ctx)
        return
    run_both(op, gpu_ctx, ignore_order=not ordered)
