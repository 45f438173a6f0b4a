"""Binder fuzz (CPU): for seeded random plans the device binder (ssgpu_plan_create on a bind-only
context) must agree with the oracle's binder on success / failure, the return code of a failure, and the
result schema (names, types, nullability) -- the reference's Bind() contract, without touching a GPU."""
import os

import pytest

import supersonic_amd as ss
from oracle import oracle
from fuzz_plans import Gen, make_view


@pytest.mark.parametrize("seed", range(int(os.environ.get("SS_FUZZ_SEEDS", "2000"))))
def test_binder_agrees_with_oracle_on_random_plans(seed):
    view = make_view(3, seed)
    op, _ordered = Gen(seed).plan(view)
    want_schema = want_err = None
    try:
        want_schema, _cols = oracle.run(op)
    except oracle.OracleError as e:
        want_err = e.return_code
    try:
        plan = ss.Plan(op, ss.Context(-1))
    except ss.SupersonicException as e:
        assert want_err is not None, "device binder failed (%s) where the oracle binds" % e
        if e.return_code != ss.ERROR_NOT_IMPLEMENTED:
            assert e.return_code == want_err
        return
    assert want_err is None, "device binder accepted a plan the oracle rejects with %s" % want_err
    rs = plan.result_schema
    got = [(rs.attribute(i).name(), rs.attribute(i).type(), rs.attribute(i).is_nullable()) for i in range(rs.attribute_count())]
    assert got == [tuple(x) for x in want_schema]


@pytest.mark.parametrize("seed", range(6000, 6200))
def test_binder_agrees_with_oracle_on_distinct_aggregates_under_a_key_limit(seed):
    view = make_view(3, seed)
    op, _ordered = Gen(seed).distinct_limit_plan(view)
    want_schema = want_err = None
    try:
        want_schema, _cols = oracle.run(op)
    except oracle.OracleError as e:
        want_err = e.return_code
    try:
        plan = ss.Plan(op, ss.Context(-1))
    except ss.SupersonicException as e:
        assert want_err is not None, "device binder failed (%s) where the oracle binds" % e
        assert e.return_code == want_err
        return
    assert want_err is None, "device binder accepted a plan the oracle rejects with %s" % want_err
    rs = plan.result_schema
    got = [(rs.attribute(i).name(), rs.attribute(i).type(), rs.attribute(i).is_nullable()) for i in range(rs.attribute_count())]
    assert got == [tuple(x) for x in want_schema]


@pytest.mark.parametrize("seed", range(4000, 4250))
def test_binder_agrees_with_oracle_on_random_ordered_aggregates(seed):
    # DISTINCT next to FIRST / LAST, key limits, DISTINCT inside AggregateClusters: the binder refuses nothing the oracle binds
    view = make_view(3, seed)
    op, _ordered = Gen(seed).ordered_aggregate_plan(view)
    want_schema = want_err = None
    try:
        want_schema, _cols = oracle.run(op)
    except oracle.OracleError as e:
        want_err = e.return_code
    try:
        plan = ss.Plan(op, ss.Context(-1))
    except ss.SupersonicException as e:
        assert want_err is not None, "device binder failed (%s) where the oracle binds" % e
        assert e.return_code == want_err
        return
    assert want_err is None, "device binder accepted a plan the oracle rejects with %s" % want_err
    rs = plan.result_schema
    got = [(rs.attribute(i).name(), rs.attribute(i).type(), rs.attribute(i).is_nullable()) for i in range(rs.attribute_count())]
    assert got == [tuple(x) for x in want_schema]
