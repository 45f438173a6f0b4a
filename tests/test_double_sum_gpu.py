"""DOUBLE SUM on data whose partial sums are NOT exact (SURVEY 8(d)'s adversarial set: uniform (-1, 1), full
mantissa).  The reference folds sequentially in input order (aggregation_operators.h:173-186), which no parallel
device can reproduce bit for bit; the parity bar of north_star is "within 1 ULP for DOUBLE aggregates".  Here both
the oracle's sequential fold and the HIP path are measured against the exactly rounded sum (math.fsum):
the HIP path must be within 1 ULP of it -- scalar SUM (per-lane double-double accumulators, fixed combine tree)
and grouped SUM in every execution shape (compensated atomics: each add's rounding error is captured exactly) --
and must give the same bits run after run.  The distances are written to gpurun_out/double_sum_ulp.json for DESIGN.md."""
import json
import math
import os

import numpy as np
import pytest

import supersonic_amd as ss
from oracle import oracle
from helpers import ulp_distance

pytestmark = pytest.mark.gpu
NA = ss.NamedAttribute
REPORT = {}


def adversarial(n, groups, seed=5):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1.0, 1.0, n)
    x[rng.integers(0, n, n // 50)] *= 1e9            # a few large terms: cancellation + very different magnitudes
    x[::7] = -x[1::7][: len(x[::7])] if len(x[1::7]) >= len(x[::7]) else x[::7]
    g = rng.integers(0, groups, n).astype(np.int32)
    schema = ss.TupleSchema([ss.Attribute("g", ss.INT32), ss.Attribute("x", ss.DOUBLE)])
    return ss.View(schema, [g, x]), g, x


def save_report():
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "double_sum_ulp.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


@pytest.mark.parametrize("n", [1000003, 10000019])
def test_scalar_double_sum_is_within_one_ulp_of_the_exact_sum(n):
    view, _g, x = adversarial(n, 8)
    op = ss.ScalarAggregate(ss.AggregationSpecification().AddAggregation(ss.SUM, "x", "s"), ss.ScanView(view))
    exact = math.fsum(x.tolist())
    ctx = ss.Context(0)
    got = [ss.drain(op.CreateCursor(ctx)).column(0).data[0] for _ in range(3)]
    _schema, want = oracle.run(op)
    d_hip = float(ulp_distance(np.array([got[0]]), np.array([exact]))[0])
    d_seq = float(ulp_distance(np.array([want[0][0][0]]), np.array([exact]))[0])
    REPORT["scalar_n%d" % n] = {"hip_ulp_vs_exact": d_hip, "sequential_fold_ulp_vs_exact": d_seq}
    save_report()
    assert d_hip <= 1.0, (got[0], exact, d_hip)
    assert got[0].tobytes() == got[1].tobytes() == got[2].tobytes()       # fixed combine tree: reproducible


@pytest.mark.parametrize("groups,partition", [(7, 0), (1000, 0), (40000, 0), (40000, 2)])
def test_grouped_double_sum_is_within_one_ulp_of_the_exact_sum(groups, partition):
    n = 2000003
    view, g, x = adversarial(n, groups)
    spec = ss.AggregationSpecification().AddAggregation(ss.SUM, "x", "s").AddAggregation(ss.COUNT, "", "n")
    op = ss.GroupAggregate(ss.ProjectNamedAttributes(["g"]), spec, None, ss.ScanView(view))
    order = np.argsort(g, kind="stable")
    gs, xs = g[order], x[order]
    bounds = np.flatnonzero(np.r_[True, gs[1:] != gs[:-1], True])
    exact = {int(gs[bounds[i]]): math.fsum(xs[bounds[i]:bounds[i + 1]].tolist()) for i in range(len(bounds) - 1)}
    ctx = ss.Context(0)
    ctx.set_option("group_partition", partition)
    runs = []
    plan = ss.Plan(op, ctx)
    for _ in range(3):                       # also walks the fed-back execution shapes of one plan
        plan.run()
        v = plan.fetch()
        o = np.argsort(v.column(0).data)
        runs.append((v.column(0).data[o], v.column(1).data[o]))
    keys, sums = runs[0]
    want = np.array([exact[int(k)] for k in keys])
    d = ulp_distance(sums, want)
    _schema, ocols = oracle.run(op)
    oo = np.argsort(ocols[0][0])
    d_seq = ulp_distance(ocols[1][0][oo], np.array([exact[int(k)] for k in ocols[0][0][oo]]))
    REPORT["grouped_%dgroups_partition%d" % (groups, partition)] = {
        "hip_max_ulp_vs_exact": float(d.max()), "hip_groups_off_by_one_ulp": int((d > 0).sum()),
        "sequential_fold_max_ulp_vs_exact": float(d_seq.max()), "sequential_fold_groups_not_exact": int((d_seq > 0).sum()), "groups": int(len(keys))}
    save_report()
    assert d.max() <= 1.0, (float(d.max()), int((d > 1).sum()))
    for k2, s2 in runs[1:]:                  # same bits whatever order the atomics arrived in
        assert np.array_equal(k2, keys) and s2.tobytes() == sums.tobytes()
