"""Dense slots (include/ssgpu.h "dense-slot GroupAggregate", csrc/launch.h DenseKeyMap; SURVEY 8(e): slot = dense key index).

A plain GroupAggregate stage whose key columns span small value ranges indexes its tables by the keys' mixed-radix number:
no hashing, no probe.  The HIP path in that shape must give the CPU oracle's rows bit for bit -- the oracle restates the
reference's hash aggregate (cursor/core/aggregate_groups.cc:332-433, row_hash_set.cc:458-517), whose result does not
depend on how a group finds its row.  `Plan.stage_info()` asserts that the dense shape is the one that ran (dense_slots > 0;
group_shape 1 = partitions of slot ranges, 3 = one table fed from the input columns), that ranges are widened when a later
input leaves them, and that inputs the shape does not fit (ranges too wide, keys piled on a few slots) leave it.
The suite's default is group_dense = 0 (tests/conftest.py); every context here sets 1, the library's own default."""
import math
import os

import numpy as np
import pytest

import bench
import supersonic_amd as ss
from fuzz_plans import make_view as fuzz_view
from helpers import assert_cols_equal, run_both, sort_rows, to_cols, ulp_distance
from oracle import oracle
from test_parity_gpu import group_query, make_view

pytestmark = pytest.mark.gpu
NA = ss.NamedAttribute


def dense_ctx(**options):
    ctx = ss.Context(0)
    ctx.set_option("group_dense", 1)
    for k, v in options.items():
        ctx.set_option(k, v)
    return ctx


def run_plan(plan, want, context, runs=1, view=None):
    infos = []
    for i in range(runs):
        plan.run(view) if view is not None else plan.run()
        assert_cols_equal(sort_rows(to_cols(plan.fetch())), sort_rows(want), context="%s (run %d)" % (context, i))
        infos.append([st for st in plan.stage_info() if st["kind"] == 3][-1])
    return infos


def group3_op(view, with_filter):
    saved = bench.GROUP_FILTER
    bench.GROUP_FILTER = with_filter
    try:
        return bench.build_group_plan(ss, view)
    finally:
        bench.GROUP_FILTER = saved


# ---- BASELINE configs #3 / #4 in the dense shape: what `bench.py --query group3 / group` runs by default -----------------------
@pytest.mark.parametrize("with_filter", [False, True])
@pytest.mark.parametrize("specialize", [0, 1])
@pytest.mark.parametrize("split", [1, 0])
def test_config3_and_4_take_dense_partitions_from_the_first_run(specialize, with_filter, split):
    # split = 1 (the default): the records cross HBM as payload words + 16-bit table entries; 0: whole records, index word first
    n = 2_000_000
    view = ss.View(bench.group_schema(ss), bench.host_columns(np, "group", n))
    op = group3_op(view, with_filter)
    _s, want = oracle.run(op)
    plan = ss.Plan(op, dense_ctx(specialize=specialize, part_split=split))
    infos = run_plan(plan, want, "config #%d, dense" % (4 if with_filter else 3), runs=3)
    assert [i["split_records"] for i in infos] == [split] * 3, infos
    # k1 = g // 317 in [0, 315], k2 = g % 317 in [0, 316]: 316 x 317 slots; no scout, no direct first run, no rerun
    assert [i["dense_slots"] for i in infos] == [316 * 317] * 3 and [i["group_shape"] for i in infos] == [1, 1, 1], infos
    assert [i["reruns"] for i in infos] == [0, 0, 0] and infos[0]["plain_scatter"] == 1, infos
    if specialize:
        assert infos[-1]["specialized"] & 12 == 12, plan.specialize_reason()      # partition aggregation + plain scatter, compiled for dense records


@pytest.mark.parametrize("specialize", [0, 1])
def test_few_groups_take_one_dense_table_fed_from_the_columns(specialize):
    n = 2_000_000
    cols = bench.host_columns(np, "group", n, seed=11)
    g = np.random.default_rng(12).integers(0, 1000, n)
    cols[1], cols[2] = (g // 37).astype(np.int32), (g % 37).astype(np.int32)
    for with_filter in (False, True):
        op = group3_op(ss.View(bench.group_schema(ss), cols), with_filter)
        _s, want = oracle.run(op)
        plan = ss.Plan(op, dense_ctx(specialize=specialize))
        infos = run_plan(plan, want, "1000 groups, dense", runs=3)
        assert [i["group_shape"] for i in infos] == [3, 3, 3] and infos[0]["dense_slots"] == 28 * 37, infos
        if specialize:
            assert infos[-1]["specialized"] & 16, plan.specialize_reason()


@pytest.mark.parametrize("ranges", [2, 3, 8])
@pytest.mark.parametrize("nullable", [False, True])
@pytest.mark.parametrize("with_filter", [False, True])
def test_dense_partitions_over_row_ranges(ranges, nullable, with_filter):
    """Large inputs take the dense partitions in row ranges: range k is aggregated (side stream, the table read back from what the launch
    before it dumped) while range k + 1 is scattered.  Forced here on a small input; three runs of one plan (the tables are reused)."""
    n = 300007
    ctx = dense_ctx(dense_min_rows=1, dense_parts=7, group_resident=0, part_overlap=ranges, part_overlap_rows=1)
    keys = ("k1",) if nullable else ("k1", "k2")
    op = group_query(make_view(n, nullable=nullable), with_filter, keys)
    _s, want = oracle.run(op)
    plan = ss.Plan(op, ctx)
    infos = run_plan(plan, want, "dense partitions, %d row ranges" % ranges, runs=3)
    assert [i["row_ranges"] for i in infos] == [ranges] * 3 and infos[-1]["group_shape"] == 1, infos


def test_config3_over_row_ranges_specialized():
    n = 2_000_000
    view = ss.View(bench.group_schema(ss), bench.host_columns(np, "group", n))
    for with_filter in (False, True):
        op = group3_op(view, with_filter)
        _s, want = oracle.run(op)
        plan = ss.Plan(op, dense_ctx(specialize=1, part_overlap=4, part_overlap_rows=1))
        infos = run_plan(plan, want, "config #%d, dense, 4 row ranges" % (4 if with_filter else 3), runs=3)
        assert [i["row_ranges"] for i in infos] == [4] * 3, infos


# ---- every row count / NULL / Filter case of the hashed GroupAggregate tests, in the dense shape (dense_min_rows = 1) --------------
@pytest.mark.parametrize("n", [1, 65, 1025, 100003])
@pytest.mark.parametrize("with_filter", [False, True])
@pytest.mark.parametrize("nullable", [False, True])
@pytest.mark.parametrize("parts", [0, 7, -7])
def test_dense_group_aggregate_small_inputs(n, with_filter, nullable, parts):
    # parts = 7: forces the partitioned dense shape with an odd partition count where the ranges would fit one table (-7: the same with whole records)
    split, parts = (0, -parts) if parts < 0 else (1, parts)
    ctx = dense_ctx(dense_min_rows=1, dense_parts=parts, group_resident=0 if parts else 1, part_split=split)
    keys = ("k1",) if nullable else ("k1", "k2")
    op = group_query(make_view(n, nullable=nullable), with_filter, keys)
    run_both(op, ctx, ignore_order=True)
    plan = ss.Plan(op, ctx)
    plan.run()
    info = plan.stage_info()[-1]
    # (parts = 0: one table when the ranges fit the LDS next to this query's 15 aggregates, partitions when they do not)
    assert info["dense_slots"] > 0 and info["group_shape"] in ((1,) if parts else (1, 3)), info


def keyed_view(keys, n, seed=3, nullable=()):
    """INT32 / INT64 / BOOL key columns from arrays + the value columns of the group tests."""
    rng = np.random.default_rng(seed)
    attrs, cols = [], []
    for name, (dtype, arr) in keys.items():
        z = (rng.random(n) < 0.1) if name in nullable else None
        attrs.append(ss.Attribute(name, dtype, ss.NULLABLE if z is not None else ss.NOT_NULLABLE))
        cols.append(ss.Column(arr, z) if z is not None else arr)
    attrs += [ss.Attribute("v", ss.INT64), ss.Attribute("d", ss.DOUBLE, ss.NULLABLE), ss.Attribute("u", ss.UINT32)]
    cols += [rng.integers(-1000, 1000, n), ss.Column(rng.integers(-4000, 4000, n) * 0.25, rng.random(n) < 0.2), rng.integers(0, 1 << 32, n).astype(np.uint32)]
    return ss.View(ss.TupleSchema(attrs), cols)


def keyed_op(view, names):
    spec = (ss.AggregationSpecification().AddAggregation(ss.SUM, "v", "s").AddAggregation(ss.COUNT, "", "c").AddAggregation(ss.MIN, "v", "mn")
            .AddAggregation(ss.SUM, "d", "sd").AddAggregation(ss.MAX, "d", "mx").AddAggregation(ss.COUNT, "d", "cd").AddAggregation(ss.MAX, "u", "mu"))
    return ss.GroupAggregate(ss.ProjectNamedAttributes(list(names)), spec, None, ss.ScanView(view))


@pytest.mark.parametrize("specialize", [0, 1])
def test_negative_keys_null_keys_and_the_all_ones_key(specialize):
    # two INT32 keys in [-5, 5] (both -1 pack into the tables' EMPTY value: that group lives in the special slot), one of them
    # NULLABLE (its NULL is the last offset of its span), an INT64 key below zero, a BOOL key
    n = 300_000
    rng = np.random.default_rng(5)
    for keys, nullable in (({"a": (ss.INT32, rng.integers(-5, 6, n).astype(np.int32)), "b": (ss.INT32, rng.integers(-5, 6, n).astype(np.int32))}, ()),
                           ({"a": (ss.INT32, rng.integers(-700, -3, n).astype(np.int32)), "t": (ss.BOOL, rng.integers(0, 2, n).astype(bool))}, ("a",)),
                           ({"a": (ss.INT64, rng.integers(-(1 << 40) - 900, -(1 << 40), n)), "t": (ss.BOOL, rng.integers(0, 2, n).astype(bool))}, ("t",)),
                           ({"a": (ss.UINT32, rng.integers((1 << 32) - 300, 1 << 32, n).astype(np.uint32))}, ())):
        view = keyed_view(keys, n, nullable=nullable)
        op = keyed_op(view, keys.keys())
        _s, want = oracle.run(op)
        plan = ss.Plan(op, dense_ctx(specialize=specialize))
        if "t" in nullable:        # INT64 + a NULLABLE BOOL do not pack into one 64-bit key word: sort + clustered aggregation, no table
            plan.run()
            assert_cols_equal(sort_rows(to_cols(plan.fetch())), sort_rows(want), context="wide keys")
            continue
        infos = run_plan(plan, want, "keys %s" % list(keys), runs=2)
        assert infos[-1]["dense_slots"] > 0 and infos[-1]["reruns"] == 0, infos
        if nullable:               # a tenth of the rows carry the NULL key: its slots' partition needs larger segments than the even share (grown once)
            assert infos[0]["reruns"] >= 1 and infos[-1]["part_seg_growth"] > 1, infos


@pytest.mark.parametrize("lazy", [0, 1])
def test_ranges_are_widened_when_a_later_input_leaves_them(lazy):
    # lazy = 1 (the sharded drivers' opt-in): the miss of a steady-state run is seen when the result is touched, and the run repeated
    n = 200_000
    rng = np.random.default_rng(7)
    first = keyed_view({"a": (ss.INT32, rng.integers(0, 100, n).astype(np.int32))}, n, seed=8)
    later = keyed_view({"a": (ss.INT32, rng.integers(-50, 300, n).astype(np.int32))}, n, seed=9)
    op = keyed_op(first, ["a"])
    _s, want1 = oracle.run(op)
    _s, want2 = oracle.run(keyed_op(later, ["a"]))
    plan = ss.Plan(op, dense_ctx(lazy_feedback=lazy))
    i1 = run_plan(plan, want1, "first ranges", runs=3)          # (the third run is past the lazy-feedback threshold: the miss of the next is seen late)
    assert i1[-1]["dense_slots"] == 100, i1
    i2 = run_plan(plan, want2, "wider ranges", runs=2, view=later)
    assert i2[0]["dense_slots"] == 350 and i2[0]["reruns"] >= 1 and i2[1]["reruns"] == 0, i2
    i3 = run_plan(plan, want1, "the first input again: inside the union", runs=1, view=first)
    assert i3[0]["dense_slots"] == 350 and i3[0]["reruns"] == 0, i3


def test_ranges_too_wide_for_a_table_keep_the_hashed_shapes():
    n = 300_000
    rng = np.random.default_rng(11)
    wide = keyed_view({"a": (ss.INT64, rng.integers(-(1 << 62), 1 << 62, n))}, n)
    many = keyed_view({"a": (ss.INT64, np.arange(n, dtype=np.int64) % 100000)}, n)      # 100000 slots for 300000 rows: more table than scan
    for view in (wide, many):
        op = keyed_op(view, ["a"])
        _s, want = oracle.run(op)
        infos = run_plan(ss.Plan(op, dense_ctx()), want, "not dense", runs=3)
        assert [i["dense_slots"] for i in infos] == [0, 0, 0], infos
    big = keyed_view({"a": (ss.INT64, np.arange(1_000_000, dtype=np.int64) % 100000)}, 1_000_000)   # the same keys under 1 M rows: dense
    op = keyed_op(big, ["a"])
    _s, want = oracle.run(op)
    infos = run_plan(ss.Plan(op, dense_ctx()), want, "dense at 1 M rows", runs=2)
    assert infos[0]["dense_slots"] == 100000 and infos[0]["group_shape"] == 1, infos


def test_keys_piled_on_a_few_slots_leave_the_dense_shape():
    # half of the rows carry ONE key pair: its partition's segments run full -- the stage goes back to the hashed shapes, which
    # take heavy hitters apart (hot_keys); same rows as the oracle all the way
    n = 2_000_000
    cols = bench.host_columns(np, "group", n, seed=7)
    hot = np.random.default_rng(8).random(n) < 0.5
    cols[1] = np.where(hot, 123, cols[1]).astype(np.int32)
    cols[2] = np.where(hot, 45, cols[2]).astype(np.int32)
    op = group3_op(ss.View(bench.group_schema(ss), cols), False)
    _s, want = oracle.run(op)
    infos = run_plan(ss.Plan(op, dense_ctx(group_partition=2)), want, "skewed keys", runs=3)
    assert infos[0]["reruns"] >= 1 and infos[-1]["dense_slots"] == 0 and infos[-1]["hot_keys"] >= 1 and infos[-1]["reruns"] == 0, infos


def test_string_keys_are_dense_dictionary_codes():
    n = 200_000
    rng = np.random.default_rng(13)
    words = np.array(["w%03d" % i for i in range(300)], dtype=object)
    schema = ss.TupleSchema([ss.Attribute("s", ss.STRING), ss.Attribute("t", ss.BOOL), ss.Attribute("v", ss.INT64), ss.Attribute("d", ss.DOUBLE)])
    view = ss.View(schema, [words[rng.integers(0, 300, n)], rng.integers(0, 2, n).astype(bool), rng.integers(-1000, 1000, n), rng.integers(-4000, 4000, n) * 0.25])
    spec = ss.AggregationSpecification().AddAggregation(ss.SUM, "v", "s_v").AddAggregation(ss.MIN, "d", "mn").AddAggregation(ss.COUNT, "", "n")
    op = ss.GroupAggregate(ss.ProjectNamedAttributes(["s", "t"]), spec, None, ss.ScanView(view))
    _s, want = oracle.run(op)
    infos = run_plan(ss.Plan(op, dense_ctx()), want, "STRING + BOOL keys", runs=2)
    assert infos[-1]["dense_slots"] == 600, infos


@pytest.mark.parametrize("groups", [1000, 40000])
def test_dense_double_sums_stay_within_one_ulp_of_the_exact_sum(groups):
    # the adversarial DOUBLE set (uniform (-1, 1), full mantissas, a few 1e9 terms): compensated LDS atomics, (hi, lo) dumps
    n = 2_000_003
    rng = np.random.default_rng(5)
    x = rng.uniform(-1.0, 1.0, n)
    x[rng.integers(0, n, n // 50)] *= 1e9
    g = rng.integers(0, groups, n).astype(np.int32)
    view = ss.View(ss.TupleSchema([ss.Attribute("g", ss.INT32), ss.Attribute("x", ss.DOUBLE)]), [g, x])
    op = ss.GroupAggregate(ss.ProjectNamedAttributes(["g"]), ss.AggregationSpecification().AddAggregation(ss.SUM, "x", "s"), None, ss.ScanView(view))
    plan = ss.Plan(op, dense_ctx())
    got = []
    for _ in range(2):
        plan.run()
        r = plan.fetch()
        order = np.argsort(r.column(0).data)
        got.append(r.column(1).data[order])
    assert plan.stage_info()[-1]["dense_slots"] == groups, plan.stage_info()
    order = np.argsort(g, kind="stable")
    bounds = np.searchsorted(g[order], np.arange(groups + 1))
    exact = np.array([math.fsum(x[order[bounds[i]:bounds[i + 1]]].tolist()) for i in range(groups)])
    d = ulp_distance(got[0], exact)
    assert d.max() <= 1.0, (d.max(), int((d > 0).sum()))
    assert got[0].tobytes() == got[1].tobytes()


# ---- random PLAIN GroupAggregates (keys and aggregate inputs are input columns, Filters are `column CMP constant`) over the fuzz
# ---- generator's table, every one in the dense shape: key subsets with NULLABLE, negative, BOOL, DATE and STRING members ----------
from fuzz_plans import random_plain_group   # noqa: E402  (shared with tests/fuzz_worker.py)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SS_DENSE_FUZZ_SEEDS", "400"))))
def test_random_plain_group_aggregates_in_the_dense_shape(seed):
    ctx = dense_ctx(dense_min_rows=1, dense_parts=(0, 5, 64)[seed % 3], group_resident=seed % 2, specialize=1 if seed % 50 == 0 else 0)
    view = fuzz_view(1537 if seed % 4 else 70001, 5000 + seed)
    op = random_plain_group(seed, view)
    run_both(op, ctx, ignore_order=True)
    plan = ss.Plan(op, ctx)
    plan.run()
    stages = [st for st in plan.stage_info() if st["kind"] == 3]
    if stages:      # (key sets beyond 64 packed bits -- a NULLABLE INT32 next to another INT32 -- run as sort + clustered aggregation: no table at all)
        assert stages[-1]["dense_slots"] > 0, (seed, stages[-1])
