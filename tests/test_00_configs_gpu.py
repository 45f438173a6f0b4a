"""HIP path vs the CPU oracle at the SHAPES of BASELINE.json's configs, at sizes the oracle does in about a second.

Collected first (file name), so that a driver run reaches every configuration's oracle comparison in its first minute:

  #1 / #2  Compute(a+b) -> Filter(a>K) -> ScalarAggregate -- `bench.build_plan`, the headline plan, on the 8-column block
  #3       GroupAggregate(k1, k2; SUM / MIN / MAX x d0..d3), random keys, 1e5 groups -- `bench.build_group_plan` exactly as
           `bench.py --query group3` builds it, through every execution shape (direct, hash partitions with both specialised
           kernels, slab), plus skewed keys that overflow segments (part_seg_growth) and partitions (part_n doubling)
  #4       the row-range-sharded Filter -> GroupAggregate: `DeviceShardedGroupAggregate` (what `bench.py --query group
           --force-distributed` steps) on a 1-rank RCCL group, pack / all-gather / unpack / merge, vs the oracle
  #5       Sort(d ASC) over the 8-column block -- `bench.build_sort_plan`: one-word hybrid passes + tie fix-up + record
           gather, and adversarial keys whose high halves collide (fallback to all eight digits)

Reference shapes: supersonic/benchmark/examples/operation_example.cc:42-171.  Every comparison is bit-exact (the DOUBLE
columns hold values whose partial sums are exact); `Plan.stage_info()` asserts that the shape a case names
is the shape that ran."""
import os
import socket

import numpy as np
import pytest

import bench
import supersonic_amd as ss
from helpers import assert_cols_equal, schema_list, sort_rows, to_cols
from oracle import oracle

pytestmark = pytest.mark.gpu
N_ROWS = 2_000_000


def make_ctx(**options):
    ctx = ss.Context(0)
    for k, v in options.items():
        ctx.set_option(k, v)
    return ctx


def check_plan(plan, want, context, ignore_order=False, runs=1, view=None):
    infos = []
    for i in range(runs):
        plan.run(view) if view is not None else plan.run()
        got = to_cols(plan.fetch())
        w = want
        if ignore_order:
            got, w = sort_rows(got), sort_rows(want)
        assert_cols_equal(got, w, context="%s (run %d)" % (context, i))
        infos.append(plan.stage_info())
    return infos


# ---- configs #1 / #2: the headline plan ----------------------------------------------------------------------------
@pytest.mark.parametrize("specialize", [0, 1])
def test_config2_filter_project_aggregate_8_columns(specialize):
    view = ss.View(bench.bench_schema(ss), bench.host_columns(np, "wide", N_ROWS))
    op = bench.build_plan(ss, view)
    oschema, want = oracle.run(op)
    plan = ss.Plan(op, make_ctx(specialize=specialize))
    assert schema_list(plan.result_schema) == oschema
    check_plan(plan, want, "config #2, specialize=%d" % specialize, runs=2)
    assert plan.specialized() == specialize, plan.specialize_reason()


def test_config1_cpu_reference_shape():
    # configs[0]: Compute(a+b) -> Filter(a>K) -> Sum/Count on a 1 M-row x 4 INT64 table
    rng = np.random.default_rng(42)
    m = 1_000_000
    s4 = ss.TupleSchema([ss.Attribute(x, ss.INT64) for x in ("a", "b", "c", "d")])
    v4 = ss.View(s4, [rng.integers(0, 1000, m), rng.integers(0, 1000, m), np.arange(m, dtype=np.int64) % 100000, rng.integers(-(1 << 62), 1 << 62, m)])
    NA = ss.NamedAttribute
    op = ss.ScalarAggregate(ss.AggregationSpecification().AddAggregation(ss.SUM, "s", "sum_s").AddAggregation(ss.COUNT, "a", "cnt"),
                            ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(bench.K_FILTER)), ss.ProjectAllAttributes(),
                                      ss.Compute(ss.CompoundExpression().Add(NA("a")).AddAs("s", ss.Plus(NA("a"), NA("b"))), ss.ScanView(v4))))
    _s, want = oracle.run(op)
    check_plan(ss.Plan(op, make_ctx()), want, "config #1")


# ---- config #3: GroupAggregate, 2 x INT32 keys, 1e5 groups, 12 DOUBLE aggregates ----------------------------------------
def group3_op(view):
    saved = bench.GROUP_FILTER
    bench.GROUP_FILTER = False                        # what `--query group3` does
    try:
        return bench.build_group_plan(ss, view)
    finally:
        bench.GROUP_FILTER = saved


@pytest.fixture(scope="module")
def group3():
    view = ss.View(bench.group_schema(ss), bench.host_columns(np, "group", N_ROWS))
    op = group3_op(view)
    oschema, want = oracle.run(op)
    assert len(want[0][0]) == bench.N_GROUPS          # 2 M uniform draws hit every one of the 1e5 (k1, k2) pairs
    return view, op, oschema, want


def test_config3_group_aggregate_by_run_feedback(group3):
    # without a scout run (inputs below group_scout_rows = 8 M rows get none; group_scout = 0 says so explicitly): the
    # first run takes the direct shape, its feedback (most rows bypass the LDS table) moves the plan to the hash partitions;
    # every run of the walk gives the oracle's rows
    _view, op, oschema, want = group3
    plan = ss.Plan(op, make_ctx(group_scout=0))
    assert schema_list(plan.result_schema) == oschema
    infos = check_plan(plan, want, "config #3 adaptive", ignore_order=True, runs=4)
    shapes = [i[0]["group_shape"] for i in infos]
    assert shapes[0] == 0 and shapes[-1] == 1, shapes


def test_config3_first_run_after_a_scout_is_partitioned(group3):
    # group_scout_rows = 1 M (default 8 M): a scout run over a prefix (an eighth of this 2 M-row input) estimates the group count,
    # and already the FIRST run -- the only one of a cursor that is drained once -- takes the hash partitions
    _view, op, _s, want = group3
    plan = ss.Plan(op, make_ctx(group_scout_rows=1 << 20))
    infos = check_plan(plan, want, "config #3 after a scout", ignore_order=True, runs=3)
    assert [i[0]["group_shape"] for i in infos] == [1, 1, 1], infos


@pytest.mark.parametrize("part_plain", [1, 0])
def test_config3_partitioned_with_specialised_kernels(group3, part_plain):
    # part_plain=1 (default): the scatter is ssgpu_part_scatter_plain_kernel over (partition, XCD) segments -- config #3 is a
    # "plain" stage (keys and aggregate inputs are input columns); part_plain=0: the scatter as a tile-VM program
    _view, op, _s, want = group3
    plan = ss.Plan(op, make_ctx(group_partition=2, specialize=1, part_plain=part_plain))
    infos = check_plan(plan, want, "config #3 partitioned + specialised, part_plain=%d" % part_plain, ignore_order=True, runs=2)
    assert infos[-1][0]["group_shape"] == 1 and infos[-1][0]["reruns"] == 0, infos
    assert infos[-1][0]["plain_scatter"] == part_plain, infos
    want_kernels = 12 if part_plain else 6       # partition aggregation + the plain scatter kernel / the partition-scatter program
    assert infos[-1][0]["specialized"] & want_kernels == want_kernels, plan.specialize_reason()


@pytest.mark.parametrize("part_plain", [1, 0])
def test_config3_partitioned_interpreted_and_few_partitions_double(group3, part_plain):
    # 64 partitions x ~1000-entry tables cannot hold 1e5 groups: "partition finer and rerun" until they fit
    _view, op, _s, want = group3
    plan = ss.Plan(op, make_ctx(group_partition=2, part_n=64, part_plain=part_plain))
    infos = check_plan(plan, want, "config #3, part_n doubling, part_plain=%d" % part_plain, ignore_order=True, runs=2)
    assert infos[0][0]["reruns"] >= 1 and infos[0][0]["part_n"] > 64, infos
    assert infos[1][0]["reruns"] == 0, infos                                    # the second run starts from what the first learnt


@pytest.mark.parametrize("specialize,part_plain", [(0, 1), (1, 1), (0, 0)])
def test_config3_skewed_keys_overflow_their_segments(specialize, part_plain):
    # half of the rows carry ONE (k1, k2) pair: the (partition, workgroup) segments of that pair's partition run full, the
    # stage reruns with 4x larger segments (part_seg_growth) -- same rows as the oracle, both kernel forms
    cols = bench.host_columns(np, "group", N_ROWS, seed=7)
    hot = np.random.default_rng(8).random(N_ROWS) < 0.5
    cols[1] = np.where(hot, 123, cols[1]).astype(np.int32)
    cols[2] = np.where(hot, 45, cols[2]).astype(np.int32)
    op = group3_op(ss.View(bench.group_schema(ss), cols))
    _s, want = oracle.run(op)
    plan = ss.Plan(op, make_ctx(group_partition=2, specialize=specialize, part_plain=part_plain))
    infos = check_plan(plan, want, "config #3 skewed", ignore_order=True, runs=2)
    if part_plain:
        # the plain scatter looks at a sample of the rows when a segment overflows: the one hot pair is aggregated apart from the
        # partitions (hot_only resident pass), the segments stay as sized and the stage stays in the partitioned shape
        assert infos[0][0]["hot_keys"] >= 1 and infos[-1][0]["group_shape"] == 1 and infos[-1][0]["part_seg_growth"] == 1, infos
        assert infos[-1][0]["reruns"] == 0, infos                                         # the second run knows the hot key already
    else:
        assert infos[0][0]["part_seg_growth"] > 1 or infos[0][0]["group_shape"] == 0, infos   # grew its segments (or, beyond x64, fell back to the direct shape)


def test_config3_many_heavy_hitters_next_to_uniform_keys():
    # 20 keys hold 60 % of the rows between them (3 % each) and the EMPTY-sentinel-free rest is uniform: all of them are found
    # in the sample and aggregated apart; a NULLABLE value column and a Filter ride along (the hot pass applies both)
    cols = bench.host_columns(np, "group", N_ROWS, seed=17)
    rng = np.random.default_rng(18)
    u = rng.random(N_ROWS)
    hot_g = rng.integers(0, 100000, 20)
    g = np.where(u < 0.6, hot_g[(u * 1e6).astype(np.int64) % 20], cols[1].astype(np.int64) * 317 + cols[2])
    cols[1], cols[2] = (g // 317).astype(np.int32), (g % 317).astype(np.int32)
    view = ss.View(bench.group_schema(ss), cols)
    for op in (group3_op(view), ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), bench.group_spec(ss), None, bench.group_child(ss, view))):
        _s, want = oracle.run(op)
        plan = ss.Plan(op, make_ctx(group_partition=2, specialize=1))
        infos = check_plan(plan, want, "config #3, 20 heavy hitters", ignore_order=True, runs=3)
        assert infos[0][0]["hot_keys"] >= 15 and infos[-1][0]["group_shape"] == 1 and infos[-1][0]["part_seg_growth"] == 1, infos


@pytest.mark.parametrize("specialize", [0, 1])
def test_config3_hot_keys_of_an_earlier_run_leave_no_phantom_group(specialize):
    # a plan keeps the heavy-hitter keys it found; run again over data WITHOUT those keys (a stepping shard, a reused plan)
    # a seeded entry no row reached must not come back as a group of zero rows
    cols = bench.host_columns(np, "group", N_ROWS, seed=31)
    hot = np.random.default_rng(32).random(N_ROWS) < 0.5
    cols[1] = np.where(hot, 123, cols[1]).astype(np.int32)
    cols[2] = np.where(hot, 45, cols[2]).astype(np.int32)
    view = ss.View(bench.group_schema(ss), cols)
    op = group3_op(view)
    _s, want = oracle.run(op)
    plan = ss.Plan(op, make_ctx(group_partition=2, specialize=specialize))
    infos = check_plan(plan, want, "skewed, first data", ignore_order=True, runs=2)
    assert infos[-1][0]["hot_keys"] >= 1, infos
    cold = bench.host_columns(np, "group", N_ROWS, seed=33)
    gone = (cold[1] == 123) & (cold[2] == 45)
    cold[1] = np.where(gone, 7, cold[1]).astype(np.int32)            # the hot pair does not occur at all
    view2 = ss.View(bench.group_schema(ss), cold)
    _s, want2 = oracle.run(group3_op(view2))
    infos = check_plan(plan, want2, "same plan, data without its hot key", ignore_order=True, runs=2, view=view2)
    assert infos[-1][0]["hot_keys"] >= 1, infos                         # the hot pass still ran (and published nothing)


@pytest.mark.parametrize("specialize", [0, 1])
def test_heavy_hitters_next_to_the_key_whose_packed_value_is_all_ones(specialize):
    # two NOT NULL INT32 keys that are both -1 pack into the EMPTY value of the tables (it has a reserved slot); with heavy
    # hitters active the scatter's hot-set probe must not take a free slot of its table for that key
    cols = bench.host_columns(np, "group", N_ROWS, seed=41)
    u = np.random.default_rng(42).random(N_ROWS)
    cols[1] = np.where(u < 0.5, 123, np.where(u < 0.502, -1, cols[1])).astype(np.int32)
    cols[2] = np.where(u < 0.5, 45, np.where(u < 0.502, -1, cols[2])).astype(np.int32)
    view = ss.View(bench.group_schema(ss), cols)
    op = group3_op(view)
    _s, want = oracle.run(op)
    plan = ss.Plan(op, make_ctx(group_partition=2, specialize=specialize))
    infos = check_plan(plan, want, "skewed keys + the all-ones key", ignore_order=True, runs=3)
    assert infos[-1][0]["hot_keys"] >= 1 and infos[-1][0]["group_shape"] == 1, infos


def test_config3_shape_with_few_groups_takes_the_slab_form():
    # same 12 aggregates over ~1000 groups: ONE whole-LDS table per aggregation workgroup, no hash partitions
    cols = bench.host_columns(np, "group", N_ROWS, seed=11)
    g = np.random.default_rng(12).integers(0, 1000, N_ROWS)
    cols[1], cols[2] = (g // 37).astype(np.int32), (g % 37).astype(np.int32)
    op = group3_op(ss.View(bench.group_schema(ss), cols))
    _s, want = oracle.run(op)
    plan = ss.Plan(op, make_ctx(specialize=1))
    infos = check_plan(plan, want, "config #3 shape, 1000 groups", ignore_order=True, runs=4)
    assert infos[-1][0]["group_shape"] in (0, 3), infos                    # a plain stage: no scatter, the input columns are read in place
    plan = ss.Plan(op, make_ctx(specialize=1, group_resident=0))
    infos = check_plan(plan, want, "config #3 shape, 1000 groups, records through memory", ignore_order=True, runs=4)
    assert infos[-1][0]["group_shape"] in (0, 2), infos


@pytest.mark.parametrize("part_plain", [1, 0])
def test_config4_per_gpu_query_filter_then_group_aggregate(part_plain):
    # the per-GPU query of config #4 (`bench.py --query group`): Filter(a > 499) below the GroupAggregate; the plain scatter
    # evaluates the predicate itself
    view = ss.View(bench.group_schema(ss), bench.host_columns(np, "group", N_ROWS, seed=21))
    op = ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), bench.group_spec(ss), None, bench.group_child(ss, view))
    _s, want = oracle.run(op)
    plan = ss.Plan(op, make_ctx(group_partition=2, part_plain=part_plain))
    infos = check_plan(plan, want, "config #4 per-GPU query, part_plain=%d" % part_plain, ignore_order=True, runs=2)
    assert infos[-1][0]["group_shape"] == 1 and infos[-1][0]["plain_scatter"] == part_plain, infos
    assert infos[-1][0]["part_seg_growth"] == 1, infos      # half of the rows are dropped: no segment comes near its capacity


# ---- config #4: row-range-sharded Filter -> GroupAggregate, RCCL exchange of the partial tables ---------------------------
def test_config4_sharded_filter_group_aggregate_one_rank():
    import torch
    import torch.distributed as dist
    from supersonic_amd.distributed import DeviceShardedGroupAggregate
    cols = bench.host_columns(np, "group", N_ROWS, seed=5)
    view = ss.View(bench.group_schema(ss), cols)
    want_op = ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), bench.group_spec(ss), None, bench.group_child(ss, view))
    oschema, want = oracle.run(want_op)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        ctx = make_ctx(specialize=1)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        job = DeviceShardedGroupAggregate(ctx, ["k1", "k2"], bench.group_spec(ss), bench.group_child(ss, view))
        for _ in range(3):                       # bench.py's loop: step until the agreed image capacity holds every table
            job.step()
            while not job.check():
                job.step()
        assert job.collectives == 1              # ONE all-gather of the packed partial tables per step
        plan = job.result()[0]
        assert schema_list(plan.result_schema) == oschema
        assert_cols_equal(sort_rows(to_cols(plan.fetch())), sort_rows(want), context="config #4, one rank")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("specialize", [0, 1])
def test_config4_dense_slot_exchange_on_one_rank(specialize):
    # config #4 through DENSE SLOTS (what `bench.py --query group --gpus N` steps by default): the ranks agree on
    # the key ranges, a step is shard scan into a slot-indexed table -> ONE all_to_all_single of slot slices -> element-wise
    # fold + extraction on the plan itself -- no merge plan.  One rank, RCCL to itself; vs the oracle on the whole input
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    from supersonic_amd.distributed import DenseShardedGroupAggregate, PlanDenseBackend
    cols = bench.host_columns(np, "group", N_ROWS, seed=5)
    view = ss.View(bench.group_schema(ss), cols)
    op = ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), bench.group_spec(ss), None, bench.group_child(ss, view))
    oschema, want = oracle.run(op)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        ctx = make_ctx(specialize=specialize)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        backend = PlanDenseBackend(ctx, op)
        job = DenseShardedGroupAggregate(backend)
        for _ in range(3):
            job.step(view)
            while not job.check():
                job.step(view)
        assert job.collectives == 1 and job.setup_collectives == 1, (job.collectives, job.setup_collectives)
        assert job.layout["slots"] == 316 * 317 and job.layout["n_parts"] % job.world == 0, job.layout
        assert schema_list(backend.plan.result_schema) == oschema
        assert_cols_equal(sort_rows(to_cols(backend.plan.fetch())), sort_rows(want), context="config #4, dense exchange, one rank")
        info = backend.plan.stage_info()[0]
        assert info["dense_slots"] == 316 * 317 and info["group_shape"] == 1, info
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_dense_tables_of_several_shards_fold_into_disjoint_owners(world):
    # `world` ranks simulated on one GPU, no process group: every rank's plan runs its shard into a chunked table (chunk r = the
    # slots rank r owns), the all-to-all is done by hand (chunk r of every table -> owner r), every owner folds its `world` images
    # and extracts.  Owners are disjoint and together they are the oracle's result over the whole input; DOUBLE sums cross as raw
    # (hi, lo) accumulators; a NULLABLE key, a NULLABLE aggregate input (contribution counts travel) and a Filter ride along
    torch = pytest.importorskip("torch")
    n = 600_000
    rng = np.random.default_rng(17 + world)
    schema = ss.TupleSchema([ss.Attribute("k", ss.INT32, ss.NULLABLE), ss.Attribute("t", ss.BOOL), ss.Attribute("a", ss.INT64),
                             ss.Attribute("v", ss.INT64), ss.Attribute("d", ss.DOUBLE, ss.NULLABLE), ss.Attribute("x", ss.DOUBLE)])
    data = [ss.Column(rng.integers(-700, 9000, n).astype(np.int32), rng.random(n) < 0.05), rng.integers(0, 2, n).astype(bool), rng.integers(0, 1000, n),
            rng.integers(-1000, 1000, n), ss.Column(rng.integers(-4000, 4000, n) * 0.25, rng.random(n) < 0.2), rng.uniform(-1.0, 1.0, n)]
    whole = ss.View(schema, data)
    spec = (ss.AggregationSpecification().AddAggregation(ss.SUM, "v", "s").AddAggregation(ss.COUNT, "", "c").AddAggregation(ss.MIN, "v", "mn")
            .AddAggregation(ss.SUM, "d", "sd").AddAggregation(ss.MAX, "d", "mx").AddAggregation(ss.COUNT, "d", "cd"))

    def op_of(view):
        return ss.GroupAggregate(ss.ProjectNamedAttributes(["k", "t"]), spec, None,
                                 ss.Filter(ss.Greater(ss.NamedAttribute("a"), ss.ConstInt64(299)), ss.ProjectAllAttributes(), ss.ScanView(view)))
    _s, want = oracle.run(op_of(whole))
    bounds = [n * r // world for r in range(world + 1)]
    bounds[1] = bounds[1] // 3                       # ragged shards (and with world = 8 a small one)
    shards = [ss.View(schema, [ss.Column(c.data[bounds[r]:bounds[r + 1]], None if c.is_null is None else c.is_null[bounds[r]:bounds[r + 1]])
                               for c in (whole.column(i) for i in range(6))]) for r in range(world)]
    ctx = make_ctx(group_dense=1)
    plans = [ss.Plan(op_of(sh), ctx) for sh in shards]
    ranges = [p.key_ranges(sh) for p, sh in zip(plans, shards)]
    union = [(min(r[k][0] for r in ranges), max(r[k][1] for r in ranges)) for k in range(2)]
    layouts = [p.set_dense(union, world) for p in plans]
    assert all(lay == layouts[0] for lay in layouts) and layouts[0]["n_parts"] % world == 0, layouts
    cb = layouts[0]["chunk_bytes"]
    tables = [torch.zeros(world * cb, dtype=torch.uint8, device="cuda") for _ in range(world)]
    for attempt in range(4):
        for p, sh, t in zip(plans, shards, tables):
            p.run_dense(sh, t.data_ptr())
        ctx.synchronize()
        got_rows, flags = [], []
        for owner in range(world):
            images = torch.cat([t[owner * cb:(owner + 1) * cb] for t in tables])     # what all_to_all_single delivers to `owner`
            torch.cuda.synchronize()
            plans[owner].fold_dense(images.data_ptr(), world)
            flags.append(plans[owner].dense_flags())
            got_rows.append(to_cols(plans[owner].fetch()))
        assert all(f == flags[0] for f in flags), flags          # every owner saw every rank's header: the same verdict everywhere
        if flags[0] == (0, 0):
            break
        # a twentieth of the rows carry the NULL key (two slots): their partition outgrows segments sized for an even spread on
        # the larger shards -- every rank enlarges its segments and the step is repeated (DenseShardedGroupAggregate.check)
        assert flags[0] == (2, 0) and attempt < 2, (attempt, flags)
        for p in plans:
            p.dense_grow()
    merged = [(np.concatenate([g[i][0] for g in got_rows]), None if got_rows[0][i][1] is None else np.concatenate([g[i][1] for g in got_rows])) for i in range(len(want))]
    # every group on exactly one owner: the concatenation has the oracle's row count and its rows
    exact = [i for i in range(len(want)) if i != 5]                                        # (column 5 = SUM(d): multiples of 0.25, exact too)
    assert_cols_equal(sort_rows(merged), sort_rows(want), context="dense tables of %d shards" % world)
    assert len(exact) == len(want) - 1


@pytest.mark.parametrize("exchange", ["dense", "key_range", "all_gather"])
def test_config4_a_failing_shard_run_still_joins_the_collective_and_fails_the_step(exchange):
    # a rank whose shard run fails BEFORE the step's collective (here: an interrupt; a memory quota is the same path) must not
    # return while the others wait in it: it sends flagged, empty chunks / images, and every rank's check() raises the code.
    # One rank, RCCL to itself: the step completes (no hang, the collective ran) and check() raises INTERRUPTED; the next step works.
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    from supersonic_amd.distributed import DenseShardedGroupAggregate, DeviceShardedGroupAggregate, PlanDenseBackend
    cols = bench.host_columns(np, "group", 300_000, seed=5)
    view = ss.View(bench.group_schema(ss), cols)
    op = ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), bench.group_spec(ss), None, bench.group_child(ss, view))
    _s, want = oracle.run(op)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        ctx = make_ctx()
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        if exchange == "dense":
            backend = PlanDenseBackend(ctx, op)
            job = DenseShardedGroupAggregate(backend)
            plan_of = lambda: backend.plan                                 # noqa: E731
        else:
            job = DeviceShardedGroupAggregate(ctx, ["k1", "k2"], bench.group_spec(ss), bench.group_child(ss, view), exchange=exchange)
            plan_of = lambda: job.result()[0]                              # noqa: E731

        def good_step():
            job.step(view)
            while not job.check():
                job.step(view)
            assert_cols_equal(sort_rows(to_cols(plan_of().fetch())), sort_rows(want), context="%s exchange" % exchange)
        good_step()
        (backend.plan if exchange == "dense" else job.first).interrupt()    # the next run of the shard plan returns INTERRUPTED
        job.step(view)                                                      # ... and the step still reaches (and leaves) its collective
        assert job.collectives == 1
        with pytest.raises(ss.SupersonicException) as e:
            job.check()
        assert e.value.return_code == ss.INTERRUPTED, e.value
        good_step()                                                         # the job is usable afterwards
    finally:
        dist.destroy_process_group()


# ---- config #5: Sort(d ASC) of the 8-column block -----------------------------------------------------------------------
def check_sort(cols, expect_mode, context, **options):
    view = ss.View(bench.bench_schema(ss), cols)
    op = bench.build_sort_plan(ss, view)
    oschema, want = oracle.run(op)
    plan = ss.Plan(op, make_ctx(**options))
    assert schema_list(plan.result_schema) == oschema
    infos = check_plan(plan, want, context, runs=2)
    assert infos[-1][0]["sort_mode"] in expect_mode, infos
    return infos


def unique_keys(cols):
    # the reference's sort is not stable (sort.h:42): parity needs a key without duplicates
    assert len(np.unique(cols[3])) == len(cols[3])
    return cols


def test_config5_sort_8_columns_one_word_passes_and_record_gather():
    infos = check_sort(unique_keys(bench.host_columns(np, "wide", N_ROWS)), (2,), "config #5")
    assert infos[-1][0]["sort_passes"] == 4, infos       # the four high digits; ties finished by ssgpu_sort_fix_ties_compact


def test_config5_sort_ragged_sizes_one_top_digit_and_mixed_signs():
    # sizes that end inside a tile; keys with ONE top digit (distinct high halves below it) and keys of both signs
    rng = np.random.default_rng(21)
    for n, shape in ((1 << 20, "uniform"), ((1 << 20) + 777, "one_top_digit"), (1500001, "both_signs")):
        cols = bench.host_columns(np, "wide", n, seed=13)
        if shape == "one_top_digit":
            cols[3] = (np.int64(5) << 56) | (rng.permutation(n).astype(np.int64) << 33) | 1
            cols[3][0] = -7
        elif shape == "both_signs":
            cols[3] = np.where(np.arange(n) % 3 == 0, -1 - rng.permutation(n).astype(np.int64) * 1000003, rng.permutation(n).astype(np.int64) * 1000003)
        # (whatever form the keys send the sort to -- the both-signs keys have few distinct high halves: long tie runs, all digits)
        check_sort(unique_keys(cols), (2, 1, 0, 2 + 16, 1 + 16), "config #5, %s, %d rows" % (shape, n))


def test_config5_sort_colliding_high_halves_fall_back_to_all_digits():
    # keys that agree in their high 32 bits in long runs: the tie runs are too long, every digit is sorted after all
    cols = bench.host_columns(np, "wide", N_ROWS, seed=3)
    rng = np.random.default_rng(4)
    hi = rng.integers(-(1 << 30), 1 << 30, 997).astype(np.int64)              # ~2000 rows per high half
    cols[3] = (hi[rng.integers(0, 997, N_ROWS)] << 32) | rng.permutation(N_ROWS).astype(np.int64)
    infos = check_sort(unique_keys(cols), (2 + 16, 1 + 16), "config #5, colliding high halves")
    assert infos[-1][0]["sort_passes"] >= 8, infos


def test_config5_sort_pairs_form_and_column_gathers():
    # the other payload forms of the same query: (key, row id) pairs through the hybrid passes, and column-by-column gathers
    cols = unique_keys(bench.host_columns(np, "wide", N_ROWS, seed=9))
    check_sort(cols, (1,), "config #5, (key, row id) pairs", sort_compact=0)
    check_sort(cols, (0, 1), "config #5, column gathers", sort_records=0)
