"""N > 1 GroupAggregate on CPU processes (gloo, world_size 2): SURVEY 8(e) / BASELINE config #4.

supersonic_amd.distributed.sharded_group_aggregate = per-shard GroupAggregate -> all-gather of the
partial group tables -> local merge GroupAggregate.  No kernel can run here, so the executor handed
to it is the CPU oracle; what is under test is the exchange (variable-size all-gather of typed
columns and NULL masks, empty shards) and the merge plan (SUM of sums, MIN of mins, MAX of maxes,
SUM of counts, COUNT kept NOT NULL), which must reproduce the oracle's answer for the WHOLE input.
The same function runs over RCCL with the device executor (tests/test_parity_gpu.py covers the
merge plan on the GPU with world_size 1)."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import supersonic_amd as ss
from supersonic_amd.distributed import sharded_group_aggregate
from oracle import oracle
from helpers import sort_rows, assert_cols_equal

NA = ss.NamedAttribute


def make_view(n, seed=11):
    rng = np.random.default_rng(seed)
    schema = ss.TupleSchema([ss.Attribute("a", ss.INT64), ss.Attribute("k1", ss.INT32), ss.Attribute("k2", ss.INT32, ss.NULLABLE),
                             ss.Attribute("v", ss.INT64, ss.NULLABLE), ss.Attribute("d", ss.DOUBLE)])
    return ss.View(schema, [rng.integers(0, 1000, n), rng.integers(0, 40, n).astype(np.int32),
                            ss.Column(rng.integers(0, 5, n).astype(np.int32), rng.random(n) < 0.1),
                            ss.Column(rng.integers(-1000, 1000, n), rng.random(n) < 0.3),
                            rng.integers(-4000, 4000, n) * 0.25])


def spec():
    return (ss.AggregationSpecification().AddAggregation(ss.SUM, "v", "sv").AddAggregation(ss.MIN, "v", "mnv")
            .AddAggregation(ss.MAX, "d", "mxd").AddAggregation(ss.SUM, "d", "sd")
            .AddAggregation(ss.COUNT, "v", "cv").AddAggregation(ss.COUNT, "", "n")
            .AddAggregation(ss.FIRST, "v", "fv").AddAggregation(ss.LAST, "d", "ld"))


def child(view, with_filter):
    op = ss.ScanView(view)
    if with_filter:
        op = ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(499)), ss.ProjectAllAttributes(), op)
    return op


def oracle_executor(op):
    schema, cols = oracle.run(op)
    ts = ss.TupleSchema([ss.Attribute(n, t, ss.NULLABLE if nullable else ss.NOT_NULLABLE) for (n, t, nullable) in schema])
    return ss.View(ts, [ss.Column(d, z) for (d, z) in cols])


def shard_of(full, lo, hi):
    return ss.View(full.schema(), [ss.Column(full.column(i).data[lo:hi], None if full.column(i).is_null is None else full.column(i).is_null[lo:hi])
                                   for i in range(full.column_count())])


def worker(rank, world, port, n, with_filter, empty_rank, q, key_range=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = make_view(n)
    bounds = [0, n, n] if empty_rank == 1 else ([0, 0, n] if empty_rank == 0 else [0, n // 3, n])
    shard = shard_of(full, bounds[rank], bounds[rank + 1])
    out = sharded_group_aggregate(["k1", "k2"], spec(), child(shard, with_filter), oracle_executor, key_range=key_range)
    cols = [(out.column(i).data, out.column(i).is_null) for i in range(out.column_count())]
    schema = [(out.schema().attribute(i).name(), out.schema().attribute(i).type(), out.schema().attribute(i).is_nullable())
              for i in range(out.schema().attribute_count())]
    q.put((rank, schema, cols))
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("key_range", [False, True])
@pytest.mark.parametrize("n,with_filter,empty_rank", [(20001, False, None), (20001, True, None), (3000, True, 1), (3000, False, 0), (0, False, None)])
def test_sharded_group_aggregate_over_gloo(n, with_filter, empty_rank, key_range):
    # key_range: a partial row travels only to the rank that owns its key (hash of the group keys modulo the world size), every
    # rank merges its own slice of the groups and the finished slices are gathered -- same answer, 1 / world of the merge work
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, n, with_filter, empty_rank, q, key_range)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want_schema, want = oracle.run(ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), spec(), None, child(make_view(n), with_filter)))
    for _rank, schema, cols in results:           # every rank holds the full answer
        assert [tuple(x) for x in schema] == [tuple(x) for x in want_schema]
        assert_cols_equal(sort_rows(cols), sort_rows(want), context="sharded group aggregate")


# ---- STRING keys and STRING MIN / MAX across shards: host Views carry the byte strings themselves ---------------------
WORDS = [b"pear", b"apple", b"fig", b"kiwi", b"", b"apple pie", b"zebra\x00x"]


def make_string_view(n, seed=5):
    rng = np.random.default_rng(seed)
    schema = ss.TupleSchema([ss.Attribute("name", ss.STRING, ss.NULLABLE), ss.Attribute("v", ss.INT64), ss.Attribute("tag", ss.STRING)])
    names = np.empty(n, dtype=object); names[:] = [WORDS[i] for i in rng.integers(0, len(WORDS), n)]
    tags = np.empty(n, dtype=object); tags[:] = [WORDS[i] for i in rng.integers(0, len(WORDS), n)]
    return ss.View(schema, [ss.Column(names, rng.random(n) < 0.1), rng.integers(-50, 50, n), tags])


def string_spec():
    return (ss.AggregationSpecification().AddAggregation(ss.SUM, "v", "sv").AddAggregation(ss.MIN, "tag", "lo")
            .AddAggregation(ss.MAX, "tag", "hi").AddAggregation(ss.COUNT, "", "n"))


def string_worker(rank, world, port, n, q, key_range=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = make_string_view(n)
    bounds = [0, n // 4, n]
    out = sharded_group_aggregate(["name"], string_spec(), ss.ScanView(shard_of(full, bounds[rank], bounds[rank + 1])), oracle_executor, key_range=key_range)
    q.put((rank, [(out.column(i).data, out.column(i).is_null) for i in range(out.column_count())]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("key_range", [False, True])
def test_sharded_group_aggregate_with_string_columns_over_gloo(key_range):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=string_worker, args=(r, 2, port, 5003, q, key_range)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    _schema, want = oracle.run(ss.GroupAggregate(ss.ProjectNamedAttributes(["name"]), string_spec(), None, ss.ScanView(make_string_view(5003))))
    for _rank, cols in results:
        assert_cols_equal(sort_rows(cols), sort_rows(want), context="sharded group aggregate, STRING key")


def test_job_strings_agree_across_ranks():
    from supersonic_amd.distributed import job_strings
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=job_strings_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1] == sorted([b"", b"a", b"b", b"c", b"zz"])
    # plans created with the job's strings give every rank the same order-preserving codes
    d0 = ss.StringDictionary(got[0])
    assert [d0.code_of(w) for w in got[0]] == list(range(len(got[0])))
    assert job_strings is not None


def job_strings_worker(rank, world, port, q):
    from supersonic_amd.distributed import job_strings
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank, job_strings([b"b", "a", b"zz"] if rank == 0 else [b"c", b"", b"a"])))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_merge_specs_carry_double_sum_residuals():
    # distributed.py: every DOUBLE SUM of the job travels as (SUM, SUM_RESIDUAL); other aggregates are unchanged and the
    # merge plan's result has the original columns only (checked end to end on the GPU in test_double_sum_gpu.py)
    import numpy as np
    import supersonic_amd as ss
    from supersonic_amd.distributed import _shard_spec, _merge_spec, RESIDUAL
    schema = ss.TupleSchema([ss.Attribute("k", ss.INT32), ss.Attribute("d", ss.DOUBLE, ss.NULLABLE), ss.Attribute("i", ss.INT64), ss.Attribute("f", ss.FLOAT)])
    spec = (ss.AggregationSpecification().AddAggregation(ss.SUM, "d", "sd").AddAggregation(ss.SUM, "i", "si").AddAggregation(ss.MIN, "d", "mn")
            .AddAggregation(ss.COUNT, "d", "c").AddAggregation(ss.SUM, "f", "sf"))
    shard, with_residual = _shard_spec(spec, schema)
    assert with_residual == ["sd"]
    assert [(e[0], e[3], e[4]) for e in shard.elements] == [(ss.SUM, "d", "sd"), (ss.SUM_RESIDUAL, "d", "sd" + RESIDUAL), (ss.SUM, "i", "si"),
                                                            (ss.MIN, "d", "mn"), (ss.COUNT, "d", "c"), (ss.SUM, "f", "sf")]
    merged, counts = _merge_spec(spec, with_residual)
    assert counts == ["c"]
    assert [(e[0], e[3]) for e in merged.elements] == [(ss.SUM, "sd"), (ss.SUM, "sd" + RESIDUAL), (ss.SUM, "si"), (ss.MIN, "mn"), (ss.SUM, "c"), (ss.SUM, "sf")]
    # binds (device-less context): a residual without its SUM, or over a non-DOUBLE column, is a bind error
    view = ss.View(schema, [np.zeros(3, np.int32), ss.Column(np.zeros(3), np.zeros(3, bool)), np.zeros(3, np.int64), np.zeros(3, np.float32)])
    ss.Plan(ss.GroupAggregate(ss.ProjectNamedAttributes(["k"]), shard, None, ss.ScanView(view)), ss.Context(-1))
    for bad in (ss.AggregationSpecification().AddAggregation(ss.SUM_RESIDUAL, "d", "r"),
                ss.AggregationSpecification().AddAggregation(ss.SUM, "i", "s").AddAggregation(ss.SUM_RESIDUAL, "i", "r")):
        try:
            ss.Plan(ss.GroupAggregate(ss.ProjectNamedAttributes(["k"]), bad, None, ss.ScanView(view)), ss.Context(-1))
            raise AssertionError("bound")
        except ss.SupersonicException as e:
            assert e.return_code == ss.ERROR_INVALID_ARGUMENT_TYPE


def test_sum_of_floating_values_into_an_integer_is_not_merged_across_shards():
    # aggregation_operators.h:173-185: the reference adds and truncates row after row, so a shard's result is not a partial sum of
    # the job's -- refused when the job is set up (on one GPU the rows are folded in order: test_parity_gpu.py)
    import pytest
    import supersonic_amd as ss
    from supersonic_amd.distributed import _shard_spec
    schema = ss.TupleSchema([ss.Attribute("k", ss.INT32), ss.Attribute("d", ss.DOUBLE, ss.NULLABLE), ss.Attribute("f", ss.FLOAT)])
    for col, t in (("d", ss.INT64), ("f", ss.INT32), ("d", ss.UINT64)):
        with pytest.raises(ss.SupersonicException) as e:
            _shard_spec(ss.AggregationSpecification().AddAggregationWithDefinedOutputType(ss.SUM, col, "s", t), schema)
        assert e.value.return_code == ss.ERROR_NOT_IMPLEMENTED
    _shard_spec(ss.AggregationSpecification().AddAggregationWithDefinedOutputType(ss.MAX, "d", "m", ss.INT64), schema)   # (MIN / MAX truncate monotonically: merged)


def test_never_null_partial_columns_are_declared_not_null_for_the_merge():
    # distributed.py: SUM / MIN / MAX / FIRST / LAST (and a sum's residual) of a NOT NULL input are never NULL in a shard's table
    # -- the merge plan reads them as NOT NULL columns (no contribution counts); COUNT, DISTINCT and NULLABLE inputs are left alone
    import supersonic_amd as ss
    from supersonic_amd.distributed import _shard_spec, _never_null, _declare_not_null, RESIDUAL
    schema = ss.TupleSchema([ss.Attribute("k", ss.INT32), ss.Attribute("d", ss.DOUBLE), ss.Attribute("n", ss.DOUBLE, ss.NULLABLE), ss.Attribute("i", ss.INT64)])
    spec = (ss.AggregationSpecification().AddAggregation(ss.SUM, "d", "sd").AddAggregation(ss.MIN, "n", "mn").AddAggregation(ss.COUNT, "i", "c")
            .AddAggregation(ss.LAST, "i", "li").AddAggregation(ss.MAX, "d", "mx"))
    shard, _with_residual = _shard_spec(spec, schema)
    names = _never_null(shard, schema)
    assert names == {"sd", "sd" + RESIDUAL, "li", "mx"}
    assert _never_null(shard, None) == set()
    result = ss.TupleSchema([ss.Attribute("k", ss.INT32), ss.Attribute("sd", ss.DOUBLE, ss.NULLABLE), ss.Attribute("sd" + RESIDUAL, ss.DOUBLE, ss.NULLABLE),
                             ss.Attribute("mn", ss.DOUBLE, ss.NULLABLE), ss.Attribute("c", ss.UINT64), ss.Attribute("li", ss.INT64, ss.NULLABLE),
                             ss.Attribute("mx", ss.DOUBLE, ss.NULLABLE), ss.Attribute("__valid", ss.BOOL)])
    view = ss.DeviceView(result, [(1000 + 16 * i, 2000 + 16 * i) for i in range(8)], 5)
    out = _declare_not_null(view, names)
    got = [(out.schema().attribute(i).name(), out.schema().attribute(i).is_nullable(), out._ptrs[i]) for i in range(8)]
    assert got == [("k", False, (1000, 2000)), ("sd", False, (1016, 0)), ("sd" + RESIDUAL, False, (1032, 0)), ("mn", True, (1048, 2048)),
                   ("c", False, (1064, 2064)), ("li", False, (1080, 0)), ("mx", False, (1096, 0)), ("__valid", False, (1112, 2112))]
    assert out.row_count() == 5 and _declare_not_null(view, set()) is view


# ---- DISTINCT, CONCAT and the row-after-row SUM across shards: what a shard's result is NOT a partial result of --------------------
def _hard_spec(kind):
    spec = ss.AggregationSpecification()
    if kind == "distinct":          # the shards send their distinct (keys, value) pairs next to the partial table
        return (spec.AddDistinctAggregation(ss.COUNT, "v", "cdv").AddAggregation(ss.SUM, "d", "sd").AddDistinctAggregation(ss.SUM, "v", "sdv")
                .AddAggregation(ss.COUNT, "", "n").AddDistinctAggregation(ss.COUNT, "a", "cda").AddAggregation(ss.LAST, "v", "lv").AddDistinctAggregation(ss.COUNT, "k2", "cdk"))
    if kind == "distinct_only":
        return spec.AddDistinctAggregation(ss.SUM, "v", "sdv")
    if kind == "concat":            # the shards send the rows themselves
        return spec.AddAggregation(ss.CONCAT, "v", "cv").AddAggregation(ss.SUM, "d", "sd").AddDistinctAggregation(ss.COUNT, "a", "cda").AddAggregation(ss.COUNT, "", "n")
    return spec.AddAggregationWithDefinedOutputType(ss.SUM, "d", "sq", ss.INT64).AddAggregation(ss.MAX, "v", "mv").AddAggregation(ss.FIRST, "a", "fa")


def hard_worker(rank, world, port, n, kind, q, key_range):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = make_view(n)
    bounds = [0, n // 3, n]
    out = sharded_group_aggregate(["k1", "k2"], _hard_spec(kind), child(shard_of(full, bounds[rank], bounds[rank + 1]), True), oracle_executor, key_range=key_range)
    cols = [(out.column(i).data, out.column(i).is_null) for i in range(out.column_count())]
    schema = [(out.schema().attribute(i).name(), out.schema().attribute(i).type(), out.schema().attribute(i).is_nullable())
              for i in range(out.schema().attribute_count())]
    q.put((rank, schema, cols))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("key_range", [False, True])
@pytest.mark.parametrize("kind", ["distinct", "distinct_only", "concat", "sequential"])
@pytest.mark.parametrize("n", [6001, 0])
def test_sharded_group_aggregate_of_unmergeable_aggregates_over_gloo(n, kind, key_range):
    """COUNT(DISTINCT v) of a group is not the sum of the shards' counts, a CONCAT not a concatenation of partial strings in any
    order, `*result += val` into an integer not a sum of truncated sums: the first travels as distinct (keys, value) pairs stacked under
    the partial table, the other two as the rows themselves.  Every rank ends with the oracle's answer for the WHOLE input."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=hard_worker, args=(r, 2, port, n, kind, q, key_range)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want_schema, want = oracle.run(ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), _hard_spec(kind), None, child(make_view(n), True)))
    for _rank, schema, cols in results:
        assert [tuple(x) for x in schema] == [tuple(x) for x in want_schema]
        assert_cols_equal(sort_rows(cols), sort_rows(want), context="sharded %s" % kind)
