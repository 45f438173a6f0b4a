"""N > 1 Sort on CPU processes (gloo, world_size 2): the sample sort of supersonic_amd.distributed.

sharded_sort = local Sort -> splitters from a gathered sample of the first key -> range Filters ->
ONE all-to-all of the rows -> local Sort.  No kernel can run here, so the executor is the CPU oracle;
what is under test is the exchange protocol (count matrix, variable-size all-to-all of typed columns
and NULL masks, empty shards / empty destinations), the range predicates (NULLs first for ASCENDING,
last for DESCENDING, duplicates of a splitter value kept together) and that the concatenation of the
ranks' results reproduces the oracle's single-process Sort of the whole input, ties included."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import supersonic_amd as ss
from supersonic_amd.distributed import sharded_sort, _range_predicate
from oracle import oracle
from helpers import assert_cols_equal


def make_view(n, seed, distinct):
    rng = np.random.default_rng(seed)
    schema = ss.TupleSchema([ss.Attribute("k", ss.INT64, ss.NULLABLE), ss.Attribute("k2", ss.INT32), ss.Attribute("d", ss.DOUBLE, ss.NULLABLE),
                             ss.Attribute("id", ss.INT64)])
    return ss.View(schema, [ss.Column(rng.integers(-distinct, distinct, n), rng.random(n) < 0.1), rng.integers(0, 5, n).astype(np.int32),
                            ss.Column(rng.integers(-4000, 4000, n) * 0.25, rng.random(n) < 0.2), np.arange(n)])


def order_of(desc, second):
    so = ss.SortOrder().add("k", ss.DESCENDING if desc else ss.ASCENDING)
    if second:
        so.add("k2", ss.ASCENDING)
    return so


def oracle_executor(op):
    schema, cols = oracle.run(op)
    ts = ss.TupleSchema([ss.Attribute(n, t, ss.NULLABLE if nullable else ss.NOT_NULLABLE) for (n, t, nullable) in schema])
    return ss.View(ts, [ss.Column(d, z) for (d, z) in cols])


def shard_of(full, lo, hi):
    return ss.View(full.schema(), [ss.Column(full.column(i).data[lo:hi], None if full.column(i).is_null is None else full.column(i).is_null[lo:hi])
                                   for i in range(full.column_count())])


def worker(rank, world, port, n, distinct, desc, second, empty_rank, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = make_view(n, 5, distinct)
    bounds = [0, n, n] if empty_rank == 1 else ([0, 0, n] if empty_rank == 0 else [0, n // 3, n])
    shard = shard_of(full, bounds[rank], bounds[rank + 1])
    out = sharded_sort(order_of(desc, second), ss.ScanView(shard), oracle_executor)
    q.put((rank, [(out.column(i).data, out.column(i).is_null) for i in range(out.column_count())]))
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("n,distinct,desc,second,empty_rank", [
    (20001, 1000, False, True, None), (20001, 1000, True, False, None), (5000, 3, False, True, None),
    (3000, 1000, False, False, 1), (3000, 1000, True, True, 0), (0, 10, False, False, None), (7, 1, False, False, None)])
def test_sharded_sort_over_gloo(n, distinct, desc, second, empty_rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, n, distinct, desc, second, empty_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    _schema, want = oracle.run(ss.Sort(order_of(desc, second), None, 0, ss.ScanView(make_view(n, 5, distinct))))
    got = []
    for c in range(len(want)):
        data = np.concatenate([results[r][c][0] for r in range(2)])
        nulls = [results[r][c][1] for r in range(2)]
        got.append((data, None if nulls[0] is None else np.concatenate(nulls)))
    assert_cols_equal(got, want, context="sharded sort")   # exact row order: ties are in original order


def test_range_predicates_partition_every_row_once():
    # every row (NULL keys included) belongs to exactly one destination, for 3 ranks and duplicate splitters
    view = make_view(4000, 9, 6)
    for desc in (False, True):
        for splitters in ([-2, 3], [1, 1], []):
            total = 0
            for d in range(3):
                pred = _range_predicate("k", ss.INT64, True, splitters, d, 3, desc)
                _s, cols = oracle.run(ss.Filter(pred, ss.ProjectAllAttributes(), ss.ScanView(view)))
                total += len(cols[0][0])
            assert total == 4000


def test_range_predicates_keep_nan_keys():
    # a DOUBLE first key with NaNs: every comparison with a NaN is false, so "key > lo AND key <= hi" buckets would match
    # none of them and the rows would vanish from the global result; the last ascending bucket is the complement instead
    rng = np.random.default_rng(4)
    n = 3000
    key = rng.integers(-50, 50, n) * 0.5
    key[rng.random(n) < 0.05] = np.nan
    schema = ss.TupleSchema([ss.Attribute("k", ss.DOUBLE, ss.NULLABLE), ss.Attribute("id", ss.INT64)])
    view = ss.View(schema, [ss.Column(key, rng.random(n) < 0.1), np.arange(n)])
    from supersonic_amd.distributed import _choose_splitters
    assert not np.isnan(_choose_splitters(np.sort(key), 3)).any()       # NaNs (sorted last by numpy) are never splitters
    for desc in (False, True):
        for splitters in ([-3.5, 7.0], [2.0, 2.0]):
            seen = np.zeros(n, int)
            for d in range(3):
                pred = _range_predicate("k", ss.DOUBLE, True, splitters, d, 3, desc)
                _s, cols = oracle.run(ss.Filter(pred, ss.ProjectAllAttributes(), ss.ScanView(view)))
                seen[cols[1][0]] += 1
            assert (seen == 1).all()


# ---- STRING payload columns cross the all-to-all (host Views carry the byte strings themselves) -----------------------
def make_string_view(n, seed=9):
    rng = np.random.default_rng(seed)
    words = [b"pear", b"apple", b"fig", b"", b"kiwi\x00k", b"zebra"]
    schema = ss.TupleSchema([ss.Attribute("k", ss.INT64), ss.Attribute("s", ss.STRING, ss.NULLABLE), ss.Attribute("id", ss.INT64)])
    s = np.empty(n, dtype=object); s[:] = [words[i] for i in rng.integers(0, len(words), n)]
    return ss.View(schema, [rng.integers(-50, 50, n), ss.Column(s, rng.random(n) < 0.15), np.arange(n)])


def string_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = make_string_view(n)
    bounds = [0, n // 3, n]
    so = ss.SortOrder().add("k", ss.ASCENDING).add("s", ss.DESCENDING)
    out = sharded_sort(so, ss.ScanView(shard_of(full, bounds[rank], bounds[rank + 1])), oracle_executor)
    q.put((rank, [(out.column(i).data, out.column(i).is_null) for i in range(out.column_count())]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_sort_with_string_columns_over_gloo():
    n = 4001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=string_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    so = ss.SortOrder().add("k", ss.ASCENDING).add("s", ss.DESCENDING)
    _schema, want = oracle.run(ss.Sort(so, None, 0, ss.ScanView(make_string_view(n))))
    got = []
    for c in range(len(want)):
        data = np.concatenate([results[r][c][0] for r in range(2)])
        nulls = [results[r][c][1] for r in range(2)]
        got.append((data, None if nulls[0] is None else np.concatenate(nulls)))
    assert_cols_equal(got, want, context="sharded sort, STRING payload and second key")
