"""N > 1 path on CPU processes (gloo, world_size 2): the multi-GPU exchange of the sharded
ScalarAggregate is eight element-wise all-reduces over the partial-aggregate STATE arrays
(ssgpu_plan_partial_segments: sum / count / double-double hi, lo / min / max / float min, max).

No kernel can run here, so each rank derives its shard's slot records with the CPU oracle, encodes
them exactly as `ssgpu_slots_to_state_kernel` does, all-reduces with the segment's reduce op
(the same calls bench.py issues over RCCL), decodes as `ssgpu_state_to_slots_kernel` does and must
obtain the oracle's answer for the WHOLE input -- including empty shards, all-NULL inputs and the
sign-corrected integer min/max domain."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import supersonic_amd as ss
from oracle import oracle

I64_MIN = -(1 << 63)
SIGN = np.uint64(1 << 63)


def make_view(n, seed, with_nulls):
    rng = np.random.default_rng(seed)
    schema = ss.TupleSchema([ss.Attribute("i", ss.INT64, ss.NULLABLE), ss.Attribute("u", ss.UINT64), ss.Attribute("d", ss.DOUBLE, ss.NULLABLE)])
    nulls = (rng.random(n) < (1.0 if with_nulls == "all" else 0.2)) if with_nulls else np.zeros(n, bool)
    return ss.View(schema, [ss.Column(rng.integers(-1000, 1000, n), nulls), rng.integers(0, 1 << 63, n).astype(np.uint64) * np.uint64(2),
                            ss.Column(rng.integers(-4000, 4000, n) * 0.25, nulls)])


def spec():
    return (ss.AggregationSpecification().AddAggregation(ss.SUM, "i", "si").AddAggregation(ss.MIN, "i", "mni")
            .AddAggregation(ss.MAX, "i", "mxi").AddAggregation(ss.MIN, "u", "mnu").AddAggregation(ss.MAX, "u", "mxu")
            .AddAggregation(ss.SUM, "d", "sd").AddAggregation(ss.MIN, "d", "mnd").AddAggregation(ss.MAX, "d", "mxd")
            .AddAggregation(ss.COUNT, "i", "ci").AddAggregation(ss.COUNT, "", "n"))


KINDS = ["sum_i", "min_i", "max_i", "min_u", "max_u", "sum_d", "min_d", "max_d", "count", "count"]


def encode(cols):
    """oracle result row -> the 8 state arrays (one element per slot), as slots_to_state does."""
    ns = len(KINDS)
    st = {"sum": np.zeros(ns, np.int64), "cnt": np.zeros(ns, np.int64), "hi": np.zeros(ns), "lo": np.zeros(ns),
          "mn": np.full(ns, np.iinfo(np.int64).max, np.int64), "mx": np.full(ns, I64_MIN, np.int64),
          "mnf": np.full(ns, np.inf), "mxf": np.full(ns, -np.inf)}
    for s, (kind, (d, z)) in enumerate(zip(KINDS, cols)):
        present = not (z is not None and z[0])
        if kind == "count":
            st["sum"][s] = int(d[0]); st["cnt"][s] = int(d[0])
            continue
        st["cnt"][s] = 1 if present else 0     # any positive count marks "not NULL"
        if not present:
            continue
        if kind == "sum_i":
            st["sum"][s] = int(d[0])
        elif kind == "sum_d":
            st["hi"][s] = float(d[0])
        elif kind in ("min_i", "max_i"):       # key_i64 then the sign-corrected view == the value itself
            st["mn" if kind == "min_i" else "mx"][s] = int(d[0])
        elif kind in ("min_u", "max_u"):       # unsigned order -> flip the top bit, compare as signed
            v = (np.uint64(d[0]) ^ SIGN).astype(np.uint64).view(np.int64)
            st["mn" if kind == "min_u" else "mx"][s] = int(v)
        elif kind == "min_d":
            st["mnf"][s] = float(d[0])
        elif kind == "max_d":
            st["mxf"][s] = float(d[0])
    return st


def decode(st):
    out = []
    for s, kind in enumerate(KINDS):
        cnt = int(st["cnt"][s])
        if kind == "count":
            out.append((int(st["sum"][s]), False)); continue
        null = cnt == 0
        if kind == "sum_i": v = int(st["sum"][s])
        elif kind == "sum_d": v = float(st["hi"][s] + st["lo"][s])
        elif kind == "min_i": v = int(st["mn"][s])
        elif kind == "max_i": v = int(st["mx"][s])
        elif kind == "min_u": v = int(np.int64(st["mn"][s]).view(np.uint64) ^ SIGN)
        elif kind == "max_u": v = int(np.int64(st["mx"][s]).view(np.uint64) ^ SIGN)
        elif kind == "min_d": v = float(st["mnf"][s])
        else: v = float(st["mxf"][s])
        out.append((v, null))
    return out


def worker(rank, world, port, n, with_nulls, empty_rank, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = make_view(n, 5, with_nulls)
    bounds = [0, n, n] if empty_rank == 1 else ([0, 0, n] if empty_rank == 0 else [0, n // 2, n])
    lo, hi = bounds[rank], bounds[rank + 1]
    shard = ss.View(full.schema(), [ss.Column(full.column(i).data[lo:hi], None if full.column(i).is_null is None else full.column(i).is_null[lo:hi])
                                    for i in range(full.column_count())])
    _s, cols = oracle.run(ss.ScalarAggregate(spec(), ss.ScanView(shard)))
    st = encode(cols)
    ops = {"sum": dist.ReduceOp.SUM, "cnt": dist.ReduceOp.SUM, "hi": dist.ReduceOp.SUM, "lo": dist.ReduceOp.SUM,
           "mn": dist.ReduceOp.MIN, "mx": dist.ReduceOp.MAX, "mnf": dist.ReduceOp.MIN, "mxf": dist.ReduceOp.MAX}
    for k, arr in st.items():
        t = torch.from_numpy(arr)
        dist.all_reduce(t, op=ops[k])
    if rank == 0:
        q.put(decode(st))
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("n,with_nulls,empty_rank", [(10001, False, None), (10001, "some", None), (513, "all", None),
                                                     (4000, "some", 1), (4000, False, 0), (0, False, None)])
def test_sharded_scalar_aggregate_state_reduces_over_gloo(n, with_nulls, empty_rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, n, with_nulls, empty_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    _s, want = oracle.run(ss.ScalarAggregate(spec(), ss.ScanView(make_view(n, 5, with_nulls))))
    for (v, null), (d, z) in zip(got, want):
        wnull = bool(z is not None and z[0])
        assert null == wnull
        if not null:
            assert v == d[0].item(), (v, d[0])
