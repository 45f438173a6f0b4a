"""TWO REAL RANKS on the one GPU of the test box.  RCCL refuses two ranks of one device, so the ranks talk gloo and every collective
of the device drivers goes through host copies (distributed._HostBounce) -- but everything else is the product path: two processes,
two contexts, each rank's shard scanned by the HIP kernels, and the exchange kernels -- pack / route / unpack of result images, the
dense-slot fold, the scalar state fold, the sample sort's range filters -- fed with images ANOTHER PROCESS produced.  (The gloo tests of
tests/test_distributed_*_gloo.py cover the protocols with the oracle as executor; the one-rank tests cover the kernels against
themselves; several ranks simulated in one process cover the folds.  This closes the gap between them as far as one GPU can.)"""
import os
import socket

import numpy as np
import pytest

import supersonic_amd as ss
from oracle import oracle
from helpers import sort_rows, assert_cols_equal, to_cols

pytestmark = pytest.mark.gpu
NA = ss.NamedAttribute
N = 400003


def make_view(n, seed=17):
    rng = np.random.default_rng(seed)
    schema = ss.TupleSchema([ss.Attribute("a", ss.INT64), ss.Attribute("k1", ss.INT32), ss.Attribute("k2", ss.INT32),
                             ss.Attribute("v", ss.INT64, ss.NULLABLE), ss.Attribute("d", ss.DOUBLE), ss.Attribute("e", ss.DOUBLE)])
    g = rng.integers(0, 30000, n)
    return ss.View(schema, [rng.integers(0, 1000, n), (g // 173).astype(np.int32), (g % 173).astype(np.int32),
                            ss.Column(rng.integers(-1000, 1000, n), rng.random(n) < 0.2), rng.integers(-4000, 4000, n) * 0.25, rng.integers(-100000, 100000, n) * 0.125])


def shard_of(full, lo, hi):
    return ss.View(full.schema(), [ss.Column(full.column(i).data[lo:hi], None if full.column(i).is_null is None else full.column(i).is_null[lo:hi])
                                   for i in range(full.column_count())])


def group_spec(first_last):
    spec = (ss.AggregationSpecification().AddAggregation(ss.SUM, "v", "sv").AddAggregation(ss.MIN, "v", "mnv").AddAggregation(ss.MAX, "d", "mxd")
            .AddAggregation(ss.SUM, "d", "sd").AddAggregation(ss.SUM, "e", "se").AddAggregation(ss.COUNT, "v", "cv").AddAggregation(ss.COUNT, "", "n"))
    if first_last:      # (values that live in the shard that saw the row: the image exchanges carry them, the dense exchange refuses)
        spec.AddAggregation(ss.FIRST, "v", "fv").AddAggregation(ss.LAST, "d", "ld")
    return spec


def child(view):
    return ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(299)), ss.ProjectAllAttributes(), ss.ScanView(view))


def scalar_op(view):
    spec = (ss.AggregationSpecification().AddAggregation(ss.SUM, "v", "sv").AddAggregation(ss.COUNT, "", "n").AddAggregation(ss.MIN, "d", "mn")
            .AddAggregation(ss.MAX, "v", "mx").AddAggregation(ss.SUM, "d", "sd").AddAggregation(ss.FIRST, "v", "fv").AddAggregation(ss.LAST, "k1", "lk"))
    return ss.ScalarAggregate(spec, child(view))


def pack(view):
    return [(view.column(i).data, view.column(i).is_null) for i in range(view.column_count())]


def worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from supersonic_amd.distributed import (DeviceShardedGroupAggregate, DenseShardedGroupAggregate, PlanDenseBackend, device_sharded_sort, _dist_for, _DevPtr)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    try:
        full = make_view(N)
        bounds = [0, N // 3, N] if world == 2 else [0, N // 3, N // 3, N]      # (three ranks: the middle one holds no row)
        shard = shard_of(full, bounds[rank], bounds[rank + 1])
        ctx = ss.Context(0)
        ctx.set_option("specialize", 0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        # (1) / (2) result images: all-gather and key-range all-to-all + merge plan
        for exchange in ("all_gather", "key_range"):
            job = DeviceShardedGroupAggregate(ctx, ["k1", "k2"], group_spec(True), child(shard), exchange=exchange)
            for _ in range(2):
                job.step()
                while not job.check():
                    job.step()
            out[exchange] = (pack(job.gather_result()), job.collectives)
        # (3) dense slots: one all-to-all of slot slices + element-wise fold
        op = ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), group_spec(False), None, child(shard))
        backend = PlanDenseBackend(ctx, op)
        dense = DenseShardedGroupAggregate(backend)
        for _ in range(2):
            dense.step(shard)
            while not dense.check():
                dense.step(shard)
        out["dense"] = (pack(dense.gather_result()), dense.collectives, dense.layout["slots"])
        # (4) scalar aggregate: partial state -> all-gather -> one fold + emit launch (bench.py's N > 1 step)
        plan = ss.Plan(scalar_op(shard), ctx)
        segs = plan.run_partial(shard, bounds[rank])
        total = sum(count for (_p, count, _d, _r) in segs)
        device = torch.device("cuda", 0)
        state = torch.as_tensor(_DevPtr(segs[0][0], total, "<i8"), device=device)
        gathered = torch.empty((world, total), dtype=torch.int64, device=device)
        ctx.synchronize()
        _dist_for(None).all_gather_into_tensor(gathered, state)
        torch.cuda.synchronize()
        plan.fold_finalize(gathered.data_ptr(), world)
        out["scalar"] = pack(plan.fetch())
        # (5) sample sort: local sort, splitters, one all-to-all of rows, sort of the arrivals
        order = ss.SortOrder().add("d", ss.ASCENDING).add("a", ss.DESCENDING)
        splan, _dv = device_sharded_sort(ctx, order, shard, always_exchange=True)
        out["sort"] = pack(splan.fetch())
        q.put((rank, out, None))
    except Exception as e:      # noqa: BLE001  (the parent reports it)
        import traceback
        q.put((rank, None, traceback.format_exc() + repr(e)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_processes_share_the_gpu_and_exchange_real_images(world):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, out, err = q.get(timeout=600)
        assert err is None, "rank %d: %s" % (rank, err)
        results[rank] = out
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    full = make_view(N)
    _s, want_fl = oracle.run(ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), group_spec(True), None, child(full)))
    _s, want = oracle.run(ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), group_spec(False), None, child(full)))
    _s, want_scalar = oracle.run(scalar_op(full))
    _s, want_sorted = oracle.run(ss.Sort(ss.SortOrder().add("d", ss.ASCENDING).add("a", ss.DESCENDING), None, 0, ss.ScanView(full)))
    for rank in range(world):
        out = results[rank]
        for exchange in ("all_gather", "key_range"):
            cols, collectives = out[exchange]
            assert collectives == 1
            assert_cols_equal(sort_rows(cols), sort_rows(want_fl), context="rank %d %s" % (rank, exchange))   # (the DOUBLE columns sum exactly)
        cols, collectives, slots = out["dense"]
        assert collectives == 1 and slots > 0
        assert_cols_equal(sort_rows(cols), sort_rows(want), context="rank %d dense" % rank)
        assert_cols_equal(out["scalar"], want_scalar, context="rank %d scalar" % rank)
    # the sorted slices, concatenated in rank order, are the global order (keys: d ascending, a descending; ties in any order)
    merged = [np.concatenate([results[r]["sort"][i][0] for r in range(world)]) for i in range(6)]
    assert np.array_equal(merged[4], want_sorted[4][0]) and np.array_equal(merged[0], want_sorted[0][0])
    assert sorted(merged[1].tolist()) == sorted(want_sorted[1][0].tolist())


@pytest.mark.parametrize("query,extra", [("wide", []), ("group", []), ("group", ["--exchange", "key_range"]), ("wide", ["--scaling", "strong"])])
def test_bench_py_takes_its_multi_rank_path_with_two_ranks_on_one_gpu(query, extra):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), with SSGPU_BENCH_SHARE_GPU=1: both
    ranks on device 0, collectives over gloo through the host.  The timings mean nothing; the line must come out, carry both regimes and
    -- for the GroupAggregate -- the checked group count."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SSGPU_BENCH_SHARE_GPU="1")
    env.pop("SSGPU_SPECIALIZE", None); env.pop("SSGPU_GROUP_DENSE", None)      # (the library's own defaults, as in the driver's run)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", "1000000", "--query", query, "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                   # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "development_mode" in line["config"]
    assert "roofline" in line and line["roofline"]["kernel_ms"] > 0
    if query == "group":
        assert line["result_row"]["groups"] > 90000 and line["config"]["collectives_per_step"] == 1
    if "--scaling" not in extra:
        assert set(line["regimes"]) == {"weak", "strong"}
