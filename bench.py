#!/usr/bin/env python3
"""bench.py -- rows/sec of the fused Filter -> Project -> Aggregate path on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8(d) "Q-FPA-wide"): per GPU a device-resident
100 M-row x 8-column Block (a,b,c,d INT64; d0..d3 DOUBLE) and the reference plan

    ScalarAggregate(SUM(s), COUNT(*), SUM(c), MIN(d), MAX(d0), SUM(d1), SUM(p))
      o Filter(a > 499, ProjectAllAttributes)
      o Compute(a, a+b AS s, c, d, d0, d1, d2*d3 AS p)
      o ScanView(block)

One "step" = one full pass of that plan over the block (inputs already in HBM).  With N > 1
ranks every rank owns an independent row-range shard of the same size (weak scaling) and a
step also all-reduces the partial aggregates over RCCL and finalises them.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes (64 B/row x rows per
launch) / average duration of the pipeline kernel measured with HIP events on the launch
stream inside the library (ssgpu_plan_counters.dominant_ms).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
K_FILTER = 499


def build_plan(ss, view):
    NA = ss.NamedAttribute
    compute = (ss.CompoundExpression().Add(NA("a")).AddAs("s", ss.Plus(NA("a"), NA("b"))).Add(NA("c")).Add(NA("d"))
               .Add(NA("d0")).Add(NA("d1")).AddAs("p", ss.Multiply(NA("d2"), NA("d3"))))
    spec = (ss.AggregationSpecification().AddAggregation(ss.SUM, "s", "sum_s").AddAggregation(ss.COUNT, "", "cnt")
            .AddAggregation(ss.SUM, "c", "sum_c").AddAggregation(ss.MIN, "d", "min_d")
            .AddAggregation(ss.MAX, "d0", "max_d0").AddAggregation(ss.SUM, "d1", "sum_d1")
            .AddAggregation(ss.SUM, "p", "sum_p"))
    return ss.ScalarAggregate(spec, ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(K_FILTER)), ss.ProjectAllAttributes(),
                                              ss.Compute(compute, ss.ScanView(view))))


def bench_schema(ss):
    return ss.TupleSchema([ss.Attribute(n, ss.INT64) for n in ("a", "b", "c", "d")] +
                          [ss.Attribute(n, ss.DOUBLE) for n in ("d0", "d1", "d2", "d3")])


def gen_device_columns(torch, rows, seed, device):
    """Synthetic block of SURVEY 8(d): every partial DOUBLE sum is exact, so the result is
    bit-exact under any reduction order."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    a = torch.randint(0, 1000, (rows,), generator=g, device=device, dtype=torch.int64)
    b = torch.randint(0, 1000, (rows,), generator=g, device=device, dtype=torch.int64)
    c = torch.arange(rows, device=device, dtype=torch.int64) % 100000
    d = torch.randint(-(1 << 62), 1 << 62, (rows,), generator=g, device=device, dtype=torch.int64)
    d0 = torch.randint(-1000000, 1000001, (rows,), generator=g, device=device, dtype=torch.int64).to(torch.float64)
    d1 = torch.randint(0, 4000, (rows,), generator=g, device=device, dtype=torch.int64).to(torch.float64) * 0.25
    d2 = torch.randint(0, 64, (rows,), generator=g, device=device, dtype=torch.int64).to(torch.float64)
    d3 = torch.randint(0, 64, (rows,), generator=g, device=device, dtype=torch.int64).to(torch.float64)
    return [a, b, c, d, d0, d1, d2, d3]


class _DevPtr(object):
    """Expose a raw device buffer to torch through the CUDA array interface."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2}


def pmc_traffic(alg_bytes):
    """HBM bytes per launch of the pipeline kernel from the committed PMC pass
    (profiles/rNN_pmc.json, collected with rocprofv3 --pmc on this same command); PMC
    counters cannot be read from inside the timed process, so the value is only reported
    when the committed pass measured the same workload (same algorithmic bytes)."""
    import glob
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc.json")), reverse=True):
        try:
            with open(path) as f:
                j = json.load(f)
            if int(j.get("algorithmic_bytes_per_launch", -1)) == int(alg_bytes):
                return float(j["traffic_bytes_per_launch"])
        except (OSError, ValueError, KeyError):
            continue
    return None


def cpu_baseline(ss, sample_rows):
    """The oracle (CPU restatement of the reference's 1024-row pull model) on a bounded sample of
    the same workload, 1 thread (the reference is single-threaded per plan)."""
    import numpy as np
    from oracle import oracle
    rng = np.random.default_rng(42)
    n = sample_rows
    cols = [rng.integers(0, 1000, n), rng.integers(0, 1000, n), np.arange(n, dtype=np.int64) % 100000,
            rng.integers(-(1 << 62), 1 << 62, n), rng.integers(-1000000, 1000001, n).astype(np.float64),
            rng.integers(0, 4000, n) * 0.25, rng.integers(0, 64, n).astype(np.float64),
            rng.integers(0, 64, n).astype(np.float64)]
    view = ss.View(bench_schema(ss), cols)
    op = build_plan(ss, view)
    reps, elapsed = 0, 0.0
    while elapsed < 12.0 and reps < 200:
        cur = oracle.Cursor(op)
        t0 = time.perf_counter()
        out_rows = cur.drain_discard()
        elapsed += time.perf_counter() - t0
        reps += 1
        assert out_rows == 1
    return {"value": n * reps / elapsed, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": "%d rows x 8 cols (same plan, seed 42), %d passes, %.1f s of CPU; host has %d cores" % (
                n, reps, elapsed, os.cpu_count() or 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=100_000_000, help="rows per GPU")
    ap.add_argument("--cpu-sample-rows", type=int, default=16_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tile-rows", type=int, default=0)
    ap.add_argument("--lds-target", type=int, default=0)
    ap.add_argument("--grid-limit", type=int, default=0)
    ap.add_argument("--no-events-in-loop", action="store_true",
                    help="do not record the per-kernel HIP events during the timed steps (kernel time from a separate loop)")
    ap.add_argument("--force-distributed", action="store_true",
                    help="take the N > 1 code path (run_partial + RCCL all-reduce + finalize) even with one rank")
    args = ap.parse_args()

    import torch
    import supersonic_amd as ss

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or args.force_distributed
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if "RANK" in os.environ:
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="nccl", device_id=device, rank=0, world_size=1)

    ctx = ss.Context(local_rank)
    # launch on torch's current stream so that RCCL collectives and our kernels are ordered
    stream = torch.cuda.current_stream(device)
    ctx.set_stream(stream.cuda_stream)
    if args.tile_rows:
        ctx.set_option("tile_rows", args.tile_rows)
    if args.lds_target:
        ctx.set_option("lds_target_bytes", args.lds_target)
    if args.grid_limit:
        ctx.set_option("grid_limit", args.grid_limit)

    rows = args.rows
    cols = gen_device_columns(torch, rows, 42 + rank, device)
    torch.cuda.synchronize(device)
    view = ss.DeviceView(bench_schema(ss), [(t.data_ptr(), 0) for t in cols], rows)
    plan = ss.Plan(build_plan(ss, view), ctx)
    row_offset = rank * rows

    seg_tensors = None

    def step():
        nonlocal seg_tensors
        if not distributed:
            plan.run(view)
            return
        segs = plan.run_partial(view, row_offset)
        if seg_tensors is None:
            # the partial-aggregate state is ONE contiguous device buffer of 8 arrays x n_slots
            # 64-bit words (a few hundred bytes): view it as int64 words without a copy
            total = sum(count for (_p, count, _d, _r) in segs)
            assert all(segs[i][0] + segs[i][1] * 8 == segs[i + 1][0] for i in range(len(segs) - 1))
            state = torch.as_tensor(_DevPtr(segs[0][0], total, "<i8"), device=device)
            gathered = torch.empty((world, total), dtype=torch.int64, device=device)
            seg_tensors = (state, gathered)
        state, gathered = seg_tensors
        # ONE collective over RCCL / xGMI (all-gather of the tiny state), then ONE kernel folds the
        # `world` images segment by segment with each segment's operator (ssgpu_plan_fold_partials)
        dist.all_gather_into_tensor(gathered, state)
        plan.fold_partials(gathered.data_ptr(), world)
        plan.finalize()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)

    # the library records HIP events around the pipeline kernel of every run on its launch stream and keeps
    # the last 256 pairs: the timed steps stay fully asynchronous and their kernel durations are read afterwards
    ctx.set_option("profile", 0 if args.no_events_in_loop else 1)
    ctx.set_option("profile_total", 0)      # only the pair around the pipeline kernel, not the whole-run pair
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    dom_ms = []
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    # the per-kernel clock: the HIP events the library recorded around the pipeline kernel of the TIMED
    # steps (the most recent min(K, 256) of them); without them, a dedicated loop after the timed region
    kernel_ms = [] if args.no_events_in_loop else plan.recent_kernel_ms(256)[-args.steps:]
    if not kernel_ms:
        ctx.set_option("profile", 1)
        for _ in range(min(args.steps, 10)):
            step()
            torch.cuda.synchronize(device)
            kernel_ms.append(plan.counters().dominant_ms)
    counters = plan.counters()

    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = plan.fetch()
    if rank == 0:
        total_rows = rows * world * args.steps
        value = total_rows / elapsed
        avg_kernel_s = (sum(kernel_ms) / len(kernel_ms)) / 1e3
        alg_bytes = counters.algorithmic_bytes  # 64 B/row x rows of one launch
        achieved = alg_bytes / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        line = {
            "metric": "rows/sec filter->project->aggregate, 100M x 8 INT64/DOUBLE",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int64/f64", "data": "synthetic",
            "config": {"workload": "Q-FPA-wide: SUM(a+b),COUNT(*),SUM(c),MIN(d),MAX(d0),SUM(d1),SUM(d2*d3) WHERE a>499 "
                                   "over a device-resident %d-row x 8-col block per GPU" % rows,
                       "rows_per_gpu": rows, "parallelism": "row-range shards x%d, one RCCL all-gather of the partial-aggregate state + one fold kernel per step" % world
                       if distributed else "single GPU",
                       "tile_rows": counters.tile_rows, "grid": counters.grid, "lds_bytes": counters.lds_bytes},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(alg_bytes),
                         "kernel": "ssgpu_pipeline_kernel", "kernel_ms": avg_kernel_s * 1e3,
                         "algorithmic_bytes_per_row": alg_bytes / max(rows, 1)},
            "result_row": [result.column(i).data[0].item() for i in range(result.column_count())],
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ss, args.cpu_sample_rows)
        # RCCL prints its version banner through C stdio (still buffered when stdout is a file):
        # drain it first so that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
