"""What a HOST-resident input costs (the PCIe-inclusive rate of the headline query): upload everything, then run -- against chunked
staging (ssgpu_plan_run_host), where chunk k + 1 is copied while chunk k is read.  Pinned host columns (torch pin_memory).

    python tools/host_staging_bench.py [rows] > gpurun_out/host_staging.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

import supersonic_amd as ss

NA = ss.NamedAttribute


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    rng = np.random.default_rng(42)
    names = ["a", "b", "c", "d", "d0", "d1", "d2", "d3"]
    types = [ss.INT64] * 4 + [ss.DOUBLE] * 4
    cols = []
    for i, t in enumerate(types):
        pinned = torch.empty(n, dtype=torch.int64 if t == ss.INT64 else torch.float64).pin_memory()
        arr = pinned.numpy()
        if i < 2:
            arr[:] = rng.integers(0, 1000, n)
        elif i == 2:
            arr[:] = np.arange(n) % 100000
        elif i == 3:
            arr[:] = rng.integers(-(1 << 62), 1 << 62, n)
        elif i == 4:
            arr[:] = rng.integers(-1000000, 1000001, n)
        elif i == 5:
            arr[:] = rng.integers(0, 4000, n) * 0.25
        else:
            arr[:] = rng.integers(0, 64, n)
        cols.append((pinned, arr))
    view = ss.View(ss.TupleSchema([ss.Attribute(nm, t) for nm, t in zip(names, types)]), [a for _p, a in cols])
    e = (ss.CompoundExpression().Add(NA("a")).AddAs("s", ss.Plus(NA("a"), NA("b"))).Add(NA("c")).Add(NA("d")).Add(NA("d0")).Add(NA("d1"))
         .AddAs("p", ss.Multiply(NA("d2"), NA("d3"))))
    spec = (ss.AggregationSpecification().AddAggregation(ss.SUM, "s", "ss").AddAggregation(ss.COUNT, "", "n").AddAggregation(ss.SUM, "c", "sc")
            .AddAggregation(ss.MIN, "d", "mn").AddAggregation(ss.MAX, "d0", "mx").AddAggregation(ss.SUM, "d1", "s1").AddAggregation(ss.SUM, "p", "sp"))
    op = ss.ScalarAggregate(spec, ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(499)), ss.ProjectAllAttributes(), ss.Compute(e, ss.ScanView(view))))
    ctx = ss.Context(0)
    ctx.set_option("specialize", 1)
    out = {"rows": n, "bytes": n * 64, "runs": {}}
    row = None
    for label, chunk in (("upload_then_run", None), ("chunked_2^20", 1 << 20), ("chunked_2^22", 1 << 22), ("chunked_2^24", 1 << 24)):
        plan = ss.Plan(op, ctx)
        times = []
        for _rep in range(4):
            plan._block_key = None           # (upload_then_run: a fresh upload every repetition)
            t0 = time.perf_counter()
            if chunk is None:
                plan.run()
            else:
                plan.run_host(chunk_rows=chunk)
            got = plan.fetch()
            times.append(time.perf_counter() - t0)
        this = [got.column(i).data[0].item() for i in range(got.column_count())]
        row = row or this
        best = min(times[1:])
        out["runs"][label] = {"seconds": best, "GB_per_s": n * 64 / best / 1e9, "rows_per_s": n / best, "same_row": this == row}
    # ---- the other chunked forms (ssgpu.h "CHUNKED STAGING" 2 and 3): a materialising Filter over the same block, and BASELINE configs[2]'s
    # GroupAggregate (2 x INT32 keys, 1e5 groups, 12 DOUBLE aggregates) over a pinned host block of its own
    def timed(op, label_prefix, row_bytes, same):
        runs = {}
        first = None
        for label, chunk in (("upload_then_run", None), ("chunked_2^22", 1 << 22), ("chunked_2^24", 1 << 24)):
            plan = ss.Plan(op, ctx)
            times = []
            for _rep in range(3):
                plan._block_key = None
                t0 = time.perf_counter()
                if chunk is None:
                    plan.run()
                else:
                    plan.run_host(chunk_rows=chunk)
                ctx.synchronize()
                rows_out = plan.result_row_count()
                times.append(time.perf_counter() - t0)
            digest = same(plan)
            first = first if first is not None else digest
            best = min(times[1:])
            runs[label] = {"seconds": best, "GB_per_s": n * row_bytes / best / 1e9, "rows_per_s": n / best, "rows_out": rows_out, "same_result": digest == first}
            del plan
        return runs
    fop = ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(499)), ss.ProjectAllAttributes(), ss.ScanView(view))
    out["filter_mat"] = {"bytes": n * 64, "runs": timed(fop, "filter", 64, lambda plan: plan.result_row_count())}
    import bench
    gcols = []
    for arr in bench.host_columns(np, "group", n):
        pinned = torch.from_numpy(np.ascontiguousarray(arr)).pin_memory()
        gcols.append(pinned)
    gview = ss.View(bench.group_schema(ss), [t.numpy() for t in gcols])
    bench.GROUP_FILTER = False
    gop = bench.build_group_plan(ss, gview)

    def group_digest(plan):
        got = plan.fetch()
        order = np.lexsort((got.column(1).data, got.column(0).data))
        return (got.row_count(), float(got.column(2).data[order][:1000].sum()), float(got.column(5).data[order][-1000:].sum()))
    out["group3"] = {"bytes": n * 48, "runs": timed(gop, "group3", 48, group_digest)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
