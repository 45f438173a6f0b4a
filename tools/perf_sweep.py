#!/usr/bin/env python3
"""Development aid: time the pipeline kernel for several plan shapes and tile/grid settings
on device-resident synthetic blocks (prints one line per configuration)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import supersonic_amd as ss  # noqa: E402
import bench  # noqa: E402

NA = ss.NamedAttribute


def q_wide(v):
    return bench.build_plan(ss, v)


def q_narrow(v):
    return ss.ScalarAggregate(
        ss.AggregationSpecification().AddAggregation(ss.SUM, "s", "sum_s").AddAggregation(ss.COUNT, "a", "cnt"),
        ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(499)), ss.ProjectAllAttributes(),
                  ss.Compute(ss.CompoundExpression().Add(NA("a")).AddAs("s", ss.Plus(NA("a"), NA("b"))), ss.ScanView(v))))


def q_stage8(v):
    # all 8 columns staged, little work: 2 aggregates over sums of columns
    e = (ss.CompoundExpression().AddAs("i", ss.Plus(ss.Plus(NA("a"), NA("b")), ss.Plus(NA("c"), NA("d"))))
         .AddAs("f", ss.Plus(ss.Plus(NA("d0"), NA("d1")), ss.Plus(NA("d2"), NA("d3")))))
    spec = ss.AggregationSpecification().AddAggregation(ss.SUM, "i", "si").AddAggregation(ss.MAX, "f", "mf")
    return ss.ScalarAggregate(spec, ss.Compute(e, ss.ScanView(v)))


def q_min8(v):
    spec = ss.AggregationSpecification()
    for c in ["a", "b", "c", "d", "d0", "d1", "d2", "d3"]:
        spec.AddAggregation(ss.MIN, c, "m" + c)
    return ss.ScalarAggregate(spec, ss.ScanView(v))


def q_fused4(v):
    # all 8 columns staged by FOUR instructions (fused binary-operator sinks): what the interpreter costs per instruction
    e = (ss.CompoundExpression().AddAs("s1", ss.Plus(NA("a"), NA("b"))).AddAs("s2", ss.Plus(NA("c"), NA("d")))
         .AddAs("p1", ss.Multiply(NA("d0"), NA("d1"))).AddAs("p2", ss.Multiply(NA("d2"), NA("d3"))))
    spec = (ss.AggregationSpecification().AddAggregation(ss.SUM, "s1", "a1").AddAggregation(ss.SUM, "s2", "a2")
            .AddAggregation(ss.SUM, "p1", "a3").AddAggregation(ss.SUM, "p2", "a4"))
    return ss.ScalarAggregate(spec, ss.Compute(e, ss.ScanView(v)))


def q_sum1(v):
    return ss.ScalarAggregate(ss.AggregationSpecification().AddAggregation(ss.SUM, "a", "sa"), ss.ScanView(v))


def q_filter_mat(v):
    return ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(499)), ss.ProjectAllAttributes(), ss.ScanView(v))


def q_group(v):
    spec = ss.AggregationSpecification()
    for c in ["d0", "d1", "d2", "d3"]:
        spec.AddAggregation(ss.SUM, c, "s" + c).AddAggregation(ss.MIN, c, "n" + c).AddAggregation(ss.MAX, c, "x" + c)
    return ss.GroupAggregate(ss.ProjectNamedAttributes(["c"]), spec, None, ss.ScanView(v))


def q_group_small(v):
    # 1000 groups (key a), SUM/MIN/MAX over 4 DOUBLE columns
    spec = ss.AggregationSpecification()
    for c in ["d0", "d1", "d2", "d3"]:
        spec.AddAggregation(ss.SUM, c, "s" + c).AddAggregation(ss.MIN, c, "n" + c).AddAggregation(ss.MAX, c, "x" + c)
    return ss.GroupAggregate(ss.ProjectNamedAttributes(["a"]), spec, None, ss.ScanView(v))


def q_group_tiny(v):
    # 8 groups, 3 aggregates (TPC-H Q1 shape)
    e = ss.CompoundExpression().AddAs("k", ss.ModulusSignaling(NA("a"), ss.ConstInt64(8))).Add(NA("d0")).Add(NA("d1"))
    spec = (ss.AggregationSpecification().AddAggregation(ss.SUM, "d0", "s0").AddAggregation(ss.SUM, "d1", "s1")
            .AddAggregation(ss.COUNT, "", "n"))
    return ss.GroupAggregate(ss.ProjectNamedAttributes(["k"]), spec, None, ss.Compute(e, ss.ScanView(v)))


_DIM = {}


def q_join(v):
    # star-join shape: 100 M-row fact table (key c, 1e5 distinct) INNER JOIN a 1e5-row dimension table, then aggregate
    if "view" not in _DIM:
        dev = torch.device("cuda", 0)
        m = 100000
        ids = torch.randperm(m, device=dev, dtype=torch.int64)
        w = torch.arange(m, device=dev, dtype=torch.float64) * 0.5
        g = (torch.arange(m, device=dev, dtype=torch.int64) % 7).to(torch.int32)
        _DIM["cols"] = [ids, w, g]
        schema = ss.TupleSchema([ss.Attribute("id", ss.INT64), ss.Attribute("w", ss.DOUBLE), ss.Attribute("g", ss.INT32)])
        _DIM["view"] = ss.DeviceView(schema, [(t.data_ptr(), 0) for t in _DIM["cols"]], m)
    proj = ss.CompoundMultiSourceProjector().add(0, ss.ProjectNamedAttributes(["a", "d0"])).add(1, ss.ProjectNamedAttributes(["w", "g"]))
    join = ss.HashJoin(ss.INNER, ss.ProjectNamedAttribute("c"), ss.ProjectNamedAttribute("id"), proj, ss.UNIQUE, ss.ScanView(v), ss.ScanView(_DIM["view"]))
    spec = ss.AggregationSpecification().AddAggregation(ss.SUM, "w", "sw").AddAggregation(ss.SUM, "d0", "sd").AddAggregation(ss.COUNT, "", "n")
    return ss.ScalarAggregate(spec, ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(499)), ss.ProjectAllAttributes(), join))


def q_addn(n):
    def q(v):
        e = NA("a")
        for i in range(n):
            e = ss.Plus(e, ss.ConstInt64(i + 1))
        return ss.ScalarAggregate(ss.AggregationSpecification().AddAggregation(ss.SUM, "x", "sx"),
                                  ss.Compute(ss.CompoundExpression().AddAs("x", e), ss.ScanView(v)))
    return q


def q_sumn(n):
    def q(v):
        spec = ss.AggregationSpecification()
        for i in range(n):
            spec.AddAggregation(ss.SUM, "a", "s%d" % i)
        return ss.ScalarAggregate(spec, ss.ScanView(v))
    return q


def q_sort(v):
    return ss.Sort(ss.SortOrder().add("d", ss.ASCENDING), ss.ProjectAllAttributes(), 0, ss.ScanView(v))


def q_sort_keyonly(v):
    return ss.Sort(ss.SortOrder().add("d", ss.ASCENDING), ss.ProjectNamedAttributes(["d"]), 0, ss.ScanView(v))


def q_group2(v):
    # BASELINE config #3 shape: 2 x INT32 keys (1e5 distinct pairs) + SUM/MIN/MAX over 4 DOUBLE columns
    e = (ss.CompoundExpression().AddAs("k1", ss.CastTo(ss.INT32, ss.CppDivideSignaling(NA("c"), ss.ConstInt64(317))))
         .AddAs("k2", ss.CastTo(ss.INT32, ss.ModulusSignaling(NA("c"), ss.ConstInt64(317)))).Add(NA("d0")).Add(NA("d1")).Add(NA("d2")).Add(NA("d3")))
    spec = ss.AggregationSpecification()
    for c in ["d0", "d1", "d2", "d3"]:
        spec.AddAggregation(ss.SUM, c, "s" + c).AddAggregation(ss.MIN, c, "n" + c).AddAggregation(ss.MAX, c, "x" + c)
    return ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), spec, None, ss.Compute(e, ss.ScanView(v)))


QUERIES = {"sort": q_sort, "sort_key": q_sort_keyonly, "group2": q_group2, "add4": q_addn(4), "add16": q_addn(16), "sum4": q_sumn(4), "sum8": q_sumn(8),"wide": q_wide, "narrow": q_narrow, "stage8": q_stage8, "min8": q_min8, "sum1": q_sum1,
           "fused4": q_fused4, "filter_mat": q_filter_mat, "group": q_group, "group_small": q_group_small, "group_tiny": q_group_tiny, "join": q_join}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--queries", default="wide,narrow,stage8,min8,sum1")
    ap.add_argument("--tiles", default="0,512,1024,2048")
    ap.add_argument("--lds", default="49152")
    ap.add_argument("--grids", default="0")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--debug", type=int, default=0)
    ap.add_argument("--wgs", type=int, default=0)
    ap.add_argument("--opts", default="", help="extra context options: key=value,key=value")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cols = bench.gen_device_columns(torch, args.rows, 42, dev)
    torch.cuda.synchronize()
    view = ss.DeviceView(bench.bench_schema(ss), [(t.data_ptr(), 0) for t in cols], args.rows)
    for qn in args.queries.split(","):
        for tile in [int(x) for x in args.tiles.split(",")]:
            for lds in [int(x) for x in args.lds.split(",")]:
                for grid in [int(x) for x in args.grids.split(",")]:
                    ctx = ss.Context(0)
                    ctx.set_option("tile_rows", tile)
                    ctx.set_option("lds_target_bytes", lds)
                    ctx.set_option("grid_limit", grid)
                    ctx.set_option("debug_timing", args.debug)
                    ctx.set_option("wgs_per_cu", args.wgs)
                    for kv in [x for x in args.opts.split(",") if x]:
                        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
                    try:
                        plan = ss.Plan(QUERIES[qn](view), ctx)
                        ms = []
                        for _ in range(args.reps):
                            plan.run(view)
                            ctx.synchronize()
                            c = plan.counters()
                            ms.append((c.dominant_ms, c.kernel_ms))
                        ms.sort()
                        dom, tot = ms[len(ms) // 2]
                        gbs = c.algorithmic_bytes / (dom / 1e3) / 1e9
                        if qn.startswith("sort") or qn in ("filter_mat", "group", "group2", "group_small", "group_tiny"):
                            print("   [%s] total kernel time %.3f ms -> %.2f Grows/s, launches %d" % (qn, tot, args.rows / tot / 1e6, c.n_launches))
                        print("%-10s tile=%-5d lds_target=%-6d grid=%-5d lds=%-6d dom=%.3f ms total=%.3f ms  %.0f GB/s (%.1f%% of 8TB/s)  %.1f Grows/s" % (
                            qn, c.tile_rows, lds, c.grid, c.lds_bytes, dom, tot, gbs, gbs / 80.0, args.rows / dom / 1e6), flush=True)
                    except ss.SupersonicException as e:
                        print("%-10s tile=%d: %s" % (qn, tile, e), flush=True)


if __name__ == "__main__":
    main()
