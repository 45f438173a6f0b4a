"""BASELINE config #3's query (GroupAggregate, 2 x INT32 keys, 12 DOUBLE aggregates, 100 M rows) over heavily skewed keys: 30 % of the
rows in ONE group, 20 % more in 16 others, the rest uniform over 1e5 groups.  Prints one JSON line: per-run wall time while the
plan adapts (the first runs: segment overflow -> heavy-hitter sample -> rerun), then the steady state's kernel time.
Usage (GPU box): python tools/skew_bench.py [rows] [specialize 0|1]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import supersonic_amd as ss

ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
SPEC = int(sys.argv[2]) if len(sys.argv) > 2 else 1
device = torch.device("cuda", 0)
a, k1, k2, d0, d1, d2, d3 = bench.gen_group_columns(torch, ROWS, 77, device)
g = torch.Generator(device=device); g.manual_seed(5)
u = torch.rand(ROWS, generator=g, device=device)
grp = k1.to(torch.int64) * 317 + k2
grp = torch.where(u < 0.3, torch.full_like(grp, 7), torch.where(u < 0.5, grp % 16, grp))
k1, k2 = (grp // 317).to(torch.int32), (grp % 317).to(torch.int32)
del u, grp
torch.cuda.synchronize()
ctx = ss.Context(0)
ctx.set_option("specialize", SPEC)
view = ss.DeviceView(bench.group_schema(ss), [(t.data_ptr(), 0) for t in (a, k1, k2, d0, d1, d2, d3)], ROWS)
plan = ss.Plan(ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), bench.group_spec(ss), None, ss.ScanView(view)), ctx)
first = []
for i in range(5):
    t0 = time.perf_counter()
    plan.run(view)
    ctx.synchronize()
    info = [s for s in plan.stage_info() if s["kind"] == 3][-1]
    first.append({"wall_ms": (time.perf_counter() - t0) * 1e3, "shape": info["group_shape"], "hot_keys": info["hot_keys"], "reruns": info["reruns"], "seg_growth": info["part_seg_growth"]})
for _ in range(20):
    plan.run(view)
ctx.synchronize()
ms = plan.recent_kernel_ms(20)
rows_out = plan.result_device_view().row_count()
print(json.dumps({"workload": "config #3's GroupAggregate over skewed keys: 30 %% of %d rows in one group, 20 %% in 16 others" % ROWS, "specialize": SPEC,
                  "adapting_runs": first, "steady_kernel_ms": sum(ms) / len(ms), "steady_kernel_ms_min": min(ms), "groups": rows_out}))
