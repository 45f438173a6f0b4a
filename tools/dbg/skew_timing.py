"""Development aid: the 100 M-row GroupAggregate over heavily skewed keys (30 % of the rows in one group), product only, timed per run."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
import supersonic_amd as ss

ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
device = torch.device("cuda", 0)
a, k1, k2, d0, d1, d2, d3 = bench.gen_group_columns(torch, ROWS, 77, device)
g = torch.Generator(device=device); g.manual_seed(5)
u = torch.rand(ROWS, generator=g, device=device)
grp = k1.to(torch.int64) * 317 + k2
grp = torch.where(u < 0.3, torch.full_like(grp, 7), torch.where(u < 0.5, grp % 16, grp))
k1, k2 = (grp // 317).to(torch.int32), (grp % 317).to(torch.int32)
torch.cuda.synchronize()
ctx = ss.Context(0)
ctx.set_option("debug_timing", 1)
view = ss.DeviceView(bench.group_schema(ss), [(t.data_ptr(), 0) for t in (a, k1, k2, d0, d1, d2, d3)], ROWS)
plan = ss.Plan(ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), bench.group_spec(ss), None, ss.ScanView(view)), ctx)
for i in range(4):
    t0 = time.time()
    plan.run(view)
    ctx.synchronize()
    print("run %d: %.1f ms, shape %s, rows %d" % (i, (time.time() - t0) * 1e3, [s["group_shape"] for s in plan.stage_info()], plan.result_device_view().row_count()), flush=True)
