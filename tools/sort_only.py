"""Development aid: config #5's Sort a few times, product only (no result check: used with kernel parts switched off)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import supersonic_amd as ss
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
device = torch.device("cuda", 0)
cols = bench.gen_device_columns(torch, rows, 42, device)
torch.cuda.synchronize()
ctx = ss.Context(0)
view = ss.DeviceView(bench.bench_schema(ss), [(t.data_ptr(), 0) for t in cols], rows)
plan = ss.Plan(bench.build_sort_plan(ss, view), ctx)
for _ in range(6):
    plan.run(view)
ctx.synchronize()
print("done", plan.stage_info())
