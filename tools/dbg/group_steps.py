"""Development: per-step wall time of bench.py's --query group plan (finds one-off stalls inside a timed region)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import supersonic_amd as ss

spec = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
ctx = ss.Context(0)
ctx.set_option("specialize", spec)
ctx.set_option("profile", 1); ctx.set_option("profile_total", 0)
rows = 100_000_000
cols = bench.gen_group_columns(torch, rows, 42, dev)
view = ss.DeviceView(bench.group_schema(ss), [(t.data_ptr(), 0) for t in cols], rows)
plan = ss.Plan(bench.build_group_plan(ss, view), ctx)
ts = []
for i in range(90):
    t0 = time.perf_counter(); plan.run(view); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    c = plan.counters() if hasattr(plan, "counters") else None
print("specialize", spec, "specialized stages", plan.specialized())
print(" ".join("%.1f" % t for t in ts))
