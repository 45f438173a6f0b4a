"""Development: the filtered group query of bench.py at full size, every run's stage shape and overflow flags."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import supersonic_amd as ss
import bench

bench.GROUP_FILTER = os.environ.get("NOFILTER") is None
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
cols = bench.gen_group_columns(torch, rows, 42, dev)
torch.cuda.synchronize()
view = ss.DeviceView(bench.group_schema(ss), [(t.data_ptr(), 0) for t in cols], rows)
ctx = ss.Context(0)
ctx.set_option("debug_timing", 1)
for kv in sys.argv[2:]:
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
plan = ss.Plan(bench.build_group_plan(ss, view), ctx)
for i in range(int(os.environ.get("RUNS", "5"))):
    plan.run(view)
    ctx.synchronize()
    c = plan.counters()
    print(i, "dom %.3f ms" % c.dominant_ms, plan.stage_info()[0], flush=True)
print("groups", plan.fetch().row_count())
