"""What a cursor that is drained ONCE pays: the first run of config #3's GroupAggregate (2 x INT32 keys, 12 DOUBLE aggregates) at sizes
below and above the scout's threshold, against the plan's steady state.  Usage: python tools/first_run_bench.py [groups]"""
import gc, os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import supersonic_amd as ss

groups = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda", 0)
out = []
for rows in (1 << 18, 1 << 20, 1 << 21, 1 << 22, 6_000_000, 1 << 23, 1 << 24):
    cols = bench.gen_group_columns(torch, rows, 77, dev)
    if groups != bench.N_GROUPS:
        g = (cols[1].to(torch.int64) * 317 + cols[2]) % groups
        cols = (cols[0], (g // 317).to(torch.int32), (g % 317).to(torch.int32)) + tuple(cols[3:])
    view = ss.DeviceView(bench.group_schema(ss), [(t.data_ptr(), 0) for t in cols], rows)
    op = ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), bench.group_spec(ss), None, ss.ScanView(view))
    torch.cuda.synchronize()
    ctx = ss.Context(0)
    ctx.set_option("specialize", 0)
    warm = ss.Plan(op, ctx); warm.run(view); ctx.synchronize()          # (module load; an earlier query of the same service)
    del warm; gc.collect()                                              # ... whose cursor is gone: its device blocks wait in the library's pool
    plan = ss.Plan(op, ctx)
    times = []
    for _ in range(6):
        t = time.perf_counter(); plan.run(view); ctx.synchronize(); times.append((time.perf_counter() - t) * 1e3)
    info = [st for st in plan.stage_info() if st["kind"] == 3][-1]
    out.append({"rows": rows, "first_run_ms": round(times[0], 3), "second_ms": round(times[1], 3), "steady_ms": round(min(times[2:]), 3), "steady_shape": info["group_shape"], "dense_slots": info["dense_slots"]})
    print(out[-1], flush=True)
print(json.dumps({"groups": groups, "runs": out}))
