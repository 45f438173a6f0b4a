"""Development: does the materialising Filter's time follow WHERE its arenas were placed?  Runs the bench query in fresh processes and
prints, per process, ms per step next to the device addresses of the input block's first column and of the result's first column
(the store pass reads the one and writes the other in lockstep).  python tools/filter_placement.py [processes] > gpurun_out/filter_placement.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
import torch
import bench, supersonic_amd as ss
dev = torch.device("cuda", 0)
ctx = ss.Context(0)
ctx.set_option("specialize", 1); ctx.set_option("lazy_feedback", 1)
rows = 100_000_000
cols = bench.gen_device_columns(torch, rows, 42, dev)
schema = bench.bench_schema(ss)
blk, placed = bench.place_in_library_block(ss, torch, ctx, schema, cols, rows, dev)
del cols; torch.cuda.empty_cache()
view = ss.DeviceView(schema, [(t.data_ptr(), 0) for t in placed], rows)
plan = ss.Plan(bench.build_filter_mat_plan(ss, view), ctx)
for _ in range(10): plan.run()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(30): plan.run()
ctx.synchronize()
ms = (time.perf_counter() - t0) / 30 * 1e3
out = plan.result_device_view()
print(json.dumps({"ms": ms, "in0": placed[0].data_ptr(), "in7": placed[7].data_ptr(), "out0": out._ptrs[0][0], "out7": out._ptrs[7][0]}))
'''

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    for i in range(n):
        r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("process %d failed: %s" % (i, r.stderr[-300:]))
            continue
        d = json.loads(line[-1])
        print("process %d: %.4f ms  in0 %#x in7 %#x out0 %#x out7 %#x  (out0 - in0) mod 2^30 = %#x, mod 2^21 = %#x" %
              (i, d["ms"], d["in0"], d["in7"], d["out0"], d["out7"], (d["out0"] - d["in0"]) % (1 << 30), (d["out0"] - d["in0"]) % (1 << 21)))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
