"""A longer fuzz hunt than the suite's: seeded random plans of one generator (tests/fuzz_plans.py) on the GPU against the oracle.
Usage: python tools/fuzz_hunt.py <plan|ordered_aggregate_plan|sequential_sum_plan|sort_plan|join_plan> <first seed> <count> [rows]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("SSGPU_SPECIALIZE", "0")
import supersonic_amd as ss
from oracle import oracle
from helpers import run_both
from fuzz_plans import Gen, make_view

gen, first, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 1537
ctx = ss.Context(0)
bad = bound = 0
for seed in range(first, first + count):
    view = make_view(rows, 1000 + seed)
    g = Gen(seed)
    out = getattr(g, gen)(view)
    op, ordered = out if isinstance(out, tuple) else (out, True)
    try:
        oracle.run(op)
    except oracle.OracleError:
        try:
            ss.Plan(op, ctx)
            bad += 1; print("seed", seed, ": the device binds a plan the oracle rejects")
        except ss.SupersonicException:
            pass
        continue
    bound += 1
    try:
        run_both(op, ctx, ignore_order=not ordered)
    except Exception:
        bad += 1
        print("seed", seed, "FAILED:", traceback.format_exc().splitlines()[-1][:300])
print("%s seeds %d..%d rows %d: %d plans ran, %d failures" % (gen, first, first + count - 1, rows, bound, bad))
