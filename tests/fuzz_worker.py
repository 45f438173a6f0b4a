"""One worker of the shipped-configuration fuzz (tests/test_fuzz_shipped_gpu.py, tools/fuzz_hunt.py): a slice of seeds of one
plan generator of tests/fuzz_plans.py (or the plain-GroupAggregate generator of tests/test_dense_gpu.py), every plan run on
the GPU under the context options the job names and compared bit for bit with the oracle.  A process of its own, so that many
of them compile their plans' kernels side by side (hiprtc is seconds per plan and single-threaded).

    python tests/fuzz_worker.py '{"gen": "plan", "first": 0, "count": 100, "rows": 1537, "options": {"specialize": 1}}'

prints ONE JSON line: plans run / rejected by the binder, failures [(seed, message)], how many plans held a specialised kernel,
how many GroupAggregate stages ran dense, hiprtc compilations and disk-cache hits of this process."""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(job):
    import supersonic_amd as ss
    from oracle import oracle
    from helpers import run_both
    from fuzz_plans import Gen, make_view

    ctx = ss.Context(0)
    for k, v in (job.get("options") or {}).items():
        ctx.set_option(k, v)
    gen, rows = job["gen"], job.get("rows", 1537)
    import fuzz_plans
    out = {"gen": gen, "rows": rows, "first": job["first"], "count": job["count"], "options": job.get("options") or {},
           "ran": 0, "rejected": 0, "failures": []}
    stats = {}
    t0 = time.time()
    for seed in range(job["first"], job["first"] + job["count"]):
        try:
            if gen == "plain_group":
                view = make_view(rows if rows > 0 else (1537 if seed % 4 else 70001), 5000 + seed)
                op, ordered = fuzz_plans.random_plain_group(seed, view), False
            else:
                view = make_view(rows, job.get("view_seed", 1000) + seed)
                made = getattr(Gen(seed), gen)(view)
                op, ordered = made if isinstance(made, tuple) else (made, True)
            try:
                oracle.run(op)
            except oracle.OracleError:
                try:
                    ss.Plan(op, ctx)
                    out["failures"].append([seed, "the device binds a plan the oracle rejects"])
                except ss.SupersonicException:
                    out["rejected"] += 1
                continue
            run_both(op, ctx, ignore_order=not ordered, stats=stats)
            out["ran"] += 1
        except Exception:
            out["failures"].append([seed, traceback.format_exc().splitlines()[-1][:400]])
    mem = ss.memory_stats()
    out.update(stats)
    out["rtc_compilations"] = mem.get("rtc_compilations", 0)
    out["rtc_disk_hits"] = mem.get("rtc_disk_hits", 0)
    out["seconds"] = round(time.time() - t0, 1)
    return out


if __name__ == "__main__":
    print(json.dumps(main(json.loads(sys.argv[1]))))
