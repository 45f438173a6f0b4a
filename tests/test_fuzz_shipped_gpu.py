"""The parity fuzz on the kernels the library SHIPS AND TIMES (round-5 review, weak #1).

tests/conftest.py pins the interpreting kernels and the hashed group shapes for contexts that do not ask (which kernel a default
plan runs must not depend on the caches an earlier test left).  This module runs the same seeded corpus -- every generator of
tests/fuzz_plans.py, same seeds, same views as tests/test_fuzz_gpu.py -- under the configurations a user gets and bench.py times:

  * `specialize = 1`: every stage runs the kernel hiprtc compiled for its program (what `specialize = 3`, the default policy,
    switches to once its background compilation is done: same source, same flags, same code object);
  * `group_dense = 1` (the default), with `dense_min_rows = 1` so that the small fuzz views take the dense-slot shapes too.

A compiled kernel costs seconds of single-threaded hiprtc time, so the corpus is cut into slices that run in worker PROCESSES
side by side (tests/fuzz_worker.py), each comparing its plans bit for bit with the oracle.  The totals -- plans run, plans that
held a specialised kernel, GroupAggregate stages that ran dense, compilations vs disk-cache hits -- go to
gpurun_out/fuzz_shipped.json (and stdout with -s)."""
import concurrent.futures
import json
import os
import subprocess
import sys

import pytest

# (the corpus runs inside a module fixture: ten minutes of hiprtc on the GPU boxes' 16 cores -- far beyond pytest.ini's per-test timeout,
#  which exists for hangs, not for this)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(3600)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALE = float(os.environ.get("SS_FUZZ_SHIPPED_SCALE", "1"))     # 0.1 for a quick look, 10 for a hunt

SPEC = {"specialize": 1}                                           # + group_dense = 1, the library's default
DENSE = {"specialize": 0, "group_dense": 1, "dense_min_rows": 1}
BOTH = {"specialize": 1, "group_dense": 1, "dense_min_rows": 1}
# (leg, generator, first seed, seeds, rows, view-seed offset, options) -- seeds and views of tests/test_fuzz_gpu.py
LEGS = [
    ("specialized/general", "plan", 0, 1300, 1537, 1000, SPEC),      # (2000 seeds in the full corpus below: 7400 worker-seconds of hiprtc)
    ("specialized/many-tiles", "plan", 2000, 200, 70001, 0, SPEC),
    ("specialized/ordered", "ordered_aggregate_plan", 4000, 250, 1537, 0, SPEC),
    ("dense/plain-groups", "plain_group", 0, 2000, 0, 0, DENSE),
    ("dense/general", "plan", 0, 2000, 1537, 1000, DENSE),
    ("dense+specialized/plain-groups", "plain_group", 2000, 400, 0, 0, BOTH),
    ("dense+specialized/many-tiles", "plan", 2000, 200, 70001, 0, BOTH),
]


# the multi-stage shapes (three to four compiled kernels per plan): part of the full corpus (SS_FUZZ_SHIPPED_FULL=1; the round's full
# run is profiles/r06_fuzz_shipped_full.json: 20 minutes with 48 workers), not of the default suite, which has to fit the driver's
# time limit next to the other 5000 tests
FULL_LEGS = [
    ("specialized/general-rest", "plan", 1300, 700, 1537, 1000, SPEC),
    ("specialized/key-limit", "distinct_limit_plan", 6000, 200, 1537, 0, SPEC),
    ("specialized/row-after-row", "sequential_sum_plan", 5000, 150, 1537, 0, SPEC),
]
if os.environ.get("SS_FUZZ_SHIPPED_FULL") == "1":
    LEGS = LEGS + FULL_LEGS


def jobs():
    out = []
    for leg, gen, first, count, rows, view_seed, options in LEGS:
        count = max(10, int(count * SCALE))
        step = 25 if options.get("specialize") else 100
        for lo in range(first, first + count, step):
            out.append((leg, {"gen": gen, "first": lo, "count": min(step, first + count - lo), "rows": rows, "view_seed": view_seed,
                              "options": options}))
    return out


def cpu_quota():
    """CPUs' worth of time the cgroup schedules (the GPU boxes report 256 CPUs and run about 16): more workers than that only
    take turns (measured: 12 and 96 workers finish the same corpus in the same time)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        return None if quota == "max" else max(1, int(round(int(quota) / float(period))))
    except (OSError, ValueError):
        return None


def run_job(job):
    env = dict(os.environ)
    env.pop("SSGPU_SPECIALIZE", None)
    env.pop("SSGPU_GROUP_DENSE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_worker.py"), json.dumps(job)], env=env, capture_output=True,
                       text=True, timeout=1500)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"ran": 0, "failures": [[job["first"], "worker died (rc %d): %s" % (p.returncode, (p.stderr or p.stdout)[-600:])]]}
    return json.loads(lines[-1])


@pytest.fixture(scope="module")
def corpus():
    work = jobs()
    workers = int(os.environ.get("SS_FUZZ_WORKERS", "0")) or max(4, min(48, cpu_quota() or (os.cpu_count() or 8) // 4))
    with concurrent.futures.ThreadPoolExecutor(workers) as pool:
        results = list(pool.map(run_job, [j for _leg, j in work]))
    legs = {}
    for (leg, _job), r in zip(work, results):
        t = legs.setdefault(leg, {"ran": 0, "rejected": 0, "plans": 0, "specialized_plans": 0, "group_stages": 0, "dense_stages": 0,
                                  "rtc_compilations": 0, "rtc_disk_hits": 0, "seconds": 0.0, "failures": []})
        for k in t:
            if k == "failures":
                t[k] += r.get(k, [])
            else:
                t[k] += r.get(k, 0)
    summary = {"workers": workers, "scale": SCALE, "legs": legs}
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "fuzz_shipped.json"), "w") as f:
            json.dump(summary, f, indent=1)
    except OSError:
        pass
    print(json.dumps(summary))
    return legs


@pytest.mark.parametrize("leg", sorted({leg[0] for leg in LEGS}))
def test_fuzz_corpus_on_the_shipped_kernels(corpus, leg):
    t = corpus[leg]
    assert not t["failures"], "%d plans differ from the oracle, first: %s" % (len(t["failures"]), t["failures"][:5])
    assert t["ran"] > 0
    if leg.startswith("specialized") or leg.startswith("dense+specialized"):
        # the point of the leg: the compiled kernels ran (a plan whose every stage was refused a kernel would be an interpreted run)
        assert t["specialized_plans"] >= 0.95 * t["plans"], t
    if leg.endswith("plain-groups"):
        assert t["dense_stages"] >= 0.9 * t["group_stages"] > 0, t


def test_the_corpus_is_as_large_as_the_review_asked(corpus):
    if SCALE < 1:
        pytest.skip("scaled-down run")
    spec = sum(t["specialized_plans"] for leg, t in corpus.items() if "specialized" in leg)
    dense = sum(t["dense_stages"] for leg, t in corpus.items() if leg.startswith("dense"))
    assert spec >= 2000, spec
    assert dense >= 2000, dense
