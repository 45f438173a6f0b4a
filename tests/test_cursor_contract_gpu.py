"""Cursor-contract tests modelled on the reference's OperationTest fixture
(supersonic/testing/operation_testing.cc:350-352,452-520): every operator is pulled with the output
block sizes {1, 2, 5, 20, 10001, "everything"}, Interrupt() is best-effort and leaves the plan
usable (cursor.h:150-186), and repeated runs of a plan do not grow device memory
(expression_test_helper.cc:213-245, the "memory stability" run)."""
import os
import threading

import numpy as np
import pytest

import supersonic_amd as ss
from helpers import run_both
from test_parity_gpu import make_view, fpa_narrow, group_query

pytestmark = pytest.mark.gpu
NA = ss.NamedAttribute


def operators(view):
    flt = ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(499)), ss.ProjectNamedAttributes(["a", "d0", "k1"]), ss.ScanView(view))
    return {
        "scalar": fpa_narrow(view),
        "compute": ss.Compute(ss.CompoundExpression().AddAs("s", ss.Plus(NA("a"), NA("b"))).Add(NA("t")), ss.ScanView(view)),
        "filter": flt,
        "group": group_query(view, True, ("k2",)),
        "sort": ss.Sort(ss.SortOrder().add("k1", ss.DESCENDING).add("a", ss.ASCENDING), None, 0, flt),
        "group_wide": group_query(view, False, ("k1", "k2")),       # > 64 key bits: materialise + sort + clustered aggregation
        "group_first_last": ss.GroupAggregate(ss.ProjectNamedAttributes(["k2"]), ss.AggregationSpecification()
                                              .AddAggregation(ss.FIRST, "d0", "f").AddAggregation(ss.LAST, "a", "l").AddAggregation(ss.COUNT, "", "n"), None, flt_all(view)),
        "join_not_unique": ss.HashJoin(ss.LEFT_OUTER, ss.ProjectNamedAttribute("k2"), ss.ProjectNamedAttribute("id"),
                                       ss.CompoundMultiSourceProjector().add(0, ss.ProjectNamedAttributes(["a", "k2"])).add(1, ss.ProjectNamedAttributes(["w"])),
                                       ss.NOT_UNIQUE, flt_all(view), ss.ScanView(DIM)),
    }


DIM = ss.View(ss.TupleSchema([ss.Attribute("id", ss.INT32), ss.Attribute("w", ss.INT64, ss.NULLABLE)]),
              [np.array([0, 3, 3, 7, 30, 3, 7], dtype=np.int32), ss.Column(np.arange(7) * 11, np.arange(7) == 2)])


def flt_all(view):
    return ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(499)), ss.ProjectAllAttributes(), ss.ScanView(view))


@pytest.mark.parametrize("max_rows", [1, 2, 5, 20, 10001, 1 << 40])
@pytest.mark.parametrize("name", ["scalar", "compute", "filter", "group", "sort", "group_wide", "group_first_last", "join_not_unique"])
def test_output_block_sizes(gpu_ctx, name, max_rows):
    n = 137 if max_rows < 20 else 30011
    op = operators(make_view(n, nullable=True))[name]
    cur = op.CreateCursor(gpu_ctx)
    seen = 0
    while True:
        r = cur.Next(max_rows)
        assert not r.is_failure(), r.exception()
        if r.is_eos():
            break
        assert 1 <= r.view().row_count() <= max_rows          # cursor.h:131-148
        seen += r.view().row_count()
    assert cur.Next(max_rows).is_eos()                          # EOS is sticky
    got = run_both(op, gpu_ctx, ignore_order=name.startswith("group"), max_rows=max_rows)
    assert got.row_count() == seen


@pytest.mark.parametrize("name", ["scalar", "filter", "group", "sort"])
def test_next_minus_one_returns_everything_that_is_left(gpu_ctx, name):
    """rowcount_t is unsigned in the reference (types.h:252-256): the guide's `cursor->Next(-1)` (primer.cc:321,
    group_sort.cc:223) asks for as many rows as the cursor has -- never a view of -1 rows, never a cursor walking backwards."""
    op = operators(make_view(30011, nullable=True))[name]
    want = run_both(op, gpu_ctx, ignore_order=name.startswith("group"))
    cur = op.CreateCursor(gpu_ctx)
    first = cur.Next(7) if want.row_count() > 7 else None          # part of the result first, then "the rest"
    r = cur.Next(-1)
    assert not r.is_failure(), r.exception()
    taken = first.view().row_count() if first is not None else 0
    assert r.view().row_count() == want.row_count() - taken and r.view().row_count() >= 1
    assert cur.Next(-1).is_eos()
    zero = op.CreateCursor(gpu_ctx).Next(0)                          # "between one and max_row_count rows"
    assert not zero.is_failure() and zero.view().row_count() == 1


def test_interrupt_before_next_and_reuse(gpu_ctx):
    op = operators(make_view(30011))["sort"]
    cur = op.CreateCursor(gpu_ctx)
    cur.Interrupt()
    r = cur.Next(1024)
    assert r.is_failure() and r.exception().return_code == ss.INTERRUPTED
    run_both(op, gpu_ctx)                                       # a fresh cursor over the same operation is unaffected


def test_interrupt_from_another_thread_is_best_effort(gpu_ctx):
    # multi-stage plan (count pass, compaction, radix sort passes, gathers): the flag is checked between stages
    view = make_view(2000003)
    op = ss.Sort(ss.SortOrder().add("d", ss.ASCENDING), None, 0,
                 ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(99)), ss.ProjectAllAttributes(), ss.ScanView(view)))
    outcomes = set()
    for delay in (0.0, 0.0005, 0.002, 0.01):
        cur = op.CreateCursor(gpu_ctx)
        timer = threading.Timer(delay, cur.Interrupt)
        timer.start()
        r = cur.Next(1 << 30)
        timer.join()
        if r.is_failure():
            assert r.exception().return_code == ss.INTERRUPTED
            outcomes.add("interrupted")
        else:
            assert r.view().row_count() > 0
            outcomes.add("completed")
    assert outcomes                                             # either outcome is legal; no crash, no other error
    cur = op.CreateCursor(gpu_ctx)
    r = cur.Next(1 << 30)
    assert not r.is_failure() and r.view().row_count() == int((view.column(0).data > 99).sum())


def _held():
    st = ss.memory_stats()
    return {k: st[k] for k in ("device_bytes", "pinned_bytes", "live_plans", "live_blocks", "rtc_modules", "rtc_code_bytes")}


@pytest.mark.parametrize("specialize", [False, True])
def test_repeated_runs_do_not_grow_device_memory(gpu_ctx, specialize):
    """The reference's "memory stability" shape (expression_test_helper.cc:213-245): 3 warm-up evaluations, then 7 more
    must not change what the allocator holds.  The measure is the library's own per-process accounting
    (ssgpu_memory_stats: every device / pinned buffer, loaded kernel modules), so the test means the same thing under
    xdist; free device memory (shared by every process on the GPU) is checked for gross leaks only."""
    import os
    import torch
    view = make_view(200003, nullable=True)
    for name, op in operators(view).items():
        plan = ss.Plan(op, gpu_ctx)
        if specialize:
            plan.specialize()                                   # the explicit compile point: never inside a steady-state run
        for _ in range(3):                                      # warm-up: buffers are sized on the first runs
            plan.run(); plan.fetch()
        gpu_ctx.synchronize()
        held_before = _held()
        compiled_before = ss.memory_stats()["rtc_compilations"]
        free_before, _total = torch.cuda.mem_get_info(0)
        for _ in range(7):
            plan.run(); plan.fetch()
        gpu_ctx.synchronize()
        free_after, _total = torch.cuda.mem_get_info(0)
        assert _held() == held_before, (name, held_before, _held())
        assert ss.memory_stats()["rtc_compilations"] == compiled_before, name     # no compiler inside a steady-state run
        if not os.environ.get("PYTEST_XDIST_WORKER"):
            assert free_before - free_after < (64 << 20), (name, free_before, free_after)
        del plan


def test_default_policy_never_compiles_and_modules_are_unloaded(gpu_ctx):
    """A plan that did not ask for specialised kernels never meets the compiler, however often it runs; one that asked
    holds references to cached modules, and the last plan to drop a kernel unloads its code object."""
    import gc
    view = make_view(50021)
    before = ss.memory_stats()
    plan = ss.Plan(fpa_narrow(view), gpu_ctx)
    for _ in range(20):
        plan.run()
    plan.fetch()
    assert plan.specialized() == 0 and ss.memory_stats()["rtc_compilations"] == before["rtc_compilations"]
    p1 = ss.Plan(fpa_narrow(view), gpu_ctx).specialize()
    assert p1.specialized() == 1, p1.specialize_reason()
    loaded = ss.memory_stats()
    assert loaded["rtc_modules"] == before["rtc_modules"] + 1 and loaded["rtc_code_bytes"] > before["rtc_code_bytes"]
    p2 = ss.Plan(fpa_narrow(make_view(777)), gpu_ctx).specialize()     # same program: same module, no second compilation
    assert ss.memory_stats()["rtc_modules"] == loaded["rtc_modules"]
    assert ss.memory_stats()["rtc_compilations"] == loaded["rtc_compilations"]
    p1.run(); p2.run()
    a, b = p1.fetch(), p2.fetch()
    assert a.row_count() == 1 and b.row_count() == 1
    del p1, a
    gc.collect()
    assert ss.memory_stats()["rtc_modules"] == loaded["rtc_modules"]   # p2 still holds it
    del p2, b
    gc.collect()
    # without a user the kernel stays loaded among the few most recently released ones (a cursor per query does not recompile) ...
    assert ss.memory_stats()["rtc_modules"] <= before["rtc_modules"] + 8
    p3 = ss.Plan(fpa_narrow(make_view(99)), gpu_ctx).specialize()
    assert ss.memory_stats()["rtc_compilations"] == loaded["rtc_compilations"]       # found, not compiled again
    del p3
    gc.collect()
    ss.specialized_kernels_trim(0)                                                    # ... and is unloaded on request, or when newer ones push it out
    after = ss.memory_stats()
    assert after["rtc_modules"] <= before["rtc_modules"] and after["rtc_code_bytes"] <= before["rtc_code_bytes"]
    del plan


def test_background_compilation_never_blocks_a_run_and_arrives_later():
    """Context option specialize = 3 (the library's default; the suite pins 0 through SSGPU_SPECIALIZE): a kernel that is in no cache is
    compiled by the library's worker thread when a run over >= specialize_min_rows rows wants it.  The run that asked -- and the ones
    after it -- use the interpreting kernel and do not wait; the plan picks the kernel up when it is there; results are the same
    before and after; smaller runs never start a compilation."""
    import time
    from helpers import to_cols, assert_cols_equal
    view = make_view(30011)
    fresh = int(time.time() * 1000) % 1000003 + 2000003      # a constant no earlier run has compiled: the program is new to every cache
    op = ss.ScalarAggregate(ss.AggregationSpecification().AddAggregation(ss.SUM, "b", "sb").AddAggregation(ss.MAX, "c", "mc").AddAggregation(ss.COUNT, "", "n"),
                            ss.Filter(ss.Less(ss.NamedAttribute("a"), ss.ConstInt64(fresh)), ss.ProjectAllAttributes(), ss.ScanView(view)))
    ctx = ss.Context(0)
    ctx.set_option("specialize", 3)
    before = ss.memory_stats()["rtc_compilations"]
    small = ss.Plan(op, ctx)                       # 30011 rows < the default threshold of 2^22: nothing is compiled for it
    small.run()
    want = to_cols(small.fetch())
    assert small.specialized() == 0 and "not in the kernel cache" in small.specialize_reason()
    ctx.set_option("specialize_min_rows", 1000)
    plan = ss.Plan(op, ctx)
    t0 = time.time()
    plan.run()
    first_run_s = time.time() - t0
    assert plan.specialized() == 0 and "being compiled in the background" in plan.specialize_reason(), plan.specialize_reason()
    assert first_run_s < 1.0, first_run_s          # (a compilation takes seconds: the run did not wait for one)
    assert_cols_equal(to_cols(plan.fetch()), want)
    interpreted_runs = 0
    deadline = time.time() + 180
    while plan.specialized() == 0 and time.time() < deadline:
        plan.run()                                 # interpreted meanwhile, same answer
        interpreted_runs += 1
        assert_cols_equal(to_cols(plan.fetch()), want)
        time.sleep(0.2)
    assert plan.specialized() == 1 and plan.specialize_reason() == "", (plan.specialize_reason(), interpreted_runs)
    assert interpreted_runs >= 1 and ss.memory_stats()["rtc_compilations"] == before + 1
    plan.run()
    assert_cols_equal(to_cols(plan.fetch()), want)
    again = ss.Plan(op, ctx)                       # the next plan with this program: found at once
    again.run()
    assert again.specialized() == 1 and ss.memory_stats()["rtc_compilations"] == before + 1
    small.specialize()                             # the plan that missed it below the threshold asks: found, no second compilation
    small.run()
    assert small.specialized() == 1 and ss.memory_stats()["rtc_compilations"] == before + 1
    assert_cols_equal(to_cols(small.fetch()), want)


def test_default_policy_uses_compiled_kernels_where_they_exist_and_never_compiles():
    """Context option specialize = 2: a plan finds a kernel
    another plan compiled, never compiles itself, says why a stage stayed with the interpreting kernel, and compiles once asked."""
    import time
    from helpers import to_cols, assert_cols_equal
    view = make_view(30011)
    fresh = int(time.time() * 1000) % 1000003 + 1000        # a constant no earlier run has compiled: the program is new to every cache
    op = ss.ScalarAggregate(ss.AggregationSpecification().AddAggregation(ss.SUM, "b", "sb").AddAggregation(ss.MIN, "c", "mc").AddAggregation(ss.COUNT, "", "n"),
                            ss.Filter(ss.Less(ss.NamedAttribute("a"), ss.ConstInt64(fresh)), ss.ProjectAllAttributes(), ss.ScanView(view)))
    found_only = ss.Context(0)
    found_only.set_option("specialize", 2)
    compiling = ss.Context(0)
    compiling.set_option("specialize", 1)
    before = ss.memory_stats()["rtc_compilations"]
    a = ss.Plan(op, found_only)
    a.run()
    want = to_cols(a.fetch())
    assert a.specialized() == 0 and "not in the kernel cache" in a.specialize_reason() and ss.memory_stats()["rtc_compilations"] == before
    b = ss.Plan(op, compiling)
    b.run()
    assert b.specialized() == 1 and ss.memory_stats()["rtc_compilations"] == before + 1
    c = ss.Plan(op, found_only)            # the same program: found in this process
    c.run()
    assert c.specialized() == 1 and ss.memory_stats()["rtc_compilations"] == before + 1
    assert_cols_equal(to_cols(c.fetch()), want)
    assert_cols_equal(to_cols(b.fetch()), want)
    a.specialize()                          # the plan that had missed it asks: found now, still no second compilation
    a.run()
    assert a.specialized() == 1 and ss.memory_stats()["rtc_compilations"] == before + 1
    assert_cols_equal(to_cols(a.fetch()), want)


def test_recent_kernel_times_ring(gpu_ctx):
    # ssgpu_plan_recent_kernel_ms: one (positive) duration per profiled run, oldest first, at most 256 kept
    view = make_view(200003)
    plan = ss.Plan(fpa_narrow(view), gpu_ctx)
    assert plan.recent_kernel_ms() == []
    for _ in range(5):
        plan.run()
    times = plan.recent_kernel_ms()
    assert len(times) == 5 and all(t > 0.0 for t in times)
    assert len(plan.recent_kernel_ms(3)) == 3
    for _ in range(260):
        plan.run()
    assert len(plan.recent_kernel_ms(1000)) == 256
    assert abs(plan.counters().dominant_ms - plan.recent_kernel_ms(1)[0]) < 1e-6


_CACHE_CHILD = r"""
import json, sys
import numpy as np
import supersonic_amd as ss
ctx = ss.Context(0)
rng = np.random.default_rng(3)
n = 50021
schema = ss.TupleSchema([ss.Attribute("a", ss.INT64), ss.Attribute("b", ss.INT64), ss.Attribute("d", ss.DOUBLE)])
view = ss.View(schema, [rng.integers(0, 1000, n), rng.integers(0, 1000, n), rng.integers(0, 4000, n) * 0.25])
NA = ss.NamedAttribute
op = ss.ScalarAggregate(ss.AggregationSpecification().AddAggregation(ss.SUM, "s", "sum_s").AddAggregation(ss.MAX, "d", "mx").AddAggregation(ss.COUNT, "", "n"),
                        ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(499)), ss.ProjectAllAttributes(),
                                  ss.Compute(ss.CompoundExpression().Add(NA("a")).AddAs("s", ss.Plus(NA("a"), NA("b"))).Add(NA("d")), ss.ScanView(view))))
policy = sys.argv[1] if len(sys.argv) > 1 else "ask"
if policy == "ask":
    plan = ss.Plan(op, ctx).specialize()
else:                                   # the context's policy decides ("default": the library's own default, no option set)
    if policy == "background":
        ctx.set_option("specialize", 3)
        ctx.set_option("specialize_min_rows", 1)
    elif policy != "default":
        ctx.set_option("specialize", int(policy))
    plan = ss.Plan(op, ctx)
plan.run()
got = plan.fetch()
st = ss.memory_stats()
print(json.dumps({"specialized": plan.specialized(), "compilations": st["rtc_compilations"], "disk_hits": st["rtc_disk_hits"],
                  "row": [got.column(i).data[0].item() for i in range(got.column_count())]}))
"""


def test_specialised_kernels_survive_the_process(tmp_path):
    """The on-disk cache of code objects (rtc.cpp): the second PROCESS that wants the same kernel loads it instead of compiling
    for seconds; a damaged cache file is ignored and replaced; SSGPU_RTC_CACHE_DIR= (empty) switches the cache off."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def child(cache_dir, policy="ask"):
        env = dict(os.environ, SSGPU_RTC_CACHE_DIR=cache_dir, PYTHONPATH=root)
        env.pop("SSGPU_SPECIALIZE", None)          # (conftest pins 0 for the suite: the children see the library's own default)
        out = subprocess.run([sys.executable, "-c", _CACHE_CHILD, policy], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])

    cache = str(tmp_path / "rtc")
    # the default policy (specialize = 3) below its row threshold: compiled kernels where they exist, never a compilation
    cold = child(cache, "default")
    assert cold["specialized"] == 0 and cold["compilations"] == 0 and cold["disk_hits"] == 0, cold
    # a process that ENDS while the worker compiles for it waits for that compilation (seconds), exits cleanly, and leaves the code object
    # to the next process
    bg_cache = str(tmp_path / "rtc_bg")
    leaving = child(bg_cache, "background")
    assert leaving["specialized"] == 0 and leaving["disk_hits"] == 0 and leaving["row"] == cold["row"], leaving
    assert len([f for f in os.listdir(bg_cache) if f.endswith(".co")]) == 1
    heir = child(bg_cache, "background")
    assert heir["specialized"] == 1 and heir["compilations"] == 0 and heir["disk_hits"] >= 1 and heir["row"] == cold["row"], heir
    first = child(cache)
    assert first["specialized"] == 1 and first["compilations"] >= 1 and first["disk_hits"] == 0, first
    files = [f for f in os.listdir(cache) if f.endswith(".co")]
    assert len(files) == first["compilations"], (files, first)
    second = child(cache)
    assert second["specialized"] == 1 and second["compilations"] == 0 and second["disk_hits"] >= 1, second      # loaded, not compiled
    assert second["row"] == first["row"] == cold["row"]
    for policy in ("default", "2"):             # ... and a plan that never asked finds it, too
        warm = child(cache, policy)
        assert warm["specialized"] == 1 and warm["compilations"] == 0 and warm["disk_hits"] >= 1 and warm["row"] == first["row"], (policy, warm)
    never = child(cache, "0")
    assert never["specialized"] == 0 and never["compilations"] == 0 and never["disk_hits"] == 0 and never["row"] == first["row"], never
    path = os.path.join(cache, files[0])
    with open(path, "r+b") as f:                                    # a torn file: checksum fails -> compiled again, file replaced
        f.truncate(os.path.getsize(path) // 2)
    third = child(cache)
    assert third["specialized"] == 1 and third["compilations"] >= 1 and third["row"] == first["row"], third
    assert child(cache)["compilations"] == 0                        # ... and the replacement is good
    off = child("")
    assert off["compilations"] >= 1 and off["disk_hits"] == 0, off   # no disk cache at all


_TRIM_CHILD = r"""
import sys
import numpy as np
import supersonic_amd as ss
ctx = ss.Context(0)
ctx.set_option("specialize", 1)
n = 5003
schema = ss.TupleSchema([ss.Attribute("a", ss.INT64), ss.Attribute("b", ss.INT64)])
view = ss.View(schema, [np.arange(n), np.arange(n)])
for k in sys.argv[1:]:
    op = ss.ScalarAggregate(ss.AggregationSpecification().AddAggregation(ss.SUM, "b", "s"),
                            ss.Filter(ss.Less(ss.NamedAttribute("a"), ss.ConstInt64(int(k))), ss.ProjectAllAttributes(), ss.ScanView(view)))
    plan = ss.Plan(op, ctx)
    plan.run()
    assert plan.specialized() == 1, plan.specialize_reason()
print("rtc_disk_hits", ss.memory_stats()["rtc_disk_hits"])
"""


def test_the_disk_cache_of_code_objects_is_bounded(tmp_path):
    """$SSGPU_RTC_CACHE_MAX_MB: after a store, the least recently USED code objects go until the directory is below 3/4 of the limit (a
    load counts as a use); the newest file always stays; <= 0 switches the limit off."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cache = str(tmp_path / "rtc")

    def child(limit_mb, *constants):
        env = dict(os.environ, SSGPU_RTC_CACHE_DIR=cache, SSGPU_RTC_CACHE_MAX_MB=str(limit_mb), PYTHONPATH=root)
        out = subprocess.run([sys.executable, "-c", _TRIM_CHILD] + [str(c) for c in constants], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
        assert out.returncode == 0, out.stderr[-2000:]
        return int([l for l in out.stdout.splitlines() if l.startswith("rtc_disk_hits")][-1].split()[1])

    def files():
        return sorted(f for f in os.listdir(cache) if f.endswith(".co"))
    child(0, 11, 12, 13, 14)                       # no limit: four kernels, four files
    four = files()
    assert len(four) == 4
    size = max(os.path.getsize(os.path.join(cache, f)) for f in four)
    time.sleep(0.05)
    assert child(0, 12) == 1                       # kernel 12 is USED again (loaded from its file): it becomes the most recent of the four
    used = max(four, key=lambda f: os.stat(os.path.join(cache, f)).st_atime_ns)
    limit_mb = 3.5 * size / (1024.0 * 1024.0)      # room for three and a half files: storing a fifth trims down to 3/4 of that = two files
    child(limit_mb, 15)
    left = files()
    assert len(left) == 2 and used in left, (left, used, four)
    assert len(set(left) - set(four)) == 1         # ... and the one just stored


# ---- INPUT LIFETIME (include/ssgpu.h, ABI 7): under the defaults a run is settled when ssgpu_plan_run returns -- run feedback read,
# ---- a NaN-exact repeat done -- so the input may be overwritten once the stream has been synchronised, and the result is still right.
# ---- (With "lazy_feedback" = 1, the sharded drivers' opt-in, the same sequence may repeat a run from the overwritten columns.) ----------
def test_input_may_be_overwritten_after_a_synchronised_run():
    torch = pytest.importorskip("torch")
    from oracle import oracle
    from helpers import assert_cols_equal, sort_rows, to_cols
    n = 300000
    rng = np.random.default_rng(11)
    schema = ss.TupleSchema([ss.Attribute("k", ss.INT64), ss.Attribute("v", ss.INT64), ss.Attribute("d", ss.DOUBLE)])

    def host(groups, nan_first=False):
        d = rng.integers(-4000, 4000, n) * 0.25
        if nan_first:
            d[:5000] = np.nan                      # leading NaNs: the reference keeps a group's first value when it is a NaN
        return [rng.integers(0, groups, n), rng.integers(-1000, 1000, n), d]
    spec = (ss.AggregationSpecification().AddAggregation(ss.SUM, "v", "s").AddAggregation(ss.COUNT, "", "c").AddAggregation(ss.MIN, "d", "mn")
            .AddAggregation(ss.MAX, "d", "mx"))
    few, many, nans = host(3000), host(250000), host(3000, nan_first=True)
    dev = torch.device("cuda", 0)
    tensors = [torch.from_numpy(np.ascontiguousarray(c)).to(dev) for c in few]
    view = ss.DeviceView(schema, [(t.data_ptr(), 0) for t in tensors], n)
    ctx = ss.Context(0)                               # defaults: lazy_feedback = 0
    ctx.set_option("group_capacity", 8192)            # 3000 groups fit the direct shape's table, 250000 overflow it (a repeated run)
    ctx.set_option("group_dense", 0)
    op = lambda cols: ss.GroupAggregate(ss.ProjectNamedAttributes(["k"]), spec, None, ss.ScanView(ss.View(schema, cols)))   # noqa: E731
    plan = ss.Plan(op(few), ctx)                      # (bound over a host View of the same schema; every run below takes the device columns)
    torch.cuda.synchronize()
    for _ in range(4):                                # into the steady state (where the lazy mode would stop reading feedback back)
        plan.run(view)
    for cols in (many, nans, few):
        for t, c in zip(tensors, cols):
            t.copy_(torch.from_numpy(np.ascontiguousarray(c)))
        torch.cuda.synchronize()
        plan.run(view)
        ctx.synchronize()
        for t in tensors:                             # the caller is done with its input: overwritten before the result is touched
            t.zero_()
        torch.cuda.synchronize()
        _s, want = oracle.run(op(cols))
        assert_cols_equal(sort_rows(to_cols(plan.fetch())), sort_rows(want), context="input overwritten after run + synchronise")


# ---- two contexts driven from two host threads at the same time (each context and its plans by ONE thread, as the reference's
# ---- cursors: cursor.h:131-148): the library's thread-local state (memory quota scope, kernel-cache policy, lowering scratch) and its
# ---- process-wide caches (compiled kernels, memory statistics) must not leak from one thread's plans into the other's ------------------
def test_two_contexts_from_two_threads_do_not_disturb_each_other():
    from oracle import oracle
    from helpers import assert_cols_equal, sort_rows, to_cols
    errors = []

    def worker(seed, specialize, quota):
        try:
            ctx = ss.Context(0)
            ctx.set_option("specialize", specialize)
            for round_no in range(6):
                view = make_view(60001 + 1000 * round_no, seed=seed + round_no, nullable=bool(round_no % 2))
                ops = [fpa_narrow(view), group_query(view, bool(round_no % 2), ("k1",) if round_no % 2 else ("k1", "k2"))]
                for op in ops:
                    plan = ss.Plan(op, ctx)
                    if quota and round_no == 3 and op is ops[1]:
                        plan.set_memory_limit(4096)                       # this thread's plan runs out of quota; the other thread's never does
                        with pytest.raises(ss.SupersonicException) as e:
                            plan.run()
                            plan.fetch()
                        assert e.value.return_code == ss.ERROR_MEMORY_EXCEEDED
                        continue
                    plan.run()
                    _s, want = oracle.run(op)
                    assert_cols_equal(sort_rows(to_cols(plan.fetch())), sort_rows(want), context="thread %d round %d" % (seed, round_no))
        except BaseException as e:   # noqa: BLE001 -- reported by the main thread
            errors.append((seed, repr(e)))
    threads = [threading.Thread(target=worker, args=(100, 1, True)), threading.Thread(target=worker, args=(200, 0, False)),
               threading.Thread(target=worker, args=(300, 1, False))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors


# ---- chunked staging of a host-resident input (ssgpu_plan_run_host) ------------------------------------------------------------------
@pytest.mark.parametrize("nullable", [False, True])
@pytest.mark.parametrize("n,chunk", [(0, 1000), (1, 1000), (999, 1000), (1000, 1000), (1001, 1000), (70001, 4096), (70001, 1 << 20), (300007, 50000), (300007, 0)])
def test_chunked_staging_of_a_host_input_gives_the_single_run_answer(gpu_ctx, n, chunk, nullable):
    """The rows travel through two alternating sets of device columns, chunk k + 1 copied while chunk k is read; every chunk leaves a
    partial state and one launch folds them in row order.  Same row as the oracle's -- FIRST / LAST by global row order, counts,
    integer and (exactly representable) floating sums, MIN / MAX, under a Filter and a Compute -- whatever the chunking."""
    from helpers import to_cols, assert_cols_equal
    from oracle import oracle
    view = make_view(n, nullable=nullable)
    NA = ss.NamedAttribute
    e = (ss.CompoundExpression().Add(NA("a")).AddAs("s", ss.Plus(NA("a"), NA("b"))).Add(NA("d")).Add(NA("d0")).Add(NA("d1")).Add(NA("u")).Add(NA("t")).Add(NA("k1")))
    spec = (ss.AggregationSpecification().AddAggregation(ss.SUM, "s", "ss").AddAggregation(ss.COUNT, "", "n").AddAggregation(ss.COUNT, "d0", "c0")
            .AddAggregation(ss.MIN, "d", "mn").AddAggregation(ss.MAX, "d0", "mx").AddAggregation(ss.SUM, "d1", "s1").AddAggregation(ss.FIRST, "d0", "f0")
            .AddAggregation(ss.LAST, "u", "lu").AddAggregation(ss.LAST, "t", "lt").AddAggregation(ss.FIRST, "k1", "fk"))
    op = ss.ScalarAggregate(spec, ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(499)), ss.ProjectAllAttributes(), ss.Compute(e, ss.ScanView(view))))
    _schema, want = oracle.run(op)
    plan = ss.Plan(op, gpu_ctx)
    for _ in range(2):                       # (the staging sets are reused by the second call)
        plan.run_host(chunk_rows=chunk)
        assert_cols_equal(to_cols(plan.fetch()), want, context="chunked staging n=%d chunk=%d" % (n, chunk))
    plan.run()                               # and the ordinary form still runs on the same plan
    assert_cols_equal(to_cols(plan.fetch()), want)


def test_chunked_staging_reports_an_evaluation_error_of_any_chunk_and_refuses_other_plans(gpu_ctx):
    NA = ss.NamedAttribute
    n = 10000
    schema = ss.TupleSchema([ss.Attribute("a", ss.INT64), ss.Attribute("b", ss.INT64)])
    b = np.ones(n, dtype=np.int64)
    b[137] = 0                               # the FIRST chunk divides by zero; the later chunks must not wipe the flag
    view = ss.View(schema, [np.arange(n), b])
    op = ss.ScalarAggregate(ss.AggregationSpecification().AddAggregation(ss.SUM, "q", "sq"),
                            ss.Compute(ss.CompoundExpression().AddAs("q", ss.DivideSignaling(NA("a"), NA("b"))), ss.ScanView(view)))
    plan = ss.Plan(op, gpu_ctx)
    with pytest.raises(ss.SupersonicException) as e:
        plan.run_host(chunk_rows=1000)
        plan.fetch()
    assert e.value.return_code == ss.ERROR_EVALUATION_ERROR
    # (a Sort needs every row before its first result row: no chunked form -- tests/test_chunked_gpu.py has the forms that exist)
    ordered = ss.Plan(ss.Sort(ss.SortOrder().add("a", ss.ASCENDING), ss.ProjectAllAttributes(), 0, ss.ScanView(view)), gpu_ctx)
    with pytest.raises(ss.SupersonicException) as e:
        ordered.run_host(chunk_rows=1000)
    assert e.value.return_code == ss.ERROR_NOT_IMPLEMENTED


@pytest.mark.parametrize("n,block,chunk", [(0, 1024, 4096), (1, 1024, 4096), (70001, 1024, 4096), (70001, 1000, 4096), (70001, 8192, 1000), (300007, 1024, 1 << 16), (300007, 77777, 0)])
def test_push_form_of_chunked_staging_takes_the_blocks_of_a_child_cursor(gpu_ctx, n, block, chunk):
    """ssgpu_plan_stream_begin / _push / _finish: the input arrives as the reference's cursors hand it out -- blocks of <= 1024 rows in ONE
    buffer that the next block overwrites (cursor.h:131-148) -- is copied into pinned staging sets, and a full set runs on the device while
    the other one fills.  Same row as the oracle's over the whole input."""
    from helpers import to_cols, assert_cols_equal
    from oracle import oracle
    view = make_view(n, nullable=True)
    NA = ss.NamedAttribute
    spec = (ss.AggregationSpecification().AddAggregation(ss.SUM, "b", "sb").AddAggregation(ss.COUNT, "", "n").AddAggregation(ss.COUNT, "d0", "c0")
            .AddAggregation(ss.MIN, "d", "mn").AddAggregation(ss.MAX, "d0", "mx").AddAggregation(ss.SUM, "d1", "s1").AddAggregation(ss.FIRST, "d0", "f0")
            .AddAggregation(ss.LAST, "u", "lu").AddAggregation(ss.LAST, "t", "lt"))
    op = ss.ScalarAggregate(spec, ss.Filter(ss.Greater(NA("b"), ss.ConstInt64(499)), ss.ProjectAllAttributes(), ss.ScanView(view)))
    _schema, want = oracle.run(op)
    schema = view.schema()
    # the child's ONE output block: every Next() overwrites it
    buf = [(np.zeros(block, dtype=view.column(i).data.dtype), None if view.column(i).is_null is None else np.zeros(block, dtype=bool)) for i in range(view.column_count())]

    def blocks():
        for lo in range(0, n, block):
            m = min(block, n - lo)
            for i, (d, z) in enumerate(buf):
                d[:m] = view.column(i).data[lo:lo + m]
                if z is not None:
                    z[:m] = view.column(i).is_null[lo:lo + m]
            yield ss.View(schema, [ss.Column(d[:m], None if z is None else z[:m]) for (d, z) in buf], m)
            for d, _z in buf:
                d[:m] = 0                    # (the memory is the child's again)
    plan = ss.Plan(op, gpu_ctx)
    for _ in range(2):
        plan.stream(blocks(), chunk_rows=chunk)
        assert_cols_equal(to_cols(plan.fetch()), want, context="push staging n=%d block=%d chunk=%d" % (n, block, chunk))
    plan.run()
    assert_cols_equal(to_cols(plan.fetch()), want)
