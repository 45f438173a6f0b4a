#!/usr/bin/env python3
"""bench.py -- rows/sec of the fused Filter -> Project -> Aggregate path on MI355X.

Default workload (BASELINE.json configs[1], SURVEY.md 8(d) "Q-FPA-wide"): per GPU a device-resident
100 M-row x 8-column Block (a,b,c,d INT64; d0..d3 DOUBLE) and the reference plan

    ScalarAggregate(SUM(s), COUNT(*), SUM(c), MIN(d), MAX(d0), SUM(d1), SUM(p))
      o Filter(a > 499, ProjectAllAttributes)
      o Compute(a, a+b AS s, c, d, d0, d1, d2*d3 AS p)
      o ScanView(block)

`--query group` is BASELINE configs[2] / configs[3] ("Q-GROUP-F"): Filter(a > 499) ->
GroupAggregate(k1, k2; SUM / MIN / MAX x d0..d3), ~1e5 groups; with N > 1 ranks every rank aggregates
its row-range shard and the partial tables meet in ONE RCCL all-gather followed by a merge plan
(supersonic_amd.distributed.DeviceShardedGroupAggregate).

One "step" = one full pass of the plan over the resident rows (inputs already in HBM).  `--scaling
weak` (default): --rows rows PER GPU; `--scaling strong`: --rows rows IN TOTAL, split into row ranges.
`--gpus N` with N > 1 re-executes itself under `python -m torch.distributed.run` (one rank per GPU)
unless it is already running under it (RANK / WORLD_SIZE set by the launcher).

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per launch / average duration
of the plan's kernels measured with HIP events on the launch stream inside the library.
`--dry-run` (CPU, gloo, no kernels) exercises only the launch / collective / report plumbing and says
so in the line (`"dry_run": true`, value 0): it is what the CPU test of the N > 1 path runs.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
K_FILTER = 499
N_GROUPS = 100000


# ---- the two workloads ------------------------------------------------------------------------
def build_plan(ss, view, k_filter=K_FILTER):
    NA = ss.NamedAttribute
    compute = (ss.CompoundExpression().Add(NA("a")).AddAs("s", ss.Plus(NA("a"), NA("b"))).Add(NA("c")).Add(NA("d"))
               .Add(NA("d0")).Add(NA("d1")).AddAs("p", ss.Multiply(NA("d2"), NA("d3"))))
    spec = (ss.AggregationSpecification().AddAggregation(ss.SUM, "s", "sum_s").AddAggregation(ss.COUNT, "", "cnt")
            .AddAggregation(ss.SUM, "c", "sum_c").AddAggregation(ss.MIN, "d", "min_d")
            .AddAggregation(ss.MAX, "d0", "max_d0").AddAggregation(ss.SUM, "d1", "sum_d1")
            .AddAggregation(ss.SUM, "p", "sum_p"))
    return ss.ScalarAggregate(spec, ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(k_filter)), ss.ProjectAllAttributes(),
                                              ss.Compute(compute, ss.ScanView(view))))


def build_sort_plan(ss, view):
    """BASELINE configs[4] / SURVEY 8(d) Q-SORT: Sort(d ASC) over the 8-column block, every column in the result."""
    return ss.Sort(ss.SortOrder().add("d", ss.ASCENDING), ss.ProjectAllAttributes(), 0, ss.ScanView(view))


def build_filter_mat_plan(ss, view, k_filter=K_FILTER):
    """SURVEY 8(d) Q-FILTER-mat: Filter(a > K, ProjectAllAttributes), the survivors materialised."""
    return ss.Filter(ss.Greater(ss.NamedAttribute("a"), ss.ConstInt64(k_filter)), ss.ProjectAllAttributes(), ss.ScanView(view))


def bench_schema(ss):
    return ss.TupleSchema([ss.Attribute(n, ss.INT64) for n in ("a", "b", "c", "d")] +
                          [ss.Attribute(n, ss.DOUBLE) for n in ("d0", "d1", "d2", "d3")])


def group_schema(ss):
    return ss.TupleSchema([ss.Attribute("a", ss.INT64), ss.Attribute("k1", ss.INT32), ss.Attribute("k2", ss.INT32)] +
                          [ss.Attribute(n, ss.DOUBLE) for n in ("d0", "d1", "d2", "d3")])


def group_spec(ss):
    spec = ss.AggregationSpecification()
    for c in ("d0", "d1", "d2", "d3"):
        spec.AddAggregation(ss.SUM, c, "sum_" + c).AddAggregation(ss.MIN, c, "min_" + c).AddAggregation(ss.MAX, c, "max_" + c)
    return spec


LAYOUT = "library"      # --layout: where the resident columns live (a Block the library lays out / one torch allocation per column)
QUERY_NAME = "wide"     # the --query as given (group3 included): names the committed PMC pass of the workload
GROUP_FILTER = True     # --query group3 (BASELINE configs[2]: no Filter below the GroupAggregate) clears it


def group_child(ss, view):
    if not GROUP_FILTER:
        return ss.ScanView(view)
    return ss.Filter(ss.Greater(ss.NamedAttribute("a"), ss.ConstInt64(K_FILTER)), ss.ProjectAllAttributes(), ss.ScanView(view))


def build_group_plan(ss, view):
    return ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), group_spec(ss), None, group_child(ss, view))


def gen_device_columns(torch, rows, seed, device, row0=0):
    """Synthetic block of SURVEY 8(d): every partial DOUBLE sum is exact, so the result is
    bit-exact under any reduction order."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    a = torch.randint(0, 1000, (rows,), generator=g, device=device, dtype=torch.int64)
    b = torch.randint(0, 1000, (rows,), generator=g, device=device, dtype=torch.int64)
    c = (torch.arange(rows, device=device, dtype=torch.int64) + row0) % 100000
    d = torch.randint(-(1 << 62), 1 << 62, (rows,), generator=g, device=device, dtype=torch.int64)
    d0 = torch.randint(-1000000, 1000001, (rows,), generator=g, device=device, dtype=torch.int64).to(torch.float64)
    d1 = torch.randint(0, 4000, (rows,), generator=g, device=device, dtype=torch.int64).to(torch.float64) * 0.25
    d2 = torch.randint(0, 64, (rows,), generator=g, device=device, dtype=torch.int64).to(torch.float64)
    d3 = torch.randint(0, 64, (rows,), generator=g, device=device, dtype=torch.int64).to(torch.float64)
    return [a, b, c, d, d0, d1, d2, d3]


def gen_group_columns(torch, rows, seed, device):
    """a | k1 = g / 317, k2 = g % 317 with g uniform in [0, 1e5) | d0..d3 as above."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    a = torch.randint(0, 1000, (rows,), generator=g, device=device, dtype=torch.int64)
    grp = torch.randint(0, N_GROUPS, (rows,), generator=g, device=device, dtype=torch.int64)
    k1 = (grp // 317).to(torch.int32)
    k2 = (grp % 317).to(torch.int32)
    del grp
    d0 = torch.randint(-1000000, 1000001, (rows,), generator=g, device=device, dtype=torch.int64).to(torch.float64)
    d1 = torch.randint(0, 4000, (rows,), generator=g, device=device, dtype=torch.int64).to(torch.float64) * 0.25
    d2 = torch.randint(0, 64, (rows,), generator=g, device=device, dtype=torch.int64).to(torch.float64)
    d3 = torch.randint(0, 64, (rows,), generator=g, device=device, dtype=torch.int64).to(torch.float64)
    return [a, k1, k2, d0, d1, d2, d3]


def host_columns(np, query, n, seed=42):
    rng = np.random.default_rng(seed)
    if query == "group":
        grp = rng.integers(0, N_GROUPS, n)
        return [rng.integers(0, 1000, n), (grp // 317).astype(np.int32), (grp % 317).astype(np.int32),
                rng.integers(-1000000, 1000001, n).astype(np.float64), rng.integers(0, 4000, n) * 0.25,
                rng.integers(0, 64, n).astype(np.float64), rng.integers(0, 64, n).astype(np.float64)]
    return [rng.integers(0, 1000, n), rng.integers(0, 1000, n), np.arange(n, dtype=np.int64) % 100000,
            rng.integers(-(1 << 62), 1 << 62, n), rng.integers(-1000000, 1000001, n).astype(np.float64),
            rng.integers(0, 4000, n) * 0.25, rng.integers(0, 64, n).astype(np.float64),
            rng.integers(0, 64, n).astype(np.float64)]


def place_in_library_block(ss, torch, ctx, schema, cols, rows, device):
    """The resident input as a device Block the LIBRARY lays out (ssgpu_block_create: one arena, column bases skewed against HBM
    channel conflicts -- include/ssgpu.h "device-resident Block"): the synthetic columns are generated by torch and copied
    into the block's columns once, before any timing.  -> (block, torch tensors over the block's columns)."""
    blk = ss.DeviceBlock(schema, rows, ctx)
    placed = []
    for i, t in enumerate(cols):
        ts = {torch.int64: "<i8", torch.float64: "<f8", torch.int32: "<i4"}[t.dtype]
        dst = torch.as_tensor(_DevPtr(blk.column_ptr(i), rows, ts), device=device)
        dst.copy_(t)
        placed.append(dst)
    torch.cuda.synchronize(device)
    return blk, placed


class _DevPtr(object):
    """Expose a raw device buffer to torch through the CUDA array interface."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2}


def pmc_source(query, alg_bytes):
    """The committed PMC pass `pmc_traffic` reads for this workload (its path relative to the repository), or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc*.json")), reverse=True):
        try:
            with open(path) as f:
                j = json.load(f)
            if j.get("query", "wide") == query and abs(int(j.get("algorithmic_bytes_per_launch", -1)) - int(alg_bytes)) <= int(alg_bytes) // 1000:
                return os.path.relpath(path, ROOT)
        except (OSError, ValueError, KeyError):
            continue
    return None


def pmc_traffic(query, alg_bytes):
    """HBM bytes per launch of the stage's kernels from the committed PMC passes (profiles/rNN_pmc_<query>.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE, one pass each, on this same command; FETCH_SIZE x 2 as calibrated on known byte counts,
    tools/pmc_calibrate.sh).  PMC counters cannot be read from inside the timed process, so the value is only reported
    when the committed pass measured the same workload (same query, same algorithmic bytes)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc*.json")), reverse=True):
        try:
            with open(path) as f:
                j = json.load(f)
            if j.get("query", "wide") == query and abs(int(j.get("algorithmic_bytes_per_launch", -1)) - int(alg_bytes)) <= int(alg_bytes) // 1000:
                return float(j["traffic_bytes_per_launch"])
        except (OSError, ValueError, KeyError):
            continue
    return None


# ---- HBM traffic of THIS run's command, counted by rocprofv3 around child runs of it ------------------------------------------
PMC_STAGE = {"wide": ["ssgpu_pipeline_kernel", "ssgpu_finish_slots", "ssgpu_emit_scalar"],
             "group3": ["ssgpu_part_scatter_plain", "ssgpu_part_agg", "ssgpu_group_extract", "ssgpu_group_count", "ssgpu_scan_counts", "ssgpu_fill", "ssgpu_group_init"],
             "group": ["ssgpu_part_scatter_plain", "ssgpu_part_agg", "ssgpu_group_extract", "ssgpu_group_count", "ssgpu_scan_counts", "ssgpu_fill", "ssgpu_group_init"],
             "sort": ["ssgpu_sort_"], "filter_mat": ["ssgpu_pipeline_kernel", "ssgpu_scan_counts"]}


def measure_traffic(query, rows, alg_bytes, timeout_s=240):
    """HBM bytes per step of the stage's kernels, counted for THIS command on THIS box: two child runs of bench.py under
    `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE -- one counter per pass and no trace domain next to it, as the guide's
    HBM section prescribes), 3 timed steps each, the per-kernel averages of the steady-state launches summed.  Unit and gfx950
    correction as calibrated on known byte counts (profiles/r04_pmc_calibration.json, r04_pmc_gather_calibration.json): both
    counters are KiB; FETCH_SIZE tallies 128-byte streaming requests at 64 bytes (x 2) except for the Sort's record gather,
    whose random 64-byte reads are counted in full.  Returns (bytes per step, per-kernel dict) or (None, reason)."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 is not on PATH"
    counts = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="ssgpu_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp", SSGPU_BENCH_CHILD="1")
        cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
               "--query", query, "--rows", str(rows), "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-configs", "--no-traffic"]
        try:
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s, text=True)
        except (OSError, subprocess.TimeoutExpired) as e:
            shutil.rmtree(out, ignore_errors=True)
            return None, "%s pass: %s" % (counter, e)
        files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
        if p.returncode != 0 or not files:
            shutil.rmtree(out, ignore_errors=True)
            return None, "%s pass: rc %d, %d counter files: %s" % (counter, p.returncode, len(files), p.stdout[-300:])
        acc = collections.defaultdict(list)
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter and any(sub in row["Kernel_Name"] for sub in PMC_STAGE[query]):
                    acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
        shutil.rmtree(out, ignore_errors=True)
        counts[counter] = acc
    per_kernel, total = {}, 0.0
    for k in sorted(set(counts["FETCH_SIZE"]) | set(counts["WRITE_SIZE"])):
        n_per_step = 4 if "onesweep" in k else 1
        tail = lambda v: (sum(v[-3 * n_per_step:]) / max(len(v[-3 * n_per_step:]), 1)) * n_per_step if v else 0.0   # noqa: E731
        fk, wk = tail(counts["FETCH_SIZE"].get(k)), tail(counts["WRITE_SIZE"].get(k))
        ff = 1.0 if "sort_gather_rec" in k else 2.0
        per_kernel[k] = {"FETCH_SIZE_KiB_per_step": fk, "WRITE_SIZE_KiB_per_step": wk, "fetch_factor": ff, "launches_per_step": n_per_step}
        total += (fk * ff + wk) * 1024.0
    if total <= 0:
        return None, "no counter rows for the stage's kernels"
    return total, per_kernel


# ---- the other BASELINE configs, measured in the same process after the headline's timed region ------------------------------
def check_group_result(torch, plan, cols, device, with_filter):
    """checksums of checksums against torch over the same device columns (every DOUBLE is a small multiple of 0.25: exact)"""
    dv = plan.result_device_view()
    out_rows = dv.row_count()
    gcol = lambda i: torch.as_tensor(_DevPtr(dv._ptrs[i][0], out_rows, "<f8"), device=device)   # noqa: E731
    keep = (cols[0] > K_FILTER) if with_filter else None
    sel = (lambda t: t[keep]) if keep is not None else (lambda t: t)                              # noqa: E731
    gid = sel(cols[1].to(torch.int64) * 317 + cols[2])
    return {"groups_match": out_rows == int((torch.bincount(gid, minlength=N_GROUPS) > 0).sum().item()),
            "sum_of_sums_d0": float(gcol(2).sum().item()) == float(sel(cols[3]).sum().item()),
            "sum_of_sums_d1": float(gcol(5).sum().item()) == float(sel(cols[4]).sum().item()),
            "min_of_mins_d2": float(gcol(9).min().item()) == float(sel(cols[5]).min().item()),
            "max_of_maxes_d3": float(gcol(13).max().item()) == float(sel(cols[6]).max().item())}, out_rows


def check_rows_result(torch, plan, cols, device, query):
    """Sort / materialising Filter: the result is as large as the input -- it stays in HBM and is checked there"""
    dv = plan.result_device_view()
    out_rows = dv.row_count()
    col = lambda i, ts: torch.as_tensor(_DevPtr(dv._ptrs[i][0], out_rows, ts), device=device)   # noqa: E731
    if query == "sort":
        key = col(3, "<i8")
        return {"sorted": bool((key[1:] >= key[:-1]).all().item()) if out_rows > 1 else True,
                "key_sum_matches": int(key.sum().item()) == int(cols[3].sum().item()),
                "payload_sum_matches": int(col(0, "<i8").sum().item()) == int(cols[0].sum().item())}, out_rows
    keep = cols[0] > K_FILTER
    return {"rows_match": out_rows == int(keep.sum().item()),
            "first_column_matches": bool(torch.equal(col(0, "<i8"), cols[0][keep])),
            "last_column_matches": bool(torch.equal(col(7, "<f8"), cols[7][keep]))}, out_rows


def measure_config(ss, torch, ctx, device, query, rows, steps, wide_cols=None):
    """One of the BASELINE configs next to the headline (group3 = configs[2], group = configs[3]'s per-GPU query, sort =
    configs[4], filter_mat = SURVEY 8(d) Q-FILTER-mat) on this GPU: set-up until the plan's shape has settled, `steps` timed
    steps, the stage's kernel time from the library's HIP events, the result checked on the device."""
    global GROUP_FILTER
    saved_filter = GROUP_FILTER
    group = query in ("group", "group3")
    GROUP_FILTER = query == "group"
    try:
        if group:
            cols, schema = gen_group_columns(torch, rows, 42, device), group_schema(ss)
        else:
            cols, schema = (wide_cols if wide_cols is not None else gen_device_columns(torch, rows, 42, device)), bench_schema(ss)
        torch.cuda.synchronize(device)
        blk = None
        if group and LAYOUT == "library":
            blk, cols = place_in_library_block(ss, torch, ctx, schema, cols, rows, device)
        view = ss.DeviceView(schema, [(t.data_ptr(), 0) for t in cols], rows)
        plan = ss.Plan(build_group_plan(ss, view) if group else build_sort_plan(ss, view) if query == "sort" else build_filter_mat_plan(ss, view), ctx)

        def settled():
            st = ss.memory_stats()
            return (st["device_bytes"], st["rtc_compilations"], st["rtc_disk_hits"], json.dumps(plan.stage_info(), sort_keys=True))
        plan.run(view)
        torch.cuda.synchronize(device)
        for _ in range(8):
            before = settled()
            plan.run(view)
            torch.cuda.synchronize(device)
            if settled() == before:
                break
        for _ in range(3):
            plan.run(view)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            plan.run(view)
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
        kernel_ms = plan.recent_kernel_ms(256)[-steps:]
        if group:
            checked, out_rows = check_group_result(torch, plan, cols, device, GROUP_FILTER)
            alg = plan.counters().algorithmic_bytes
        else:
            checked, out_rows = check_rows_result(torch, plan, cols, device, query)
            alg = 128 * rows if query == "sort" else 64 * rows + 64 * out_rows
        kms = sum(kernel_ms) / max(len(kernel_ms), 1)
        out = {"rows": rows, "steps": steps, "ms_per_step": elapsed / steps * 1e3, "kernel_ms": kms, "rows_per_s": rows * steps / elapsed,
               "algorithmic_bytes_per_row": alg / max(rows, 1), "frac": (alg / (kms / 1e3) / 1e9 / HBM_PEAK_GBS) if kms > 0 else 0.0,
               "result_rows": out_rows, "checked": bool(all(checked.values())), "specialized_stages": plan.specialized(),
               "traffic_source": pmc_source(query, alg)}
        if not out["checked"]:
            out["failed_checks"] = [k for k, v in checked.items() if not v]
        del plan, view, cols, blk
        torch.cuda.empty_cache()
        return out
    finally:
        GROUP_FILTER = saved_filter


# ---- CPU baseline: the oracle (CPU restatement of the reference's 1024-row pull model) ----------
def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    """(physical cores, logical CPUs) of this host: distinct (physical id, core id) pairs of /proc/cpuinfo."""
    logical = os.cpu_count() or 1
    try:
        cores, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
        if cores:
            return len(cores), logical
    except OSError:
        pass
    return logical, logical


def _pin_cpus():
    """One logical CPU per PHYSICAL core this process may run on (first sibling of every (physical id, core id) of /proc/cpuinfo,
    intersected with the affinity mask): where the N-thread baseline pins its threads."""
    try:
        allowed = set(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = set(range(os.cpu_count() or 1))
    first, cpu, phys, core = {}, None, None, None
    try:
        with open("/proc/cpuinfo") as f:
            for line in list(f) + [""]:
                if line.startswith("processor"):
                    cpu = int(line.split(":", 1)[1])
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if cpu is not None and cpu in allowed:
                        first.setdefault((phys, core) if phys is not None and core is not None else ("cpu", cpu), cpu)
                    cpu = phys = core = None
    except OSError:
        pass
    return sorted(first.values()) or sorted(allowed)


def _cpu_quota():
    """CPUs' worth of time this process's cgroup may use (cgroup v2 cpu.max / v1 cpu.cfs_quota_us), or None if unlimited: threads
    beyond it are throttled, not run -- the GPU boxes of this pool report 256 CPUs and schedule about 16."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return max(1, int(round(int(quota) / float(period))))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0 and period > 0:
            return max(1, int(round(quota / float(period))))
    except (OSError, ValueError):
        pass
    return None


def _mem_available():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable"):
                    return int(line.split()[1]) * 1024
    except (OSError, ValueError):
        pass
    return 8 << 30


def cpu_baseline(ss, query, sample_rows, budget_s=10.0, with_config0=True, rows_per_thread=1_000_000):
    """The oracle (the CPU restatement of the reference's 1024-row pull model) on this host's cores, through its C harness
    (oracle/ss_oracle.c orc_bench_threads: pthreads, barrier-to-barrier wall time, nothing of Python inside the timed region):
      * 1 thread over `sample_rows` rows (the reference is single-threaded per plan);
      * one thread per PHYSICAL core, pinned, each over its own contiguous `rows_per_thread` rows that IT first touched
        (NUMA-local), the same plan per shard and -- GroupAggregate -- a final merge of the partial tables by key-hash ranges,
        which is timed and charged to every pass; Sort: N sorted runs, the merge of the runs is NOT included (an upper bound);
      * the same threads' plain read bandwidth over the same columns, the ceiling the N-thread figure is read against;
    plus BASELINE configs[0]'s exact shape (1 M rows x 4 INT64)."""
    import numpy as np
    from oracle import oracle
    n = sample_rows
    schema = group_schema(ss) if query == "group" else bench_schema(ss)
    make = {"group": lambda v: build_group_plan(ss, v), "sort": lambda v: build_sort_plan(ss, v),
            "filter_mat": lambda v: build_filter_mat_plan(ss, v)}.get(query, lambda v: build_plan(ss, v))
    cols = host_columns(np, query, n)
    row_bytes = sum(c.dtype.itemsize for c in cols)

    def drain(op):
        cur = oracle.Cursor(op)
        t0 = time.perf_counter()
        cur.drain_discard()
        return time.perf_counter() - t0

    def timed(bench, threads, cpus, budget):
        """passes sized from one untimed-for-the-report pass so that the measured run takes about `budget` seconds"""
        first = bench.run(threads, 1, cpus)
        per_pass = max(first["seconds"], 1e-6)
        passes = max(1, min(2000, int(budget / per_pass)))
        r = bench.run(threads, passes, cpus)
        r["passes"] = passes
        return r

    op = make(ss.View(schema, cols))
    pin = _pin_cpus()
    one_bench = oracle.ThreadedBench(op)
    r1 = timed(one_bench, 1, pin[:1], budget_s)
    one = n * r1["passes"] / r1["seconds"]
    # N threads: one contiguous row range per thread, filled by the thread itself from the 1-thread sample
    physical, nproc = _physical_cores()
    quota = _cpu_quota()
    nthreads = max(1, min(len(pin), physical, 256, quota or 256))     # one thread per physical core the cgroup actually schedules
    if nthreads < len(pin):                                             # spread the threads over the sockets / CCDs (memory channels)
        pin = [pin[i * len(pin) // nthreads] for i in range(nthreads)]
    per = int(min(rows_per_thread, n, max(65536, _mem_available() // 4 // max(nthreads * row_bytes, 1))))
    total = per * nthreads
    big = [np.empty(total, dtype=c.dtype) for c in cols]          # untouched pages: first written by the thread that reads them
    big_op = make(ss.View(schema, big))
    nbench = oracle.ThreadedBench(big_op, sample=op)
    rn = timed(nbench, nthreads, pin, budget_s / 2)
    per_pass = rn["seconds"] / rn["passes"] + (rn["merge_seconds"] or 0.0)
    stream = nbench.stream_read(nthreads, max(1, int(1.0 / max(total * row_bytes / 200e9, 1e-3))), pin)
    threads = {"value": total / per_pass, "cores": nthreads, "gb_per_s": total * row_bytes / per_pass / 1e9,
               "host_read_gb_per_s": stream / 1e9, "pinned": True,
               "sample": "%d pinned threads x %d rows each (copies of the 1-thread sample, first touched by their thread), %d passes, %.1f s"
                         % (nthreads, per, rn["passes"], rn["seconds"])}
    if rn["merge_seconds"] is not None:
        threads["merge_ms"] = rn["merge_seconds"] * 1e3
        threads["sample"] += "; + the merge of the %d partial tables (%d groups) by key-hash ranges, %.1f ms, charged to every pass" % (
            nthreads, rn["merged_groups"], rn["merge_seconds"] * 1e3)
    elif query == "sort":
        threads["sample"] += "; %d sorted runs -- the merge of the runs is NOT included (an upper bound for the CPU)" % nthreads
    del big, big_op, nbench
    out = {"value": one, "unit": "rows/s", "cores": 1, "kind": "port", "gb_per_s": one * row_bytes / 1e9,
           "sample": "%d rows (same plan, seed 42), %d passes, %.1f s of CPU" % (n, r1["passes"], r1["seconds"]),
           "threads": threads,
           "host": {"nproc": nproc, "physical_cores": physical, "cpu_quota_cores": quota, "model": _cpu_model()}}
    if not with_config0:
        return out
    # BASELINE configs[0]: Compute(a+b) -> Filter(a>K) -> Sum/Count on a 1 M-row x 4 INT64 table
    rng = np.random.default_rng(42)
    m = 1000000
    s4 = ss.TupleSchema([ss.Attribute(x, ss.INT64) for x in ("a", "b", "c", "d")])
    v4 = ss.View(s4, [rng.integers(0, 1000, m), rng.integers(0, 1000, m), np.arange(m, dtype=np.int64) % 100000, rng.integers(-(1 << 62), 1 << 62, m)])
    NA = ss.NamedAttribute
    op4 = ss.ScalarAggregate(ss.AggregationSpecification().AddAggregation(ss.SUM, "s", "sum_s").AddAggregation(ss.COUNT, "a", "cnt"),
                             ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(K_FILTER)), ss.ProjectAllAttributes(),
                                       ss.Compute(ss.CompoundExpression().Add(NA("a")).AddAs("s", ss.Plus(NA("a"), NA("b"))), ss.ScanView(v4))))
    r4, e4 = 0, 0.0
    while e4 < 2.0 and r4 < 500:
        e4 += drain(op4)
        r4 += 1
    out["config0"] = {"value": m * r4 / e4, "unit": "rows/s", "cores": 1,
                      "sample": "1M rows x 4 INT64: Compute(a, a+b) -> Filter(a>499) -> SUM, COUNT; %d passes" % r4}
    return out


def config_cpu_baseline(ss, q):
    """cpu_baseline of one of the extra configs (group3 / group / sort / filter_mat): 2 M-row sample, about 3 s of CPU."""
    global GROUP_FILTER
    saved = GROUP_FILTER
    GROUP_FILTER = q == "group"
    try:
        return cpu_baseline(ss, "group" if q in ("group", "group3") else q, 2_000_000, budget_s=2.0, with_config0=False)
    finally:
        GROUP_FILTER = saved


# ---- launch plumbing ----------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def relaunch_under_torchrun(n):
    """`bench.py --gpus N` started by hand: become N ranks (one per GPU) on this node."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=100_000_000, help="rows per GPU (weak scaling) or in total (strong scaling)")
    ap.add_argument("--query", choices=["wide", "group", "group3", "sort", "filter_mat"], default="wide",
                    help="wide = configs[1] (headline); group = configs[3]'s per-GPU query (Filter -> GroupAggregate); group3 = configs[2] "
                         "(GroupAggregate alone); sort = configs[4] (Sort(d) of the 8-column block); filter_mat = materialising Filter(a > 499)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--exchange", choices=["all_gather", "key_range", "dense"], default="dense",
                    help="--query group on N > 1 ranks: dense = dense-slot tables (SURVEY 8(e)): the ranks agree on the key ranges once, a step is "
                         "shard scan -> ONE all-to-all of slot slices -> element-wise fold + extraction, no merge plan; all_gather = every "
                         "packed partial table to every rank, every rank merges all of them; key_range = a partial row goes to the owner of "
                         "its key (one all-to-all), every rank merges 1/N of the groups")
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="0 = 16 M (wide) / 4 M (group)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not count HBM traffic with rocprofv3 --pmc child runs (roofline.traffic is then read from the committed pass)")
    ap.add_argument("--no-specialize", action="store_true",
                    help="run the interpreting pipeline kernel instead of the one specialised for the plan by runtime compilation")
    ap.add_argument("--extras", action="store_true", help="also report 1 % / 99 % selectivity and the PCIe-inclusive rate (N = 1)")
    ap.add_argument("--no-configs", action="store_true",
                    help="N = 1, --query wide: do not measure the other BASELINE configs (group3, group, sort, filter_mat) after the headline's timed region")
    ap.add_argument("--config-steps", type=int, default=30, help="timed steps of each of those configs")
    ap.add_argument("--no-regimes", action="store_true",
                    help="distributed runs: report only the regime --scaling names, not both (weak: --rows per GPU; strong: --rows in total)")
    ap.add_argument("--tile-rows", type=int, default=0)
    ap.add_argument("--lds-target", type=int, default=0)
    ap.add_argument("--grid-limit", type=int, default=0)
    ap.add_argument("--opts", default="", help="extra context options (development): key=value,key=value")
    ap.add_argument("--layout", choices=["library", "torch"], default="library",
                    help="library = the resident columns live in a device Block the library lays out (ssgpu_block_create: one arena, skewed column "
                         "bases); torch = one torch allocation per column, handed over as a DeviceView (what rounds 1-5 measured)")
    ap.add_argument("--stagger", type=int, default=-1,
                    help="development (A/B of HBM channel phase): place the input columns in ONE allocation, column i at i x (2 MiB-rounded size + this many "
                         "bytes); -1 = one allocation per column as torch makes them")
    ap.add_argument("--no-events-in-loop", action="store_true",
                    help="do not record the per-kernel HIP events during the timed steps (kernel time from a separate loop)")
    ap.add_argument("--force-distributed", action="store_true",
                    help="take the N > 1 code path (partial run + RCCL exchange + merge / finalize) even with one rank")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU / gloo, no kernels: only the launch, collective and report plumbing (tests)")
    return ap.parse_args(argv)


def metric_name(query):
    return {"wide": "rows/sec filter->project->aggregate, 100M x 8 INT64/DOUBLE",
            "sort": "rows/sec Sort (ORDER BY 1 INT64 key), 100M x 8 INT64/DOUBLE",
            "filter_mat": "rows/sec materialising Filter, 100M x 8 INT64/DOUBLE",
            }.get(query, "rows/sec filter->group-aggregate (2 INT32 keys, 1e5 groups, SUM/MIN/MAX x 4 DOUBLE)")


def dry_run(args, world, rank):
    """The N > 1 plumbing without a GPU: same launch contract, same collective shape (an all-gather of a small
    per-rank state + a fold), same barrier / max-over-ranks timing, same report -- no kernels, value 0."""
    import torch
    import torch.distributed as dist
    if world > 1 or args.force_distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if "RANK" in os.environ:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="gloo", rank=0, world_size=1)
    state = torch.full((56,), rank, dtype=torch.int64)
    gathered = torch.empty(world * 56, dtype=torch.int64)   # gloo wants the flat form
    collectives = 0

    def step():
        nonlocal collectives
        if dist.is_initialized():
            dist.all_gather_into_tensor(gathered, state)
            collectives += 1
            gathered.view(world, 56).sum(0)

    for _ in range(args.warmup):
        step()
    if dist.is_initialized():
        dist.barrier()
    collectives = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        assert gathered.view(world, 56)[:, 0].tolist() == list(range(world))
    # the second regime of a weak run (see main): the same loop once more, timed the same way
    regimes = None
    if dist.is_initialized() and args.scaling == "weak" and not args.no_regimes:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        regimes = {"weak": {"rows_per_gpu": args.rows, "rows_total": args.rows * world, "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "value": 0.0},
                   "strong": {"rows_total": args.rows // world * world, "rows_per_gpu": args.rows // world, "ms_per_step": float(t.item()) / max(args.steps, 1) * 1e3, "value": 0.0}}
        collectives //= 2
    if rank == 0:
        rows = args.rows if args.scaling == "weak" else args.rows // world
        line = {"metric": metric_name(args.query), "value": 0.0, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True,
                "scaling": args.scaling, "vs_baseline": None, "dtype": "int64/f64", "data": "none (dry run)", "dry_run": True,
                "config": {"workload": "dry run of --query %s: no kernels" % args.query, "rows_per_gpu": rows,
                           "collectives_per_step": collectives / max(args.steps, 1)}}
        if regimes is not None:
            line["regimes"] = regimes
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def extras(ss, torch, ctx, device, rows, cols, view):
    """Secondary figures (N = 1): the headline plan at 1 % / 99 % selectivity, and the PCIe-inclusive rate of the
    same plan over a pinned host block (staged on the copy stream, then run) -- never `value`."""
    import ctypes as C
    import numpy as np
    out = {}
    for name, k in (("selectivity_1pct", 989), ("selectivity_99pct", 9)):
        plan = ss.Plan(build_plan(ss, view, k), ctx)
        for _ in range(5):
            plan.run(view)
        ms = plan.recent_kernel_ms(256)[-3:]
        out[name] = {"filter": "a > %d" % k, "kernel_ms": sum(ms) / len(ms), "rows_per_s": rows / (sum(ms) / len(ms) / 1e3)}
        del plan
    # PCIe-inclusive: 16 M rows x 8 columns (1 GB) from pinned host memory
    n = min(rows, 16_000_000)
    host = host_columns(np, "wide", n)
    schema = bench_schema(ss)
    lib = ctx.lib
    blk = C.c_void_p()
    from supersonic_amd import _lib as L
    attrs = (L.Attr * 8)(*[L.Attr(schema.attribute(i).name().encode(), schema.attribute(i).type(), 0) for i in range(8)])
    ctx.check(lib.ssgpu_block_create(ctx.handle, attrs, 8, n, C.byref(blk)))
    pinned = []
    for c in host:
        p = C.c_void_p()
        ctx.check(lib.ssgpu_host_alloc(ctx.handle, c.nbytes, C.byref(p)))
        C.memmove(p, c.ctypes.data, c.nbytes)
        pinned.append(p)
    plan = ss.Plan(build_plan(ss, ss.DeviceView(schema, [(0, 0)] * 8, n)), ctx)
    best = None
    for _ in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        for i, p in enumerate(pinned):
            ctx.check(lib.ssgpu_block_upload(blk, i, p, None, 0, n))
        res = C.c_void_p()
        ctx.check(lib.ssgpu_plan_run_block(plan.handle, blk, C.byref(res)))
        ctx.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["pcie_inclusive"] = {"rows": n, "bytes": n * 64, "seconds": best, "rows_per_s": n / best, "GB_per_s": n * 64 / best / 1e9,
                             "note": "8 pinned host columns -> device block on the copy stream -> plan; best of 3"}
    for p in pinned:
        lib.ssgpu_host_free(ctx.handle, p)
    lib.ssgpu_block_destroy(blk)
    return out


def main():
    args = parse_args()
    global QUERY_NAME, LAYOUT
    QUERY_NAME = args.query
    LAYOUT = args.layout
    if args.query == "group3":
        global GROUP_FILTER
        GROUP_FILTER = False
        args.query = "group"
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args.gpus)          # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        return dry_run(args, world, rank)

    import torch
    import supersonic_amd as ss

    distributed = world > 1 or args.force_distributed
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # SSGPU_BENCH_SHARE_GPU=1 (development): every rank on device 0 and the collectives through the host over gloo -- RCCL refuses two
    # ranks of one device.  The N > 1 code path of this file on a one-GPU box; its timings mean nothing and the line says so.
    share_gpu = os.environ.get("SSGPU_BENCH_SHARE_GPU", "") == "1" and distributed
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if share_gpu:
            from supersonic_amd.distributed import _dist_for
            dist.init_process_group(backend="gloo")
            dist = _dist_for(None)
        elif "RANK" in os.environ:
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="nccl", device_id=device, rank=0, world_size=1)

    ctx = ss.Context(local_rank)
    # launch on torch's current stream so that RCCL collectives and our kernels are ordered
    stream = torch.cuda.current_stream(device)
    ctx.set_stream(stream.cuda_stream)
    if args.tile_rows:
        ctx.set_option("tile_rows", args.tile_rows)
    if args.lds_target:
        ctx.set_option("lds_target_bytes", args.lds_target)
    if args.grid_limit:
        ctx.set_option("grid_limit", args.grid_limit)
    # the plan's stages run kernels specialised for them (hiprtc, csrc/rtc.cpp: the interpreter's own handlers with the
    # opcode dispatch folded away; compiled at the first, untimed run below).  Same work, bit-identical results -- the
    # parity tests run both forms.
    ctx.set_option("specialize", 0 if args.no_specialize else 1)
    # timed steps are enqueued back to back and never wait for the host: the opt-in of ssgpu.h's INPUT LIFETIME rule (the
    # default settles every run before ssgpu_plan_run returns); the resident columns outlive every step
    ctx.set_option("lazy_feedback", 1)
    for kv in [x for x in args.opts.split(",") if x]:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))

    # rows of this rank: a contiguous row range of the job
    if args.scaling == "strong":
        lo, hi = args.rows * rank // world, args.rows * (rank + 1) // world
        rows, row_offset, total_rows_per_step = hi - lo, lo, args.rows
    else:
        rows, row_offset, total_rows_per_step = args.rows, rank * args.rows, args.rows * world
    group = args.query == "group"
    if group:
        cols = gen_group_columns(torch, rows, 42 + rank, device)
        schema = group_schema(ss)
    else:
        cols = gen_device_columns(torch, rows, 42 + rank, device, row_offset)
        schema = bench_schema(ss)
    input_block = None
    if args.stagger < 0 and args.layout == "library":
        input_block, cols = place_in_library_block(ss, torch, ctx, schema, cols, rows, device)
        torch.cuda.empty_cache()
    if args.stagger >= 0:
        pitch = [((t.numel() * t.element_size() + (2 << 20) - 1) // (2 << 20)) * (2 << 20) + args.stagger for t in cols]
        arena = torch.empty(sum(pitch) + (2 << 20), dtype=torch.uint8, device=device)
        base = (-arena.data_ptr()) % (2 << 20)
        placed, off = [], base
        for t, pch in zip(cols, pitch):
            nb = t.numel() * t.element_size()
            dst = arena[off:off + nb].view(t.dtype)
            dst.copy_(t)
            placed.append(dst)
            off += pch
        cols = placed
    torch.cuda.synchronize(device)
    view = ss.DeviceView(schema, [(t.data_ptr(), 0) for t in cols], rows)
    job = None
    if group and distributed:
        from supersonic_amd.distributed import DenseShardedGroupAggregate, DeviceShardedGroupAggregate, PlanDenseBackend
        if args.exchange == "dense":
            # ONE plan per rank -- the job's own GroupAggregate -- whose table slot of a group is the keys' mixed-radix number on every rank
            backend = PlanDenseBackend(ctx, ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), group_spec(ss), None, group_child(ss, view)))
            job = DenseShardedGroupAggregate(backend)
            try:
                job.setup(view)
                job.first, job.capacity = backend.plan, None
                job.result = lambda: (backend.plan, None)
            except ss.SupersonicException as e:      # the same verdict on every rank (same plan, same agreed ranges): the image exchange
                sys.stderr.write("[bench] dense slots do not fit this job (%s): key-range exchange\n" % e)
                args.exchange, job = "key_range", None
        if job is None:
            job = DeviceShardedGroupAggregate(ctx, ["k1", "k2"], group_spec(ss), group_child(ss, view), exchange=args.exchange)
        plan = job.first
    else:
        if (args.query in ("sort", "filter_mat")) and distributed:
            raise SystemExit("--query %s is a single-GPU line (SURVEY 8(e): no single-exchange shape)" % args.query)
        plan = ss.Plan(build_group_plan(ss, view) if group else build_sort_plan(ss, view) if args.query == "sort"
                       else build_filter_mat_plan(ss, view) if args.query == "filter_mat" else build_plan(ss, view), ctx)

    seg_tensors = None

    def step(view=view, row_offset=row_offset):
        nonlocal seg_tensors
        if job is not None:
            job.step(view)
            return
        if not distributed:
            plan.run(view)
            return
        segs = plan.run_partial(view, row_offset)
        if seg_tensors is None:
            # the partial-aggregate state is ONE contiguous device buffer of 8 arrays x n_slots
            # 64-bit words (a few hundred bytes): view it as int64 words without a copy
            total = sum(count for (_p, count, _d, _r) in segs)
            assert all(segs[i][0] + segs[i][1] * 8 == segs[i + 1][0] for i in range(len(segs) - 1))
            state = torch.as_tensor(_DevPtr(segs[0][0], total, "<i8"), device=device)
            gathered = torch.empty((world, total), dtype=torch.int64, device=device)
            seg_tensors = (state, gathered)
        state, gathered = seg_tensors
        # ONE collective over RCCL / xGMI (all-gather of the tiny state), then ONE kernel folds the
        # `world` images segment by segment with each segment's operator and emits the row (ssgpu_plan_fold_finalize)
        dist.all_gather_into_tensor(gathered, state)
        plan.fold_finalize(gathered.data_ptr(), world)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)

    # the library records HIP events around the stage's kernels of every run on its launch stream and keeps
    # the last 256 pairs: the timed steps stay asynchronous and their kernel durations are read afterwards
    ctx.set_option("profile", 0 if args.no_events_in_loop else 1)
    ctx.set_option("profile_total", 0)      # only the pair around the stage's kernels, not the whole-run pair
    # set-up (not warm-up steps): buffers are allocated, the stage kernels compiled, and a GroupAggregate walks through its
    # fed-back execution shapes -- until two consecutive steps compile nothing, allocate nothing and keep their shape
    def settled_state():
        st = ss.memory_stats()
        return (st["device_bytes"], st["rtc_compilations"], json.dumps((job.first if job is not None else plan).stage_info(), sort_keys=True))
    step()
    barrier()
    for _ in range(8):
        before = settled_state()
        step()
        barrier()
        if settled_state() == before:
            break
    for _ in range(args.warmup):
        step()
    if job is not None:
        while not job.check():               # set-up: the image capacity every rank agreed on holds every partial table
            step()
    barrier()
    trace = [] if os.environ.get("SSGPU_BENCH_TRACE") else None    # development: host time of every step call
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if trace is not None:
            trace.append(time.perf_counter())
    barrier()
    elapsed = time.perf_counter() - t0
    if trace:
        d = [(b - a) * 1e3 for a, b in zip([t0] + trace[:-1], trace)]
        worst = sorted(range(len(d)), key=lambda i: -d[i])[:5]
        sys.stderr.write("[bench trace] steps %d, median %.2f ms, slowest: %s\n" % (len(d), sorted(d)[len(d) // 2], ", ".join("#%d %.1f ms" % (i, d[i]) for i in worst)))
    # the per-kernel clock: the HIP events the library recorded around the stage's kernels of the TIMED
    # steps (the most recent min(K, 256) of them); without them, a dedicated loop after the timed region
    kernel_ms = [] if args.no_events_in_loop else plan.recent_kernel_ms(256)[-args.steps:]
    if not kernel_ms:
        ctx.set_option("profile", 1)
        for _ in range(min(args.steps, 10)):
            step()
            torch.cuda.synchronize(device)
            kernel_ms.append(plan.counters().dominant_ms)
    counters = plan.counters()
    if job is not None and not job.check():
        raise SystemExit("a partial group table outgrew the agreed image capacity inside the timed region")

    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Both scaling regimes in ONE line: the timed region above is the regime --scaling names; the other one runs
    # here over the same resident columns -- strong scaling of a weak run = the job of --rows rows IN TOTAL, every rank taking
    # the first rows / N of its columns -- with the same barrier + max-over-ranks clock.  The single-GPU time of the same
    # plan over all --rows rows, measured on this very box, is reported next to them (the reader divides; no efficiency is claimed here).
    regimes = None
    if distributed and not args.no_regimes and args.scaling == "weak":
        def timed(fn, k):
            barrier()
            t0 = time.perf_counter()
            for _ in range(k):
                fn()
            barrier()
            dt = time.perf_counter() - t0
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item()) / k * 1e3
        k = max(10, min(args.steps, 100))
        regimes = {"weak": {"rows_per_gpu": rows, "rows_total": rows * world, "ms_per_step": elapsed / args.steps * 1e3, "value": total_rows_per_step * args.steps / elapsed}}
        # the one-GPU reference: the same plan (the shard plan of a GroupAggregate job) over all --rows rows, no exchange
        single = (lambda: job.first.run(view)) if job is not None else (lambda: plan.run(view))
        for _ in range(3):
            single()
        n1_ms = timed(single, k)
        regimes["weak"]["n1_ms_per_step"] = n1_ms
        share = rows // world
        if share > 0:
            sview = ss.DeviceView(schema, [(t.data_ptr(), 0) for t in cols], share)
            strong_step = lambda: step(sview, rank * share)   # noqa: E731
            for _ in range(5):
                strong_step()
            if job is not None:
                while not job.check():
                    strong_step()
            ms = timed(strong_step, k)
            regimes["strong"] = {"rows_total": share * world, "rows_per_gpu": share, "ms_per_step": ms, "value": share * world / (ms / 1e3),
                                 "n1_ms_per_step": n1_ms,
                                 "note": "every rank takes the first rows / N of its resident columns: a job of --rows rows in total"}
            for _ in range(3):                     # leave the plans' results as the primary regime left them (the checks below read them)
                step()
            if job is not None:
                job.check()
            barrier()
    groups_total = None
    if group:
        groups_total = (job.result()[0] if job is not None else plan).fetch().row_count()
        if job is not None and args.exchange in ("key_range", "dense") and world > 1:     # every rank holds the groups it owns: the table's size is their sum
            g = torch.tensor([groups_total], device=device, dtype=torch.int64)
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            groups_total = int(g.item())

    group_checked = None
    if group and job is None:
        group_checked, _rows = check_group_result(torch, plan, cols, device, GROUP_FILTER)
        if not all(group_checked.values()):
            raise SystemExit("bench.py --query %s: the device result failed its check: %r" % (QUERY_NAME, group_checked))
    if args.query in ("sort", "filter_mat"):
        checked, out_rows = check_rows_result(torch, plan, cols, device, args.query)
        if not all(checked.values()):
            raise SystemExit("bench.py --query %s: the device result failed its check: %r" % (args.query, checked))
        result = None
    else:
        result = (job.result()[0] if job is not None else plan).fetch()
    if rank == 0:
        value = total_rows_per_step * args.steps / elapsed
        avg_kernel_s = (sum(kernel_ms) / len(kernel_ms)) / 1e3
        alg_bytes = counters.algorithmic_bytes  # bytes/row of the staged input columns x rows of one launch
        if group:
            workload = ("%s: GroupAggregate(k1,k2; SUM/MIN/MAX x d0..d3)%s over a device-resident %d-row x 7-col "
                        "block per GPU (~1e5 groups)" % ("Q-GROUP-F" if GROUP_FILTER else "Q-GROUP", " o Filter(a>499)" if GROUP_FILTER else "", rows))
            par = ("row-range shards x%d: per-shard GroupAggregate into a dense-slot table, ONE RCCL all-to-all of slot slices, element-wise fold (no merge plan)" % world
                   if distributed and args.exchange == "dense" else
                   "row-range shards x%d: per-shard GroupAggregate, ONE RCCL %s of the packed partial tables, merge plan" % (
                       world, "all-to-all by key range (every rank merges the groups it owns)" if args.exchange == "key_range" else "all-gather")
                   if distributed else "single GPU")
            kernel = "group stage (partition scatter + per-partition aggregation kernels)"
            result_row = dict({"groups": groups_total}, **(group_checked or {}))
        elif args.query == "sort":
            workload = "Q-SORT: Sort(d ASC, all 8 columns) of a device-resident %d-row x 8-col block" % rows
            par, kernel = "single GPU", "sort stage (key load + histograms, radix passes, tie fix-up, record pack + gather)"
            alg_bytes = 128 * rows               # every column read once and written once
            result_row = dict(rows=out_rows, **checked)
        elif args.query == "filter_mat":
            workload = "Q-FILTER-mat: Filter(a>499, all 8 columns) of a device-resident %d-row x 8-col block, survivors materialised" % rows
            par, kernel = "single GPU", "filter stage (count pass + scan + compacting store pass)"
            alg_bytes = 64 * rows + 64 * out_rows   # 64 B/row read + 64 B per surviving row written
            result_row = dict(rows=out_rows, **checked)
        else:
            workload = ("Q-FPA-wide: SUM(a+b),COUNT(*),SUM(c),MIN(d),MAX(d0),SUM(d1),SUM(d2*d3) WHERE a>499 "
                        "over a device-resident %d-row x 8-col block per GPU" % rows)
            par = ("row-range shards x%d, one RCCL all-gather of the partial-aggregate state + one fold kernel per step" % world
                   if distributed else "single GPU")
            kernel = "ssgpu_pipeline_kernel"
            result_row = [result.column(i).data[0].item() for i in range(result.column_count())]
        achieved = alg_bytes / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        line = {
            "metric": metric_name(args.query),
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "int64/f64", "data": "synthetic",
            "config": {"workload": workload, "rows_per_gpu": rows, "parallelism": par,
                       "input_layout": ("arena, column pitch 2 MiB-rounded + %d B (--stagger)" % args.stagger if args.stagger >= 0 else
                                        "device Block laid out by the library (ssgpu_block_create: one arena, column bases skewed by 512 B)" if args.layout == "library"
                                        else "one torch allocation per column"),
                       "tile_rows": counters.tile_rows, "grid": counters.grid, "lds_bytes": counters.lds_bytes,
                       "specialized_stages": plan.specialized()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(QUERY_NAME, alg_bytes),
                         "traffic_source": pmc_source(QUERY_NAME, alg_bytes),    # `traffic` is read from this committed PMC pass, not counted in this run
                         "traffic_measured": False,                               # (PMC counters need rocprofv3 around the process: tools/profile_round5.sh)
                         "kernel": kernel, "kernel_ms": avg_kernel_s * 1e3,
                         "algorithmic_bytes_per_row": alg_bytes / max(rows, 1)},
            "result_row": result_row,
        }
        if share_gpu:
            line["config"]["development_mode"] = "SSGPU_BENCH_SHARE_GPU: all ranks on ONE GPU, collectives through the host over gloo -- the timings are not a measurement"
        if job is not None:
            line["config"]["collectives_per_step"] = job.collectives
            line["config"]["image_capacity_rows"] = job.capacity
            line["config"]["exchange"] = args.exchange
            if args.exchange == "dense":
                line["config"]["dense_layout"] = job.layout
        if regimes is not None:
            line["regimes"] = regimes
        if world == 1 and not distributed and args.query == "wide" and QUERY_NAME == "wide" and not args.no_configs:
            # the other BASELINE configs on the SAME box, after the headline's timed region (its value, ms_per_step and roofline are
            # untouched): config #3 (group3), config #4's per-GPU query (group), config #5 (sort), the materialising Filter
            line["configs"] = {}
            for q in ("group3", "group", "sort", "filter_mat"):
                try:
                    line["configs"][q] = measure_config(ss, torch, ctx, device, q, rows, args.config_steps, wide_cols=cols)
                    if not args.no_cpu_baseline:       # the oracle on the same query, a bounded sample (about 3 s of CPU each)
                        line["configs"][q]["cpu_baseline"] = config_cpu_baseline(ss, q)
                except Exception as e:   # noqa: BLE001 -- the headline line must survive a failing extra
                    line["configs"][q] = dict(line["configs"].get(q) or {}, error="%s: %s" % (type(e).__name__, e))
        if world == 1 and args.extras and args.query == "wide":
            line["extras"] = extras(ss, torch, ctx, device, rows, cols, view)
        if world == 1 and not distributed and args.query == "wide" and QUERY_NAME == "wide" and not os.environ.get("SSGPU_BENCH_CHILD"):
            # the same plan under the LIBRARY'S DEFAULT options (a context nobody tuned: specialize = 3 -- the compiled kernel where one
            # exists, here the one the headline compiled --, lazy_feedback = 0 -- every run settled before ssgpu_plan_run returns):
            # what a caller who sets nothing gets, next to the tuned headline (round-5 advice)
            try:
                dctx = ss.Context(local_rank)
                dctx.set_stream(stream.cuda_stream)
                dplan = ss.Plan(build_plan(ss, view), dctx)
                for _ in range(5):
                    dplan.run(view)
                torch.cuda.synchronize(device)
                k = max(10, min(args.steps, 50))
                t0 = time.perf_counter()
                for _ in range(k):
                    dplan.run(view)
                torch.cuda.synchronize(device)
                dt = time.perf_counter() - t0
                line["default_options"] = {"ms_per_step": dt / k * 1e3, "value": rows * k / dt, "steps": k, "specialized_stages": dplan.specialized(),
                                           "options": "library defaults: specialize = 3 (compiled kernel where one exists, never waits for the compiler), "
                                                      "lazy_feedback = 0 (every run settled before it returns)"}
                del dplan, dctx
            except Exception as e:   # noqa: BLE001
                line["default_options"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if (world == 1 and not distributed and not args.no_traffic and not os.environ.get("SSGPU_BENCH_CHILD")
                and not any(k.startswith("ROCP") for k in os.environ)):      # (not when this process itself runs under rocprofv3)
            # this command's own counters (the timed region is over; the children are separate processes on the same GPU)
            measured, detail = measure_traffic(QUERY_NAME, rows, alg_bytes)
            if measured is not None:
                line["roofline"].update({"traffic": measured, "traffic_measured": True, "traffic_over_algorithmic": measured / max(alg_bytes, 1),
                                         "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE around child runs of this command (3 steps each), "
                                                           "FETCH_SIZE x 2 for streaming reads (gfx950), counters in KiB", "traffic_kernels": detail})
            else:
                line["roofline"]["traffic_note"] = "not counted in this run (%s): read from the committed pass" % detail
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ss, args.query, args.cpu_sample_rows or {"wide": 16_000_000, "filter_mat": 16_000_000}.get(args.query, 4_000_000))
        # RCCL prints its version banner through C stdio (still buffered when stdout is a file):
        # drain it first so that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
