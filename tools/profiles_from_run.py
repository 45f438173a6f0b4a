#!/usr/bin/env python3
"""Turns gpurun_out/prof_r03 (tools/profile_round3.sh, collected on the GPU box) into the committed files under profiles/:
r03_bench_<query>_line.json, r03_<query>_kernel_stats.csv (the library's kernels only), r03_pmc_<query>.json (HBM traffic
per launch of the stage's kernels: 2 x FETCH_SIZE + WRITE_SIZE, factors from the calibration pass) and r03_summary.md."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", os.environ.get("SSGPU_PROFILE_SRC", "prof_r03"))
DST = os.path.join(ROOT, "profiles")
ROUND = os.environ.get("SSGPU_PROFILE_TAG", "r03")   # r03b: the second collection of round 3 (tools/profile_refresh.sh, another box)
PMC_ROUND = os.environ.get("SSGPU_PROFILE_PMC_TAG", "r03")
# the kernels of the timed stage, per query (substrings of rocprof's kernel names)
STAGE = {"wide": ["ssgpu_pipeline_kernel", "ssgpu_finish_slots", "ssgpu_emit_scalar"],
         "group3": ["ssgpu_part_scatter_plain", "ssgpu_part_agg", "ssgpu_group_extract", "ssgpu_group_count", "ssgpu_scan_counts", "ssgpu_fill", "ssgpu_group_init"],
         "group": ["ssgpu_part_scatter_plain", "ssgpu_part_agg", "ssgpu_group_extract", "ssgpu_group_count", "ssgpu_scan_counts", "ssgpu_fill", "ssgpu_group_init"],
         "sort": ["ssgpu_sort_"], "filter_mat": ["ssgpu_pipeline_kernel", "ssgpu_scan_counts"]}


def counters(path, name):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == name and "ssgpu" in row["Kernel_Name"]:
                acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return acc


def steady(values, launches_per_step):
    """Average over the launches of the steady state: the last steps of the run (the first ones are set-up / other shapes)."""
    keep = values[-max(launches_per_step * 3, 1):]
    return sum(keep) / len(keep)


def main():
    summary = ["# %s profiles (MI355X, 100 M rows, `tools/profile_round%s.sh`%s)" % (ROUND, ROUND[2] if len(ROUND) > 2 else "3", " -- second collection, `tools/profile_refresh.sh`" if ROUND == "r03b" else ""), "",
               "| query | kernel ms (bench line) | frac of 8 TB/s | algorithmic B/row | HBM traffic / algorithmic | interpreted kernel ms |", "|---|---|---|---|---|---|"]
    factors = {}
    cal = os.path.join(SRC, "pmc_calibration.json")
    if os.path.exists(cal):
        shutil.copy(cal, os.path.join(DST, ROUND + "_pmc_calibration.json"))
        factors = json.load(open(cal))
    for q in ("wide", "group3", "group", "sort", "filter_mat"):
        d = os.path.join(SRC, q)
        if not os.path.isdir(d):
            continue
        line = json.loads(open(os.path.join(d, "line.json")).read().strip().splitlines()[-1])
        with open(os.path.join(DST, "%s_bench_%s_line.json" % (ROUND, q)), "w") as f:
            f.write(json.dumps(line) + "\n")
        interp = None
        ip = os.path.join(d, "line_interpreted.json")
        if os.path.exists(ip) and open(ip).read().strip():
            interp = json.loads(open(ip).read().strip().splitlines()[-1])
            with open(os.path.join(DST, "%s_bench_%s_interpreted_line.json" % (ROUND, q)), "w") as f:
                f.write(json.dumps(interp) + "\n")
        ks = os.path.join(d, "kernel_stats.csv")
        if os.path.exists(ks):
            rows = list(csv.reader(open(ks)))
            keep = [rows[0]] + [r for r in rows[1:] if "ssgpu" in r[0]]
            with open(os.path.join(DST, "%s_%s_kernel_stats.csv" % (ROUND, q)), "w", newline="") as f:
                csv.writer(f).writerows(keep)
        traffic = None
        if os.path.exists(os.path.join(d, "FETCH_SIZE.csv")) and os.path.exists(os.path.join(d, "WRITE_SIZE.csv")):
            fetch, write = counters(os.path.join(d, "FETCH_SIZE.csv"), "FETCH_SIZE"), counters(os.path.join(d, "WRITE_SIZE.csv"), "WRITE_SIZE")
            per_kernel, total = {}, 0.0
            for k in sorted(set(fetch) | set(write)):
                if not any(sub in k for sub in STAGE[q]):
                    continue
                n_per_step = 4 if "onesweep" in k else 1
                fk = steady(fetch.get(k, [0.0]), n_per_step) * n_per_step
                wk = steady(write.get(k, [0.0]), n_per_step) * n_per_step
                per_kernel[k] = {"FETCH_SIZE_KiB_per_step": fk, "WRITE_SIZE_KiB_per_step": wk, "launches_per_step": n_per_step}
                # FETCH_SIZE tallies every request at 64 bytes: 128-byte streaming requests are under-counted by 2, the record gather's
                # random 64-byte requests are counted in full (profiles/r04_pmc_gather_calibration.json)
                ff = 1.0 if "sort_gather_rec" in k else 2.0
                per_kernel[k]["fetch_factor"] = ff
                total += fk * 1024 * ff + wk * 1024
            alg = line["roofline"]["algorithmic_bytes_per_row"] * line["config"]["rows_per_gpu"]
            traffic = total
            j = {"round": ROUND, "query": q, "command": "python bench.py --query %s --steps 5 --warmup 2 --no-cpu-baseline" % q,
                 "kernels": per_kernel, "correction": "FETCH_SIZE x 2 for streaming kernels (gfx950: the counter tallies 128-byte requests at 64 bytes; measured 2.000 for 4 / 8 / 16 B-per-lane "
                                                      "reads and 40-byte records), x 1 for the Sort's record gather (random 64-byte reads are 64-byte requests: measured 1.000, "
                                                      "profiles/r04_pmc_gather_calibration.json), WRITE_SIZE x 1 (measured 1.000)",
                 "traffic_bytes_per_launch": total, "algorithmic_bytes_per_launch": int(alg), "traffic_over_algorithmic": total / alg if alg else None}
            with open(os.path.join(DST, "%s_pmc_%s.json" % (ROUND, q)), "w") as f:
                json.dump(j, f, indent=1, sort_keys=True)
        if traffic is None and os.path.exists(os.path.join(DST, "%s_pmc_%s.json" % (PMC_ROUND, q))):
            # a collection without counter passes (tools/profile_refresh.sh): the committed counters of this round stand
            traffic = json.load(open(os.path.join(DST, "%s_pmc_%s.json" % (PMC_ROUND, q))))["traffic_bytes_per_launch"]
        r = line["roofline"]
        summary.append("| %s | %.3f | %.3f | %.0f | %s | %s |" % (q, r["kernel_ms"], r["frac"], r["algorithmic_bytes_per_row"],
                       "%.2f" % (traffic / (r["algorithmic_bytes_per_row"] * line["config"]["rows_per_gpu"])) if traffic else "-",
                       "%.3f" % interp["roofline"]["kernel_ms"] if interp else "-"))
    gs = os.path.join(SRC, "group_small")
    if os.path.isdir(gs):
        ks = os.path.join(gs, "kernel_stats.csv")
        if os.path.exists(ks):
            rows = list(csv.reader(open(ks)))
            with open(os.path.join(DST, "%s_group_small_kernel_stats.csv" % ROUND), "w", newline="") as f:
                csv.writer(f).writerows([rows[0]] + [r for r in rows[1:] if "ssgpu" in r[0]])
        summary += ["", "GroupAggregate(a; SUM / MIN / MAX x d0..d3), 1000 groups, 100 M rows (`tools/perf_sweep.py --queries group_small`, 40 B/row):", ""]
        for name, label in (("sweep_specialize1.txt", "resident form, specialised"), ("sweep_specialize0.txt", "resident form, generic kernel"),
                            ("sweep_records_through_memory.txt", "slab form (group_resident=0), specialised")):
            fp = os.path.join(gs, name)
            if os.path.exists(fp):
                for ln in open(fp).read().splitlines():
                    if ln.startswith("group_small"):
                        summary.append("* %s: `%s`" % (label, " ".join(ln.split())))
    with open(os.path.join(DST, ROUND + "_summary.md"), "w") as f:
        f.write("\n".join(summary) + "\n")
    print("\n".join(summary))


if __name__ == "__main__":
    sys.exit(main())
