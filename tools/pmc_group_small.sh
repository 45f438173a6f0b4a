#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) of the 1000-group GroupAggregate of tools/perf_sweep.py in its resident form:
# the claim to check is that 40 B/row cross HBM ONCE.  -> gpurun_out/pmc_group_small/counters.json
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_group_small
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o p -- python $REPO/tools/perf_sweep.py --queries group_small --tiles 0 --reps 8 --opts specialize=1 > $OUT/$c.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {"command": "python tools/perf_sweep.py --queries group_small --tiles 0 --reps 8 --opts specialize=1", "rows": 100000000, "algorithmic_bytes": 40 * 100000000, "kernels": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + "/%s/**/*counter_collection.csv" % c, recursive=True)
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] == c and "ssgpu" in row["Kernel_Name"]:
            acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res["kernels"].setdefault(k, {})[c + "_KiB_per_launch"] = sorted(v)[len(v) // 2]
        res["kernels"][k]["launches"] = len(v)
k = [n for n in res["kernels"] if "group_resident" in n]
if k:
    r = res["kernels"][k[0]]
    res["resident_kernel_traffic_bytes"] = r.get("FETCH_SIZE_KiB_per_launch", 0) * 1024 * 2 + r.get("WRITE_SIZE_KiB_per_launch", 0) * 1024   # gfx950: FETCH_SIZE x 2 (profiles/r03_pmc_calibration.json)
    res["traffic_over_algorithmic"] = res["resident_kernel_traffic_bytes"] / res["algorithmic_bytes"]
json.dump(res, open(out + "/counters.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))
for k, v in sorted(res["kernels"].items()):
    print("%-60s fetch %14.1f KiB  write %14.1f KiB  (n=%d)" % (k[:60], v.get("FETCH_SIZE_KiB_per_launch", 0), v.get("WRITE_SIZE_KiB_per_launch", 0), v["launches"]))
PY
