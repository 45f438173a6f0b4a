#!/bin/bash
# Per-kernel FETCH_SIZE / WRITE_SIZE (KiB per launch, raw counters, separate passes) of one bench.py invocation.
# Usage (GPU box): tools/pmc_kernels.sh <tag> <bench.py args...>   -> gpurun_out/pmc_<tag>/counters.json
tag=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$tag
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o p -- python $REPO/bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline > $OUT/$c.log 2>&1
done
python - "$OUT" "$tag" "$*" <<'PY'
import csv, glob, json, sys, collections
out, tag, args = sys.argv[1], sys.argv[2], sys.argv[3]
res = {"tag": tag, "command": "python bench.py %s --steps 5 --warmup 2 --no-cpu-baseline" % args, "kernels": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + "/%s/**/*counter_collection.csv" % c, recursive=True)
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] == c and "ssgpu" in row["Kernel_Name"]:
            acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        v = sorted(v)[len(v) // 4:]          # the first launches of a plan are its set-up runs (other shapes): keep the steady ones
        res["kernels"].setdefault(k, {})[c + "_KiB_per_launch"] = sum(v) / len(v)
        res["kernels"][k]["launches"] = len(v)
json.dump(res, open(out + "/counters.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(res["kernels"].items()):
    print("%-60s fetch %14.1f KiB  write %14.1f KiB  (n=%d)" % (k[:60], v.get("FETCH_SIZE_KiB_per_launch", 0), v.get("WRITE_SIZE_KiB_per_launch", 0), v["launches"]))
PY
