#!/bin/bash
# Raw PMC counters per kernel of one bench.py invocation, one rocprofv3 pass per counter group.
# Usage (GPU box): tools/pmc_raw.sh <tag> "<C1 C2>;<C3 C4>" <bench.py args...>   -> gpurun_out/pmcraw_<tag>/counters.json
tag=$1; groups=$2; shift; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcraw_$tag
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra GROUPS_ <<< "$groups"
i=0
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o p -- python $REPO/bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline > $OUT/g$i.log 2>&1 || echo "pass $i ($grp) failed" >> $OUT/failed.txt
done
python3 - "$OUT" "$tag" "$*" <<'PY'
import csv, glob, json, sys, collections
out, tag, args = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "ssgpu" in row["Kernel_Name"]:
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {"tag": tag, "command": "python bench.py %s --steps 5 --warmup 2 --no-cpu-baseline" % args, "kernels": {}}
for k, cs in acc.items():
    for c, v in cs.items():
        v = sorted(v)[len(v) // 4:]
        res["kernels"].setdefault(k, {})[c] = sum(v) / len(v)
json.dump(res, open(out + "/counters.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(res["kernels"].items()):
    print("%-56s %s" % (k[:56], "  ".join("%s=%.4g" % kv for kv in sorted(v.items()))))
PY
