#!/bin/bash
# Development aid: one rocprofv3 --pmc pass per counter group over a short bench run; prints the
# per-launch average of each counter for the pipeline kernel.  Usage: tools/pmc_pass.sh "C1 C2" "C3" ...
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for grp in "$@"; do
  i=$((i+1))
  out=$REPO/gpurun_out/pmc/g$i
  rm -rf $out; mkdir -p $out
  rocprofv3 --pmc $grp --output-format csv -d $out -o p -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/log.txt 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "pipeline_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-28s %16.1f  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
done
