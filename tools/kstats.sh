#!/bin/bash
# Development aid: rocprofv3 kernel statistics of one command; prints the ssgpu kernels' rows.
# Usage: tools/kstats.sh <tag> <command...>
tag=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/kstats/$tag
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- "$@" > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
echo "== $tag: $*"
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ssgpu" in r["Name"]:
        print("%-70s calls %4s avg %10.1f us min %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
grep -h "^{" $OUT/log.txt | tail -1 | python -c "import sys,json
for l in sys.stdin:
    j=json.loads(l); print('   value %.3g rows/s  ms_per_step %.3f  kernel_ms %.3f frac %.3f' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac']))" 2>/dev/null
