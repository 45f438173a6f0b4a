cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r04j; rm -rf $OUT; mkdir -p $OUT
for ex in all_gather key_range; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$ex -o k -- python $REPO/bench.py --query group --force-distributed --exchange $ex --rows 12500000 --steps 200 --warmup 20 --no-cpu-baseline --no-regimes > $OUT/stats_$ex.log 2>&1
f=$(find $OUT/stats_$ex -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/group_12m5_${ex}_kernel_stats.csv
rm -rf $OUT/stats_$ex
done
python - <<'PY'
import csv, os
for ex in ("all_gather","key_range"):
    rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04j/group_12m5_%s_kernel_stats.csv'%ex)))
    print(ex)
    for r in rows[:22]:
        print("  %-70s calls %6s avg %8.1f us  total %8.2f ms" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
