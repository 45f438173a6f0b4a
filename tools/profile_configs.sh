#!/bin/bash
# rocprofv3 kernel statistics for the other BASELINE configs (#3 GroupAggregate shape, #5 Sort) and
# the materialising Filter, through tools/perf_sweep.py.  Outputs under gpurun_out/prof_cfg/.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_cfg
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for q in group2 sort sort_key filter_mat group_small group_tiny join; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$q -o $q -- python $REPO/tools/perf_sweep.py --queries $q --tiles 0 --reps 7 --opts specialize=1 > $OUT/$q.log 2>&1
  f=$(find $OUT/$q -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] || { echo "$q: no kernel stats"; continue; }
  python - "$f" "$OUT/${q}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
out = [rows[0]] + [[r[0][:110]] + r[1:] for r in rows[1:] if "ssgpu" in r[0]]
csv.writer(open(sys.argv[2], "w")).writerows(out)
PY
  grep -v amdgpu.ids $OUT/$q.log | grep "^$q\|^   \[" | tail -2
done
