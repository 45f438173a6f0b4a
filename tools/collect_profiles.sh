#!/bin/bash
# Copies the judged summaries of a GPU run (tools/profile_round.sh + tools/profile_configs.sh + the bench lines, all
# written under gpurun_out/ on the box) into profiles/ under the round's prefix.  Run on the build host after gpurun.
# Usage: tools/collect_profiles.sh r02
R=${1:-r02}
cd "$(dirname "$0")/.."
S=gpurun_out/prof_$R
cp $S/bench_line.json profiles/${R}_bench_line.json
cp $S/pmc.json profiles/${R}_pmc.json
python - "$(find $S/stats -name '*kernel_stats.csv' | head -1)" profiles/${R}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
csv.writer(open(sys.argv[2], "w")).writerows([rows[0]] + [[r[0][:120]] + r[1:] for r in rows[1:]])
PY
cp "$(find $S/stats -name '*domain_stats.csv' | head -1)" profiles/${R}_domain_stats.csv
for q in group2 sort sort_key filter_mat group_small group_tiny join; do
  [ -f gpurun_out/prof_cfg/${q}_kernel_stats.csv ] && cp gpurun_out/prof_cfg/${q}_kernel_stats.csv profiles/${R}_${q}_kernel_stats.csv
done
for q in group3 group extras interp; do
  [ -f gpurun_out/bench_$q.json ] && grep "^{" gpurun_out/bench_$q.json | tail -1 > profiles/${R}_bench_${q}_line.json
done
[ -f gpurun_out/double_sum_ulp.json ] && cp gpurun_out/double_sum_ulp.json profiles/${R}_double_sum_ulp.json
ls profiles/${R}_*
