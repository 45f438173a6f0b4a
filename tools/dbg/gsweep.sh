for o in specialize=1 specialize=1,part_agg_debug=2 specialize=1,part_agg_debug=1; do
echo "== $o"; bash tools/kstats.sh gs python tools/perf_sweep.py --queries group_small --tiles 0 --reps 5 --opts $o 2>&1 | grep "part_agg\|pipeline_kernel" | cut -c1-140
done
