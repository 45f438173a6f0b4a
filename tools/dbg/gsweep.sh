for o in debug_timing=0 part_agg_debug=1 part_agg_debug=2 part_agg_debug=3; do
  echo "== $o"; bash tools/kstats.sh pa python bench.py --query group3 --no-cpu-baseline --steps 10 --warmup 3 --opts $o 2>&1 | grep "part_agg\|pipeline_kernel" | cut -c1-140
done
