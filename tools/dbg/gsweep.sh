for o in debug_timing=0 part_rec_align=64 part_rec_align=128; do
  echo "== $o"; bash tools/kstats.sh pa python bench.py --query group3 --no-cpu-baseline --steps 10 --warmup 3 --opts $o 2>&1 | grep "part_agg\|pipeline_kernel" | cut -c1-140
done
