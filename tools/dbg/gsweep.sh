for q in wide group3 group; do for o in "" "--no-specialize"; do
  echo "== $q $o"; python bench.py --query $q --no-cpu-baseline --steps 50 --warmup 5 $o 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms'], round(j['roofline']['frac'],3), j['config'].get('specialized_stages'))"
done; done
for q in filter_mat group_tiny group_small join narrow sum1; do for o in specialize=1 specialize=0; do
  echo "== $q $o"; python tools/perf_sweep.py --queries $q --tiles 0 --reps 5 --opts $o 2>&1 | grep "^$q" | tail -1 | cut -c1-150
done; done
