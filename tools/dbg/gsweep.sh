python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('wide', j['ms_per_step'], j['roofline']['kernel_ms'], round(j['roofline']['frac'],3), j['config'].get('specialized_stages'))"
for q in narrow group_tiny join filter_mat sum8 add16; do
  echo "== $q"; python tools/perf_sweep.py --queries $q --tiles 0 --reps 5 --opts specialize=1 2>&1 | grep "^$q" | tail -1 | cut -c1-150
done
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "specialized" 2>&1 | tail -2
