for o in specialize=1 specialize=1,wgs_per_cu=4 specialize=1,wgs_per_cu=2 specialize=1,tile_rows=1024 specialize=1,tile_rows=1024,wgs_per_cu=2; do
  echo "== $o"; python bench.py --no-cpu-baseline --steps 50 --warmup 5 --opts $o 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms'], round(j['roofline']['frac'],3), j['config'].get('specialized_stages'), j['config']['tile_rows'], j['config']['grid'])"
done
