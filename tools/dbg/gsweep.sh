for f in "" "-DSSGPU_RTC_DYNAMIC_STAGING" "-DSSGPU_RTC_LAUNDER" "-DSSGPU_RTC_DYNAMIC_STAGING -DSSGPU_RTC_LAUNDER" ""; do
  echo "== flags: $f"; SSGPU_RTC_FLAGS="$f" python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('wide', j['ms_per_step'], j['roofline']['kernel_ms'], round(j['roofline']['frac'],3), j['config'].get('specialized_stages'))"
done
python bench.py --no-cpu-baseline --steps 100 --warmup 10 --no-specialize 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('interp', j['ms_per_step'], j['roofline']['kernel_ms'], round(j['roofline']['frac'],3))"
