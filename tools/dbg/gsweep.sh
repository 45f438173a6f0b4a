for f in "" "-DSS_PART_NT_STORES" ""; do
  echo "== flags: $f"; SSGPU_RTC_FLAGS="$f" python bench.py --query group3 --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms'], round(j['roofline']['frac'],3), j['config'].get('specialized_stages'))"
done
