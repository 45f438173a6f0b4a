for lib in libssgpu.so libssgpu_v1.so; do for q in group3 group; do
  echo "== $lib $q"; SSGPU_LIB=/root/repo/supersonic_amd/lib/$lib python bench.py --query $q --no-cpu-baseline --steps 20 --warmup 3 --opts debug_timing=1 > /tmp/o.txt 2>&1; grep "group stage: part" /tmp/o.txt | tail -1 | cut -c1-200; tail -1 /tmp/o.txt | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['config'].get('tile_rows'), j['config'].get('grid'), j['config'].get('lds_bytes'))"
done; done
