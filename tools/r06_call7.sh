#!/bin/bash
# round 6, call 7: launch shape of the plain partition scatter (threads x rows per thread x workgroups per CU)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r06_pscat_geom2.txt
: > $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.4f kernel_ms %.4f frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))'
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic --no-configs "$@" 2>/dev/null | tail -1 | python -c "$pick"; }
for q in group3 group; do
  for g in "0 0 0" "1024 3 1" "512 6 1" "512 3 2" "0 0 0" "1024 3 1"; do
    set -- $g
    echo "$q threads $1 rows $2 wgs $3: $(b --query $q --opts pscat_threads=$1,pscat_rows=$2,pscat_wgs=$3)" >> $out
  done
done
cat $out
bash tools/kstats.sh r06_geom_1024_3_1 python bench.py --query group3 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-configs --opts pscat_threads=1024,pscat_rows=3,pscat_wgs=1
