#!/bin/bash
# materialising Filter: tile size x workgroups per CU (one box, one process per point)
cd "$(dirname "$0")/.."
out=gpurun_out/r05_filter_sweep.txt
: > $out
for cfg in "512 2" "512 3" "512 4" "512 6" "1024 2" "1024 3" "2048 1" "2048 2"; do
  set -- $cfg
  line=$(timeout 120 python bench.py --query filter_mat --steps 40 --warmup 10 --no-cpu-baseline --tile-rows $1 --opts wgs_per_cu=$2 2>/dev/null | tail -1)
  echo "tile $1 wgs $2: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["config"].get("grid"), d["config"].get("lds_bytes"), d["result_row"])')" >> $out
done
cat $out
