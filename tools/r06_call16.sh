#!/bin/bash
# round 6, call 16: what the partition aggregation spends its time on (debug switch 1 = records loaded, no aggregation), partition counts
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/kstats.sh r06_aggdbg1_group3 python bench.py --query group3 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-configs --opts part_agg_debug=1
bash tools/kstats.sh r06_np512_group3 python bench.py --query group3 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-configs --opts dense_parts=512
bash tools/kstats.sh r06_np384_group3 python bench.py --query group3 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-configs --opts dense_parts=384
bash tools/kstats.sh r06_np512_group python bench.py --query group --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-configs --opts dense_parts=512
bash tools/kstats.sh r06_np256_group3 python bench.py --query group3 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-configs
