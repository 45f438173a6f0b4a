#!/bin/bash
# round 5: the GroupAggregate lines with and without dense slots, on ONE box (A/B), and the sharded step at the 8-GPU shard size
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
B="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-configs"
run() { name=$1; shift; timeout 300 $B "$@" > gpurun_out/r05_bench_$name.json 2> gpurun_out/r05_bench_$name.err; tail -c 600 gpurun_out/r05_bench_$name.json | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(l['ms_per_step'],3), 'ms/step kernel', round(l['roofline']['kernel_ms'],3), 'frac', round(l['roofline']['frac'],3), l['config'].get('exchange',''))" 2>/dev/null || tail -3 gpurun_out/r05_bench_$name.err; }
run group3_dense --query group3
run group3_hashed --query group3 --opts group_dense=0
run group3_dense256 --query group3 --opts dense_parts=256
run group3_dense1024 --query group3 --opts dense_parts=1024
run group_dense --query group
run group_hashed --query group --opts group_dense=0
run group_12m5_plain_dense --query group --rows 12500000
run group_12m5_plain_hashed --query group --rows 12500000 --opts group_dense=0
run group_12m5_dist1_dense --query group --rows 12500000 --force-distributed --exchange dense
run group_12m5_dist1_key_range --query group --rows 12500000 --force-distributed --exchange key_range --opts group_dense=0
run group_12m5_dist1_all_gather --query group --rows 12500000 --force-distributed --exchange all_gather --opts group_dense=0
run group_100m_dist1_dense --query group --force-distributed --exchange dense
