for q in group3 group; do
  echo "== $q"; bash tools/kstats.sh pa python bench.py --query $q --no-cpu-baseline --steps 10 --warmup 3 2>&1 | grep "part_agg\|pipeline_kernel\|value" | cut -c1-140
done
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_double_sum_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -n 4 -k "group or sum" 2>&1 | tail -2
