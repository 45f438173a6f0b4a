#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r06_layout_ab.txt
: > $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.4f kernel_ms %.4f frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))'
b() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic --no-configs "$@" 2>/dev/null | tail -1 | python -c "$pick"; }
( time python -m pytest tests/test_best_effort_gpu.py tests/test_seams_gpu.py tests/test_file_format.py tests/test_cursor_contract_gpu.py tests/test_00_configs_gpu.py -m gpu -x -q ) > gpurun_out/r06_call4_tests.log 2>&1
tail -4 gpurun_out/r06_call4_tests.log
for rep in 1 2 3; do
  for q in wide filter_mat group3 group sort; do
    echo "$q layout torch   rep $rep: $(b --query $q --layout torch)" >> $out
    echo "$q layout library rep $rep: $(b --query $q --layout library)" >> $out
  done
done
cat $out
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r06_bench_default2.json 2> gpurun_out/r06_bench_default2.err
tail -c 400 gpurun_out/r06_bench_default2.err
python - <<'PY'
import json
s=[l for l in open('gpurun_out/r06_bench_default2.json').read().splitlines() if l.startswith('{')][-1]
d=json.loads(s)
print('headline', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic_measured'), d.get('default_options'))
print('cpu', json.dumps(d['cpu_baseline'])[:1200])
for q,c in d['configs'].items(): print(q, c.get('ms_per_step'), c.get('frac'), c.get('error'), json.dumps(c.get('cpu_baseline',{}).get('threads'))[:400])
PY
