#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 2000 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06_suite_final.log 2>&1
tail -5 gpurun_out/r06_suite_final.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r06_smoke_final.log 2>&1
tail -2 gpurun_out/r06_smoke_final.log
( time python bench.py ) > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err
python - <<'PY'
import json
s=[l for l in open('gpurun_out/r06_bench_final.json').read().splitlines() if l.startswith('{')][-1]
d=json.loads(s)
print('headline', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic_measured'))
for q,c in d['configs'].items(): print(q, c.get('ms_per_step'), c.get('frac'), c.get('error'))
PY
