#!/bin/bash
# round 6, call 12: the whole GPU suite + smoke + the driver-style bench line on the current tree
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06_suite2.log 2>&1
tail -6 gpurun_out/r06_suite2.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r06_smoke2.log 2>&1
tail -3 gpurun_out/r06_smoke2.log
( time python bench.py ) > gpurun_out/r06_bench_default3.json 2> gpurun_out/r06_bench_default3.err
tail -c 300 gpurun_out/r06_bench_default3.err
python - <<'PY'
import json
s=[l for l in open('gpurun_out/r06_bench_default3.json').read().splitlines() if l.startswith('{')][-1]
d=json.loads(s)
print('headline', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic_measured'), d['roofline'].get('traffic'))
print('cpu', json.dumps(d['cpu_baseline'])[:600])
for q,c in d['configs'].items(): print(q, c.get('ms_per_step'), c.get('frac'), c.get('error'))
PY
