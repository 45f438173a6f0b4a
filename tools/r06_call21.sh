#!/bin/bash
# round 6, call 21: final tree -- chunked + dense + seams tests, host staging rates, smoke, the driver-style bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_chunked_gpu.py tests/test_dense_gpu.py tests/test_seams_gpu.py tests/test_00_configs_gpu.py tests/test_cursor_contract_gpu.py -m gpu -x -q --timeout 300 ) > gpurun_out/r06_call21_tests.log 2>&1
tail -4 gpurun_out/r06_call21_tests.log
( time timeout 600 python tools/host_staging_bench.py 50000000 ) > gpurun_out/r06_host_staging.json 2> gpurun_out/r06_host_staging.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_host_staging.json').read().strip().splitlines()[0])
for label, r in d['runs'].items(): print('scalar', label, round(r['GB_per_s'], 1), 'GB/s', r['same_row'])
for q in ('filter_mat', 'group3'):
    for label, r in d[q]['runs'].items(): print(q, label, round(r['GB_per_s'], 1), 'GB/s', r['rows_out'], r['same_result'])
PY
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r06_smoke3.log 2>&1
tail -3 gpurun_out/r06_smoke3.log
( time python bench.py ) > gpurun_out/r06_bench_default4.json 2> gpurun_out/r06_bench_default4.err
tail -c 300 gpurun_out/r06_bench_default4.err
python - <<'PY'
import json
s=[l for l in open('gpurun_out/r06_bench_default4.json').read().splitlines() if l.startswith('{')][-1]
d=json.loads(s)
print('headline', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic_measured'), d['roofline'].get('traffic'), d.get('default_options'))
for q,c in d['configs'].items(): print(q, c.get('ms_per_step'), c.get('frac'), c.get('error'))
PY
