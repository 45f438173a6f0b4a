#!/bin/bash
# round 6, call 11: chunked execution of row-local and GroupAggregate plans -- parity, the facade, PCIe-inclusive rates
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_chunked_gpu.py tests/test_cursor_contract_gpu.py tests/test_cpp_facade.py tests/test_dense_gpu.py -m gpu -x -q --timeout 300 ) > gpurun_out/r06_call11_tests.log 2>&1
tail -15 gpurun_out/r06_call11_tests.log
( time timeout 600 python tools/host_staging_bench.py 50000000 ) > gpurun_out/r06_host_staging.json 2> gpurun_out/r06_host_staging.err
tail -3 gpurun_out/r06_host_staging.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_host_staging.json').read().strip().splitlines()[0])
for k in ('runs',):
    for label, r in d[k].items(): print('scalar', label, round(r['GB_per_s'], 1), 'GB/s', r['same_row'])
for q in ('filter_mat', 'group3'):
    for label, r in d[q]['runs'].items(): print(q, label, round(r['GB_per_s'], 1), 'GB/s', r['rows_out'], r['same_result'])
PY
