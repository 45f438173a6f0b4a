cd $GRAFT_REPO_ROOT
python -m pytest tests/test_parity_gpu.py -x -q -k "sum_of_floating or other_result_types or clusters or distinct or max_unique" -n 4 2>&1 | tail -40
