cd $GRAFT_REPO_ROOT
python -m pytest tests/test_parity_gpu.py -q -x -k "four_partial_tables" 2>&1 | tail -15
