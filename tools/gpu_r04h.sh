cd $GRAFT_REPO_ROOT
python -m pytest tests/test_cursor_contract_gpu.py -x -q 2>&1 | tail -15
