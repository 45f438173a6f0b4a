cd $GRAFT_REPO_ROOT
python -m pytest tests/test_parity_gpu.py -x -q -k "clusters or distinct" -n 4 2>&1 | tail -30
