cd $GRAFT_REPO_ROOT
python -m pytest tests/test_parity_gpu.py -x -q -k "max_unique or distinct or first or last or clusters" -n 4 2>&1 | tail -25
