cd $GRAFT_REPO_ROOT
python -m pytest tests/test_parity_gpu.py -x -q -k "specialized_first_last" 2>&1 | tail -25
