cd $GRAFT_REPO_ROOT
python -m pytest tests/test_parity_gpu.py -q -x -k "more_than_16_words" 2>&1 | tail -15
