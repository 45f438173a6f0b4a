cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fuzz_gpu.py -q -k "sequential or ordered" -n 4 2>&1 | tail -40
