cd $GRAFT_REPO_ROOT
python -m pytest tests/test_cpp_facade.py -q -x 2>&1 | tail -3
