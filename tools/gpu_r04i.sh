cd $GRAFT_REPO_ROOT
for i in 1 2; do
python bench.py --query group --force-distributed --rows 12500000 --no-cpu-baseline --no-regimes 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('group dist1 12.5M key_range', round(d['ms_per_step'],4))"
python bench.py --query group --force-distributed --exchange all_gather --rows 12500000 --no-cpu-baseline --no-regimes 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('group dist1 12.5M all_gather', round(d['ms_per_step'],4))"
python bench.py --query group --rows 12500000 --no-cpu-baseline --no-configs 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('group plain 12.5M', round(d['ms_per_step'],4))"
done
