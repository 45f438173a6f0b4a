cd $GRAFT_REPO_ROOT
export SSGPU_SPECIALIZE=0
for i in 1 2 3 4; do python tools/dbg/route_stress.py 150 2>&1 | grep -v "amdgpu.ids" | tail -3 > gpurun_out/route_stress_$i.log & done
wait
cat gpurun_out/route_stress_*.log | tail -12
for i in 1 2 3; do
python -m pytest tests -q -x -m gpu -k "sharded or distributed or merge or partition or group or route" -n 4 2>&1 | tail -1
done
