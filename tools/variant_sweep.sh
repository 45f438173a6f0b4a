cd /root/repo
for v in B E A; do
  if [ $v = A ]; then export SSGPU_LIB=; else export SSGPU_LIB=/root/repo/supersonic_amd/lib/var_$v/libssgpu.so; fi
  for w in 3 4 5; do
      echo "== variant $v wgs=$w"
      timeout 120 python tools/perf_sweep.py --queries wide,narrow,sum1 --tiles 0 --wgs $w --reps 7 2>&1 | grep -v "^ *\[\|amdgpu.ids" 
  done
done
