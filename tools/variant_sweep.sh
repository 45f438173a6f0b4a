# A/B runs of build variants (make OBJDIR=../lib/var_X/obj OUT=../lib/var_X/libssgpu.so VMDEF="-D..."), selected with SSGPU_LIB
cd /root/repo
Q=${Q:-sort_key}
for v in "" $VARIANTS; do
  if [ -z "$v" ]; then export SSGPU_LIB=; else export SSGPU_LIB=/root/repo/supersonic_amd/lib/var_$v/libssgpu.so; fi
  echo "== variant ${v:-default}"
  timeout 120 python tools/perf_sweep.py --queries $Q --tiles 0 --reps 7 2>&1 | grep -v "amdgpu.ids" | tail -2
done
