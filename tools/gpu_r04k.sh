# the headline reads 6 % slower after the GPU suite on the same box (not thermal: 46 C before and after, persists after 45 s idle).  What is it?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04k
B() { python bench.py --no-cpu-baseline --no-configs $2 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"; }
T() { python - <<'PY'
import torch, time
x = torch.empty(3_200_000_000, dtype=torch.uint8, device="cuda"); y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): y.copy_(x)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
print("torch 3.2 GB copy: %.4f ms = %.2f TB/s (r+w)" % (dt*1e3, 6.4e9/dt/1e12))
s = torch.zeros(1, device="cuda")
z = x.view(torch.int64)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): s = z.sum()
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
print("torch 3.2 GB int64 sum: %.4f ms = %.2f TB/s" % (dt*1e3, 3.2e9/dt/1e12))
PY
}
B fresh; B fresh_interpreted --no-specialize; T
timeout 1200 python -m pytest tests -m gpu -x -q -n 4 > gpurun_out/r04k/suite.log 2>&1; tail -1 gpurun_out/r04k/suite.log
B after_suite; B after_suite_interpreted --no-specialize; T
SSGPU_RTC_CACHE_DIR= B after_suite_fresh_compile
cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null
grep -i "huge\|MemFree\|MemAvailable" /proc/meminfo | head -6
