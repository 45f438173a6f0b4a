"""A COLD process under the library's default policy (specialize = 3, empty code-object cache): the headline query runs interpreted at
once, the worker thread compiles its kernel meanwhile, the plan switches when it is there.  Prints how long that took and the step
time before / after.      SSGPU_RTC_CACHE_DIR=$(mktemp -d) python tools/cold_process_bench.py > gpurun_out/cold_process.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.pop("SSGPU_SPECIALIZE", None)
import torch          # noqa: E402
import bench          # noqa: E402
import supersonic_amd as ss   # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    device = torch.device("cuda", 0)
    ctx = ss.Context(0)
    ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
    cols = bench.gen_device_columns(torch, rows, 42, device)
    view = ss.DeviceView(bench.bench_schema(ss), [(t.data_ptr(), 0) for t in cols], rows)
    t_create = time.perf_counter()
    plan = ss.Plan(bench.build_plan(ss, view), ctx)

    def steps_ms(k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            plan.run(view)
        ctx.synchronize()
        return (time.perf_counter() - t0) / k * 1e3
    plan.run(view)
    ctx.synchronize()
    first_run_s = time.perf_counter() - t_create
    row = [plan.fetch().column(i).data[0].item() for i in range(7)]
    before = steps_ms(50)
    reason = plan.specialize_reason()
    n_interpreted = 51
    while plan.specialized() == 0 and time.perf_counter() - t_create < 300:
        plan.run(view)
        n_interpreted += 1
        ctx.synchronize()
        time.sleep(0.05)
    switched_s = time.perf_counter() - t_create
    after = steps_ms(50)
    same = [plan.fetch().column(i).data[0].item() for i in range(7)] == row
    print(json.dumps({"rows": rows, "policy": "default (specialize = 3)", "first_result_after_s": first_run_s, "ms_per_step_interpreted": before,
                      "reason_while_waiting": reason, "runs_before_the_switch": n_interpreted, "specialised_after_s": switched_s,
                      "ms_per_step_specialised": after, "specialized_stages": plan.specialized(), "same_row": same,
                      "rtc": {k: v for k, v in ss.memory_stats().items() if k.startswith("rtc")}}))


if __name__ == "__main__":
    main()
