"""`bench.py --gpus N` must itself become N ranks (the driver's contract command) and print ONE line with
n_gpus = N.  No kernel can run here, so the run is `--dry-run`: the same launch path (self re-exec under
torch.distributed.run), the same collective shape over gloo, the same report -- flagged `dry_run`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "5", "--warmup", "2"] + list(extra),
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout            # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


@pytest.mark.parametrize("query,scaling", [("wide", "weak"), ("group", "strong")])
def test_bench_gpus_2_launches_two_ranks(query, scaling):
    line = run_bench("--gpus", "2", "--query", query, "--scaling", scaling, "--rows", "1000")
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["scaling"] == scaling
    assert line["steps"] == 5 and line["warmup"] == 2
    assert line["config"]["collectives_per_step"] == 1
    assert line["config"]["rows_per_gpu"] == (1000 if scaling == "weak" else 500)
    if scaling == "weak":
        # a weak run reports BOTH regimes in its one line: --rows per GPU (the timed region) and --rows in total (strong)
        assert line["regimes"]["weak"]["rows_total"] == 2000 and line["regimes"]["strong"]["rows_total"] == 1000
        assert line["regimes"]["strong"]["rows_per_gpu"] == 500 and line["regimes"]["strong"]["ms_per_step"] > 0
    else:
        assert "regimes" not in line


def test_bench_default_is_one_gpu():
    line = run_bench()
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["unit"] == "rows/s"
