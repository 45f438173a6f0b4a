"""The golden vectors transcribed from the reference's own tests, through the HIP path (C ABI)."""
import pytest

import supersonic_amd as ss
from golden_runner import build_plan, build_view, check, load_cases
from helpers import schema_list, to_cols

pytestmark = pytest.mark.gpu
CASES = [c for c in load_cases() if c["kind"] != "binding"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_hip_path_reproduces_reference_golden_vector(gpu_ctx, case):
    view = build_view(case["input"])
    if case.get("input2"):
        view = (view, build_view(case["input2"]))
    op = build_plan(case["plan"], view)
    if case["expect_error"]:
        with pytest.raises(ss.SupersonicException) as e:
            ss.drain(op.CreateCursor(gpu_ctx))
        # -1: the reference's test only requires the bind to FAIL (TestBoundFactoryFailure): any bind-time code (4xx)
        assert (400 <= e.value.return_code < 500) if case["expect_error"] == -1 else e.value.return_code == case["expect_error"]
        return
    cur = op.CreateCursor(gpu_ctx)
    got = ss.drain(cur, 1024)
    check(case, schema_list(cur.schema()), to_cols(got))
