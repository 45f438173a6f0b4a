"""Pins the CPU oracle: every golden vector transcribed from the reference's own tests
(tests/golden/reference_tests.json, see transcribe_reference_tests.py for the citations) must be
reproduced by oracle/ss_oracle.c -- values, NULLs, result names, types, nullability and the
reference's bind / evaluation error codes.  Runs on CPU."""
import pytest

from golden_runner import build_plan, build_view, check, load_cases
from oracle import oracle

CASES = load_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_reproduces_reference_golden_vector(case):
    view = build_view(case["input"])
    if case.get("input2"):
        view = (view, build_view(case["input2"]))
    op = build_plan(case["plan"], view)
    if case["expect_error"]:
        with pytest.raises(oracle.OracleError) as e:
            oracle.run(op)
        # -1: the reference's test only requires the bind to FAIL (TestBoundFactoryFailure): any bind-time code (4xx)
        assert (400 <= e.value.return_code < 500) if case["expect_error"] == -1 else e.value.return_code == case["expect_error"]
        return
    schema, cols = oracle.run(op)
    check(case, schema, cols)
