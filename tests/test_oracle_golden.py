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


def test_oracle_date_printing_agrees_with_the_c_library():
    """PrintTyped<DATE / DATETIME> is gmtime_r + strftime in the reference (types_infrastructure.cc:92-114, utils/walltime.cc:173-189); the
    oracle (and the host side of CONCAT) compute the calendar themselves -- pinned here against this box's C library, any year."""
    import ctypes
    import numpy as np
    from oracle import oracle
    libc = ctypes.CDLL(None)

    class Tm(ctypes.Structure):
        _fields_ = [(n, ctypes.c_int) for n in ("sec", "min", "hour", "mday", "mon", "year", "wday", "yday", "isdst")] + [("gmtoff", ctypes.c_long), ("zone", ctypes.c_char_p)]
    libc.gmtime_r.restype = ctypes.c_void_p
    libc.gmtime_r.argtypes = [ctypes.POINTER(ctypes.c_long), ctypes.POINTER(Tm)]
    libc.strftime.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(Tm)]
    libc.strftime.restype = ctypes.c_size_t

    def c_print(seconds, fmt):
        tm, buf = Tm(), ctypes.create_string_buffer(128)
        assert libc.gmtime_r(ctypes.byref(ctypes.c_long(seconds)), ctypes.byref(tm))
        n = libc.strftime(buf, 128, fmt, ctypes.byref(tm))
        return buf.raw[:n]
    rng = np.random.default_rng(7)
    stamps = [0, -1, 1, 86399, 86400, -86400, -86401, 951782400, -62135596800, -62135596801, -62167219200, 253402300800, -(1 << 62), 1 << 62]
    stamps += [int(x) for x in rng.integers(-(1 << 62), 1 << 62, 3000)] + [int(x) for x in rng.integers(-4 * 10 ** 15, 4 * 10 ** 15, 3000)]
    for us in stamps:
        secs = abs(us) // 1000000 * (1 if us >= 0 else -1)
        assert oracle._print_typed(oracle.T_DATETIME, us) == c_print(secs, b"%Y/%m/%d-%H:%M:%S"), us
    for day in [0, 1, -1, 11016, 24855, -24855, 24856, -24856, (1 << 31) - 1, -(1 << 31)] + [int(x) for x in rng.integers(-(1 << 31), 1 << 31, 3000)]:
        wrapped = ((day * 86400 + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)        # the reference's int32 product
        assert oracle._print_typed(oracle.T_DATE, day) == c_print(wrapped, b"%Y/%m/%d"), day
