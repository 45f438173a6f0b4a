"""Fuzz hunt for the chunked forms (ssgpu_plan_run_host / _stream_*, ssgpu.h "CHUNKED STAGING"): seeded random plans of tests/fuzz_plans.py whose
chunked_form() is 1 / 2 / 3 run over their HOST view in chunks of a random size and must give the oracle's rows (ordered where the plan's
order is defined); plans without a chunked form must be refused with ERROR_NOT_IMPLEMENTED and nothing else.
Usage: python tools/fuzz_chunked.py <plan|ordered_aggregate_plan|plain_group> <first seed> <count> [rows]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("SSGPU_SPECIALIZE", "0")
import numpy as np
import supersonic_amd as ss
from oracle import oracle
from helpers import assert_cols_equal, schema_list, sort_rows, to_cols
from fuzz_plans import Gen, make_view, random_plain_group

gen, first, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 1537
ctx = ss.Context(0)
ctx.set_option("dense_min_rows", 1)
bad = ran = refused = 0
kinds = {1: 0, 2: 0, 3: 0}
for seed in range(first, first + count):
    view = make_view(rows, 1000 + seed)
    if gen == "plain_group":
        op, ordered = random_plain_group(seed, view), False
    else:
        out = getattr(Gen(seed), gen)(view)
        op, ordered = out if isinstance(out, tuple) else (out, True)
    try:
        oschema, want = oracle.run(op, 1 << 20)
    except oracle.OracleError:
        continue
    rng = np.random.default_rng(seed)
    chunk = int(rng.choice([1, 7, 100, 333, 1000, 1024, 4096, rows // 2 + 1, rows, rows + 5, 0]))
    try:
        plan = ss.Plan(op, ctx)
        try:
            kind = plan.chunked_form()[0]
        except ss.SupersonicException as e:
            assert e.return_code == ss.ERROR_NOT_IMPLEMENTED, e
            refused += 1
            try:
                plan.run_host(chunk_rows=chunk)
                raise AssertionError("run_host ran a plan chunked_form refuses")
            except ss.SupersonicException as e2:
                assert e2.return_code == ss.ERROR_NOT_IMPLEMENTED, e2
            continue
        kinds[kind] += 1
        try:
            plan.run_host(chunk_rows=chunk)
        except ss.SupersonicException as e:
            if e.return_code == ss.ERROR_NOT_IMPLEMENTED and "NaN" in str(e):
                refused += 1          # a NaN reached a floating MIN / MAX: the documented refusal of form 3
                continue
            raise
        got = to_cols(plan.fetch())
        assert schema_list(plan.result_schema) == oschema, (schema_list(plan.result_schema), oschema)
        if not ordered or kind == 3 and not ordered:
            got, want = sort_rows(got), sort_rows(want)
        assert_cols_equal(got, want, context="seed %d kind %d chunk %d" % (seed, kind, chunk))
        ran += 1
    except Exception:
        bad += 1
        print("seed", seed, "chunk", chunk, "FAILED:", traceback.format_exc().splitlines()[-1][:400])
print("%s seeds %d..%d rows %d: %d chunked runs (forms %s), %d refused, %d failures" % (gen, first, first + count - 1, rows, ran, kinds, refused, bad))
