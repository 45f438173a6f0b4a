"""Development: plain partition scatter under a Filter -- which predicates break it?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import supersonic_amd as ss
import bench
from helpers import assert_cols_equal, sort_rows, to_cols
from oracle import oracle

n = 200000
cols = bench.host_columns(np, "group", n, seed=3)
view = ss.View(bench.group_schema(ss), cols)
NA = ss.NamedAttribute
for name, pred in (("a > -1", ss.Greater(NA("a"), ss.ConstInt64(-1))), ("a > 499", ss.Greater(NA("a"), ss.ConstInt64(499))),
                   ("a < 500", ss.Less(NA("a"), ss.ConstInt64(500))), ("a > 998", ss.Greater(NA("a"), ss.ConstInt64(998)))):
    op = ss.GroupAggregate(ss.ProjectNamedAttributes(["k1", "k2"]), bench.group_spec(ss), None,
                           ss.Filter(pred, ss.ProjectAllAttributes(), ss.ScanView(view)))
    ctx = ss.Context(0)
    ctx.set_option("group_partition", 2); ctx.set_option("debug_timing", 1); ctx.set_option("part_n", 256)
    plan = ss.Plan(op, ctx)
    plan.run()
    got = plan.fetch()
    _s, want = oracle.run(op)
    print(name, "rows", got.row_count(), "want", len(want[0][0]), plan.stage_info()[0], flush=True)
    assert_cols_equal(sort_rows(to_cols(got)), sort_rows(want), context=name)
