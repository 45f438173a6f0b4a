"""Development aid: the forced one-table forms of the GroupAggregate with the runtime's debug prints."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import supersonic_amd as ss
from test_parity_gpu import group_query, make_view

for n in (65, 1025, 100003):
    for resident in (1, 0):
        for nullable in (False, True):
            ctx = ss.Context(0)
            for k, v in (("group_partition", 2), ("group_slab", 2), ("group_resident", resident), ("debug_timing", 1)):
                ctx.set_option(k, v)
            op = group_query(make_view(n, nullable=nullable), False, ("k1",) if nullable else ("k1", "k2"))
            plan = ss.Plan(op, ctx)
            plan.run()
            print("n=%d resident=%d nullable=%s -> shape %d rows %d" % (n, resident, nullable, plan.stage_info()[-1]["group_shape"], plan.fetch().row_count()), flush=True)
