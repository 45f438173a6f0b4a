"""Development: Compute over MANY input columns -- which outputs are wrong, as a function of the number of staged input columns."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import supersonic_amd as ss
NA = ss.NamedAttribute
n = 5000
rng = np.random.default_rng(1)
for spec_mode in (0, 1):
    for n_in in (24, 26, 28, 32):
        for nullable in (False, True):
            schema = ss.TupleSchema([ss.Attribute("c%d" % i, ss.DOUBLE, ss.NULLABLE if nullable else ss.NOT_NULLABLE) for i in range(n_in)])
            data = [rng.integers(-1000, 1000, n).astype(np.float64) for _ in range(n_in)]
            nulls = [(rng.random(n) < 0.1) if nullable else None for _ in range(n_in)]
            view = ss.View(schema, [ss.Column(d, z) for d, z in zip(data, nulls)])
            e = ss.CompoundExpression()
            for i in range(0, n_in - 1, 2):
                e.AddAs("s%d" % i, ss.Plus(NA("c%d" % i), NA("c%d" % (i + 1))))
            ctx = ss.Context(0); ctx.set_option("specialize", spec_mode)
            plan = ss.Plan(ss.Compute(e, ss.ScanView(view)), ctx)
            plan.run()
            got = plan.fetch()
            bad = []
            for j, i in enumerate(range(0, n_in - 1, 2)):
                want = data[i] + data[i + 1]
                wz = (nulls[i] | nulls[i + 1]) if nullable else np.zeros(n, bool)
                gz = got.column(j).is_null if got.column(j).is_null is not None else np.zeros(n, bool)
                if not np.array_equal(gz, wz) or not np.array_equal(got.column(j).data[~wz], want[~wz]):
                    bad.append(j)
            print("specialize %d inputs %d nullable %s: %s" % (spec_mode, n_in, nullable, "ok" if not bad else "BAD outputs %s of %d" % (bad, n_in // 2)))
