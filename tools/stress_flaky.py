import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import supersonic_amd as ss
from oracle import oracle
import test_parity_gpu as T
from helpers import to_cols

ctx = ss.Context(0)
bad = 0
for n in [65, 513, 100003]:
    view = T.make_view(n, nullable=True)
    op = T.compute_exprs(view)
    _s, want = oracle.run(op)
    for rep in range(40):
        cur = op.CreateCursor(ctx)
        got = to_cols(ss.drain(cur))
        for i, ((gd, gz), (wd, wz)) in enumerate(zip(got, want)):
            gz_ = np.zeros(len(gd), bool) if gz is None else gz
            wz_ = np.zeros(len(wd), bool) if wz is None else wz
            if not np.array_equal(gz_, wz_):
                idx = np.nonzero(gz_ != wz_)[0]
                print("n=%d rep=%d col=%d (%s): %d null-mask diffs, first %s got %s want %s" % (
                    n, rep, i, cur.schema().attribute(i).name(), len(idx), idx[:8], gz_[idx[:8]], wz_[idx[:8]]))
                bad += 1
            else:
                live = ~wz_
                if not np.array_equal(gd[live].view(np.uint8), wd[live].view(np.uint8)):
                    idx = np.nonzero(gd[live] != wd[live])[0]
                    print("n=%d rep=%d col=%d (%s): %d value diffs first %s" % (n, rep, i, cur.schema().attribute(i).name(), len(idx), idx[:8]))
                    bad += 1
print("bad =", bad)
