"""Development aid: which input rows belong to the groups the partitioned GroupAggregate gets wrong."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import supersonic_amd as ss

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
rng = np.random.default_rng(7)
key = rng.integers(0, 120000, n).astype(np.int64)
if len(sys.argv) > 2 and sys.argv[2] == "empty":
    key[::1000] = -1
val = rng.integers(-1000, 1000, n).astype(np.int64)
schema = ss.TupleSchema([ss.Attribute("k", ss.INT64), ss.Attribute("v", ss.INT64)])
view = ss.View(schema, [ss.Column(key), ss.Column(val)], n)
spec = ss.AggregationSpecification().AddAggregation(ss.SUM, "v", "s").AddAggregation(ss.COUNT, "", "c").AddAggregation(ss.MIN, "v", "mn")
op = ss.GroupAggregate(ss.ProjectNamedAttributes(["k"]), spec, None, ss.ScanView(view))
ctx = ss.Context(0)
ctx.set_option("group_partition", 2)
ctx.set_option("debug_timing", 1)
got = ss.drain(op.CreateCursor(ctx), 1 << 20)
gk, gs, gc, gm = [got.column(i).data for i in range(4)]
o = np.argsort(gk); gk, gs, gc, gm = gk[o], gs[o], gc[o], gm[o]
uk, inv = np.unique(key, return_inverse=True)
ws = np.bincount(inv, weights=val).astype(np.int64); wc = np.bincount(inv)
wm = np.full(len(uk), 1 << 40); np.minimum.at(wm, inv, val)
print("groups got/want", len(gk), len(uk), "keys equal", np.array_equal(gk, uk))
if np.array_equal(gk, uk):
    bad = np.nonzero((gs != ws) | (gc != wc) | (gm != wm))[0]
    print("bad groups", len(bad), "sum bad", int((gs != ws).sum()), "count bad", int((gc != wc).sum()), "min bad", int((gm != wm).sum()))
    rows = np.nonzero(np.isin(inv, bad))[0]
    print("rows of bad groups:", len(rows), "min", rows.min() if len(rows) else None, "max", rows.max() if len(rows) else None)
    print("histogram of row index / 512 (tile) for bad groups' rows, top:", np.unique(rows // 512, return_counts=True)[0][:20], np.bincount(rows // 51200)[:8])
    for b in bad[:8]:
        r = np.nonzero(inv == b)[0]
        print(" key", uk[b], "rows", r, "vals", val[r], "got sum/cnt/min", gs[b], gc[b], gm[b], "want", ws[b], wc[b], wm[b])
