"""Stress of the key-range routing test: repeats its body (four sources route their tables into four owners' images on one GPU,
the owners unpack and merge) and reports groups that end up with more than one owner.  Found with four of these in parallel
(20 % of the iterations): torch's concatenation of the images had not finished on torch's stream when the library's unpack kernel
read it on the context's stream -- a race of the TEST's two streams (the product orders them with events: distributed.py), fixed
there with a synchronise.  Usage: python tools/stress_route_images.py [iterations]"""
import sys, os, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import supersonic_amd as ss
from supersonic_amd.distributed import _shard_spec, _merge_spec, _merge_plan
from test_parity_gpu import make_view
from helpers import to_cols

ctx = ss.Context(0)
n, world = 2000, 4
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    view = make_view(n, nullable=True)
    spec = (ss.AggregationSpecification().AddAggregation(ss.SUM, "b", "sb").AddAggregation(ss.SUM, "d1", "sd").AddAggregation(ss.MIN, "d0", "mn")
            .AddAggregation(ss.MAX, "d", "mx").AddAggregation(ss.COUNT, "d0", "c0").AddAggregation(ss.COUNT, "", "n"))
    keys = ["k1", "t"]
    shard_spec, with_residual = _shard_spec(spec, view.schema())
    merged_spec, counts = _merge_spec(spec, with_residual)
    cuts = [n * i // world for i in range(world + 1)]
    sources, cap = [], max(1024, int(n / world * 1.3 / world) + 2048)
    dev = torch.device("cuda", 0)
    for s in range(world):
        sv = ss.View(view.schema(), [ss.Column(view.column(i).data[cuts[s]:cuts[s + 1]], None if view.column(i).is_null is None else view.column(i).is_null[cuts[s]:cuts[s + 1]])
                                     for i in range(view.column_count())])
        plan = ss.Plan(ss.GroupAggregate(ss.ProjectNamedAttributes(keys), shard_spec, None, ss.ScanView(sv)), ctx)
        plan.run()
        image_bytes, _ub, _offs = plan.image_layout(cap, 1)
        out = torch.zeros(world * image_bytes, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        plan.route_images(len(keys), world, cap, out.data_ptr())
        ctx.synchronize()
        hdr = [out[d * image_bytes:d * image_bytes + 64].view(torch.int64).tolist() for d in range(world)]
        local = to_cols(plan.fetch())
        sources.append((plan, out, hdr, len(local[0][0])))
    plan0 = sources[0][0]
    _ib, unpacked_bytes, _offs = plan0.image_layout(cap, world)
    per_owner = []
    for d in range(world):
        arrived = torch.cat([o[d * image_bytes:(d + 1) * image_bytes] for (_p, o, _h, _n) in sources])
        unpacked = torch.zeros(unpacked_bytes, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        everyone = plan0.unpack_images(arrived.data_ptr(), world, cap, unpacked.data_ptr())
        ctx.synchronize()
        got = ss.drain(_merge_plan(keys, merged_spec, counts, plan0.result_schema, everyone, valid="__valid").CreateCursor(ctx), 1 << 30)
        cols = to_cols(got)
        ks = list(zip(*[np.where(z, -7, dcol).tolist() if z is not None else dcol.tolist() for (dcol, z) in cols[:2]]))
        per_owner.append(ks)
    allk = [k for ks in per_owner for k in ks]
    if len(allk) != len(set(allk)):
        bad += 1
        cnt = collections.Counter(allk)
        dups = [k for k, c in cnt.items() if c > 1]
        print("iteration", it, "duplicates", dups[:10])
        for k in dups[:4]:
            print("  key", k, "owners", [d for d in range(world) if k in per_owner[d]], "times within owner", [per_owner[d].count(k) for d in range(world)])
        print("  headers (rows, cap, flag, have) per source:", [[h[:4] for h in hd] for (_p, _o, hd, _n) in sources], "local groups", [x[3] for x in sources])
print("done", bad, "bad iterations")
