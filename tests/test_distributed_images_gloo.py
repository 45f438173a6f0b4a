"""The ONE-collective exchange of partial group tables (ssgpu.h "result images", BASELINE config #4) on CPU
processes (gloo, world_size 2).

No kernel can run here.  What runs for real: `ssgpu_plan_image_layout` (the library's own layout function, on a
bind-only context), ONE `all_gather_into_tensor` of fixed-size images per step, and the merge plan
`supersonic_amd.distributed._merge_plan(..., valid="__valid")` (executed by the CPU oracle).  The pack / unpack
kernels are stood in for by numpy copies that follow the offsets the library reports; the kernels themselves are
covered on the GPU by tests/test_parity_gpu.py::test_device_sharded_group_aggregate_*."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import supersonic_amd as ss
from supersonic_amd.distributed import _merge_plan, _merge_spec
from oracle import oracle
from helpers import sort_rows, assert_cols_equal
from test_distributed_group_gloo import make_view, spec, child, oracle_executor, shard_of, free_port

KEYS = ["k1", "k2"]


def pack(view, cap, image_bytes, offs):
    img = np.zeros(image_bytes, np.uint8)
    rows = min(view.row_count(), cap)
    img[:32].view(np.int64)[:] = [rows, cap, int(view.row_count() > cap), view.row_count()]
    for i in range(view.column_count()):
        col = view.column(i)
        raw = np.ascontiguousarray(col.data[:rows]).view(np.uint8).reshape(-1)
        img[offs[i][0]: offs[i][0] + raw.size] = raw
        if offs[i][1] >= 0:
            z = np.zeros(rows, np.uint8) if col.is_null is None else col.is_null[:rows].astype(np.uint8)
            img[offs[i][1]: offs[i][1] + rows] = z
    return img


def unpack(images, world, cap, image_bytes, schema, offs):
    cols, valid = [], np.zeros(world * cap, bool)
    for i in range(schema.attribute_count()):
        dt = np.dtype(ss.numpy_dtype(schema.attribute(i).type()))
        data = np.zeros(world * cap, dt)
        nulls = np.zeros(world * cap, bool) if offs[i][1] >= 0 else None
        for r in range(world):
            img = images[r * image_bytes: (r + 1) * image_bytes]
            rows = int(img[:8].view(np.int64)[0])
            data[r * cap: r * cap + rows] = img[offs[i][0]: offs[i][0] + rows * dt.itemsize].view(dt)
            if nulls is not None:
                nulls[r * cap: r * cap + rows] = img[offs[i][1]: offs[i][1] + rows] != 0
            valid[r * cap: r * cap + rows] = True
        cols.append(ss.Column(data, nulls))
    attrs = [schema.attribute(i) for i in range(schema.attribute_count())] + [ss.Attribute("__valid", ss.BOOL)]
    return ss.View(ss.TupleSchema(attrs), cols + [ss.Column(valid)])


def worker(rank, world, port, n, with_filter, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = make_view(n)
    bounds = [0, n // 3, n]
    shard = shard_of(full, bounds[rank], bounds[rank + 1])
    op = ss.GroupAggregate(ss.ProjectNamedAttributes(KEYS), spec(), None, child(shard, with_filter))
    layout_plan = ss.Plan(op, ss.Context(-1))                         # bind-only context: layout is host code
    partial = oracle_executor(op)
    image_bytes, _unpacked, offs = layout_plan.image_layout(cap, world)
    mine = torch.from_numpy(pack(partial, cap, image_bytes, offs))
    everyone = torch.empty(world * image_bytes, dtype=torch.uint8)
    dist.all_gather_into_tensor(everyone, mine)                        # the step's ONE collective
    images = everyone.numpy()
    overflow = any(int(images[r * image_bytes + 16: r * image_bytes + 24].view(np.int64)[0]) for r in range(world))
    merged_spec, counts = _merge_spec(spec())
    table = unpack(images, world, cap, image_bytes, partial.schema(), offs)
    out = oracle_executor(_merge_plan(KEYS, merged_spec, counts, partial.schema(), table, valid="__valid"))
    cols = [(out.column(i).data, out.column(i).is_null) for i in range(out.column_count())]
    schema = [(out.schema().attribute(i).name(), out.schema().attribute(i).type(), out.schema().attribute(i).is_nullable())
              for i in range(out.schema().attribute_count())]
    q.put((rank, overflow, schema, cols))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,with_filter,cap", [(20001, True, 1024), (20001, False, 256), (3000, False, 64), (0, False, 16)])
def test_group_tables_travel_as_fixed_size_images(n, with_filter, cap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, n, with_filter, cap, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want_schema, want = oracle.run(ss.GroupAggregate(ss.ProjectNamedAttributes(KEYS), spec(), None, child(make_view(n), with_filter)))
    groups = len(want[0][0])
    for _rank, overflow, schema, cols in results:
        if groups > cap:                       # a table that does not fit is truncated AND flagged (every rank sees the flag)
            assert overflow
            continue
        assert not overflow
        assert [tuple(x) for x in schema] == [tuple(x) for x in want_schema]
        assert_cols_equal(sort_rows(cols), sort_rows(want), context="image exchange")
