"""Multi-GPU execution of the sharded hot path: one process per GPU, contiguous row
ranges per rank, ONE exchange step.

* ScalarAggregate: `Plan.run_partial` + element-wise all-reduce of the partial-aggregate state +
  `Plan.finalize` (see bench.py).
* GroupAggregate with general keys (this module): every rank aggregates its shard into a partial
  group table, the tables (a few MB: groups x (keys + partial aggregates)) are all-gathered, and
  every rank merges them with a second, local GroupAggregate whose aggregations are the merge
  functions of the first (SUM of sums, MIN of mins, MAX of maxes, SUM of counts).  No row of the
  input ever crosses the fabric.

The reference is a single-process library (no counterpart to cite); the operations are built with
the same factories the single-GPU path uses, so binding, typing and NULL rules stay the reference's.
`executor(op) -> View` runs an operation tree to completion: `device_executor(ctx)` on a GPU, or any
callable with the same contract (the CPU tests pass the oracle)."""
import numpy as np

from . import api as ss


def device_executor(ctx):
    def run(op):
        return ss.drain(op.CreateCursor(ctx), 1 << 20)
    return run


def _merge_spec(spec):
    merged = ss.AggregationSpecification()
    counts = []
    for (aggregation, distinct, out_type, _input_name, output_name) in spec.elements:
        if distinct or aggregation == ss.CONCAT:   # FIRST / LAST merge as themselves: partial tables are gathered in rank (= row) order
            raise ss.SupersonicException(ss.ERROR_NOT_IMPLEMENTED, "aggregation cannot be merged across shards")
        if aggregation == ss.COUNT:
            merged.AddAggregation(ss.SUM, output_name, output_name)   # COUNT merges as SUM of the partial counts
            counts.append(output_name)
        else:
            merged.AddAggregation(aggregation, output_name, output_name)
    return merged, counts


def _all_gather_view(view, group, device):
    """Concatenate every rank's View (same schema) in rank order."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rows = torch.tensor([view.row_count()], dtype=torch.int64, device=device)
    all_rows = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(all_rows, rows, group=group)
    counts = [int(t.item()) for t in all_rows]
    cap = max(max(counts), 1)
    cols = []
    for i in range(view.column_count()):
        col = view.column(i)
        parts = []
        for arr, present in ((col.data, True), (col.is_null, col.is_null is not None)):
            # every rank must issue the same collectives: nullability comes from the schema
            if arr is None and not view.schema().attribute(i).is_nullable():
                parts.append(None)
                continue
            if arr is None:
                arr = np.zeros(view.row_count(), dtype=np.bool_)
            raw = np.zeros(cap * arr.dtype.itemsize, dtype=np.uint8)
            raw[: arr.nbytes] = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)[: arr.nbytes]
            mine = torch.from_numpy(raw).to(device)
            gathered = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine, group=group)
            chunks = [g.cpu().numpy()[: counts[r] * arr.dtype.itemsize].view(arr.dtype) for r, g in enumerate(gathered)]
            parts.append(np.concatenate(chunks) if chunks else np.zeros(0, arr.dtype))
        cols.append(ss.Column(parts[0], parts[1]))
    return ss.View(view.schema(), cols, sum(counts))


def sharded_group_aggregate(group_by, spec, local_child, executor, group=None, device="cpu"):
    """GroupAggregate(group_by, spec, <all shards of local_child>) with one all-gather.

    group_by: list of key attribute names; spec: AggregationSpecification; local_child: this rank's
    Operation (e.g. Filter(...ScanView(shard))).  Returns the same View on every rank (group order
    unspecified, as in the reference)."""
    merged_spec, counts = _merge_spec(spec)
    partial = executor(ss.GroupAggregate(ss.ProjectNamedAttributes(list(group_by)), spec, None, local_child))
    everyone = _all_gather_view(partial, group, device)
    merged = ss.GroupAggregate(ss.ProjectNamedAttributes(list(group_by)), merged_spec, None, ss.ScanView(everyone))
    if counts:
        # SUM(...) is NULLABLE, COUNT is not: restore the schema of the single-process result
        schema = partial.schema()
        e = ss.CompoundExpression()
        for i in range(schema.attribute_count()):
            a = schema.attribute(i)
            if a.name() in counts:
                zero = ss.ConstUint64(0) if a.type() == ss.UINT64 else ss.ConstUint32(0)
                e.AddAs(a.name(), ss.IfNull(ss.NamedAttribute(a.name()), zero))
            else:
                e.Add(ss.NamedAttribute(a.name()))
        merged = ss.Compute(e, merged)
    return executor(merged)
