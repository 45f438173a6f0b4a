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


RESIDUAL = "$res"    # suffix of the hidden column that carries a DOUBLE sum's residual across shards


class _HostBounce(object):
    """torch.distributed for DEVICE tensors over a backend that only moves host memory (gloo): every collective of the device
    drivers goes device -> host -> collective -> device.  It exists for ONE purpose: several ranks on ONE GPU (RCCL refuses two
    ranks of a device), so that the drivers' kernels -- pack / route / unpack, the dense fold -- meet images that a second real
    process produced (tests/test_two_ranks_one_gpu.py).  A job on RCCL never comes here."""

    def __init__(self, dist):
        self._dist = dist
        self.ReduceOp = dist.ReduceOp

    def __getattr__(self, name):
        return getattr(self._dist, name)

    def all_reduce(self, tensor, op=None, group=None):
        host = tensor.cpu()
        self._dist.all_reduce(host, op=op, group=group)
        tensor.copy_(host)

    def all_to_all_single(self, output, input, output_split_sizes=None, input_split_sizes=None, group=None):
        host = output.cpu()
        self._dist.all_to_all_single(host, input.cpu(), output_split_sizes, input_split_sizes, group=group)
        output.copy_(host)

    def all_gather_into_tensor(self, output, input, group=None):
        world = self._dist.get_world_size(group)
        parts = [input.cpu().clone() for _ in range(world)]
        self._dist.all_gather(parts, input.cpu(), group=group)
        import torch
        output.copy_(torch.cat([p.reshape(-1) for p in parts]).view(output.dtype).reshape(output.shape))

    def all_gather(self, outputs, input, group=None):
        if not input.is_cuda:
            return self._dist.all_gather(outputs, input, group=group)
        parts = [o.cpu() for o in outputs]
        self._dist.all_gather(parts, input.cpu(), group=group)
        for o, p in zip(outputs, parts):
            o.copy_(p)


def _dist_for(group):
    """torch.distributed as the device drivers use it: itself on RCCL, through host copies on gloo (_HostBounce)."""
    import torch.distributed as dist
    return _HostBounce(dist) if dist.get_backend(group) == "gloo" else dist


def _shard_spec(spec, input_schema):
    """The specification a SHARD runs: `spec` plus, behind every DOUBLE SUM, the SUM_RESIDUAL of the same column
    (include/ssgpu.h) -- the partial sum travels as the double-double pair (s, e), so that the cross-shard total is the
    rounded sum of exact pairs instead of a sum of already rounded sums.  -> (shard spec, names of the sums that got one)."""
    shard = ss.AggregationSpecification()
    with_residual = []
    for (aggregation, distinct, out_type, input_name, output_name) in spec.elements:
        shard.elements.append((aggregation, distinct, out_type, input_name, output_name))
        if aggregation != ss.SUM or distinct or input_schema is None:
            continue
        pos = input_schema.LookupAttributePosition(input_name)
        if pos >= 0 and input_schema.attribute(pos).type() in (ss.FLOAT, ss.DOUBLE) and out_type in (ss.INT32, ss.UINT32, ss.INT64, ss.UINT64):
            # the reference adds and truncates row after row (aggregation_operators.h:173-185): a shard's result is not a partial sum
            raise ss.SupersonicException(ss.ERROR_NOT_IMPLEMENTED, "SUM of a floating input into an integer output cannot be merged across shards")
        if pos >= 0 and input_schema.attribute(pos).type() == ss.DOUBLE and out_type in (-1, ss.DOUBLE):
            shard.elements.append((ss.SUM_RESIDUAL, 0, -1, input_name, output_name + RESIDUAL))
            with_residual.append(output_name)
    return shard, with_residual


def _never_null(shard_spec, input_schema):
    """Result columns of a shard's table that are never NULL although their type says NULLABLE: SUM / MIN / MAX / FIRST / LAST
    (and a sum's residual) of a NOT NULL input column -- a group of a partial table has at least one row.  The merge plan reads
    them as NOT NULL columns: its aggregates then keep no contribution counts (half the atomics of a merge whose rows all
    belong to different groups); the merged results are NULLABLE again by the aggregates' own typing."""
    names = set()
    if input_schema is None:
        return names
    for (aggregation, distinct, _out_type, input_name, output_name) in shard_spec.elements:
        if distinct or aggregation in (ss.COUNT, ss.CONCAT):
            continue
        pos = input_schema.LookupAttributePosition(input_name)
        if pos >= 0 and not input_schema.attribute(pos).is_nullable():
            names.add(output_name)
    return names


def _declare_not_null(view, names):
    """`view` (a DeviceView) with the columns `names` declared NOT NULL (their NULL masks are all zero: _never_null)."""
    if not names:
        return view
    schema = view.schema()
    attrs, ptrs = [], []
    for i in range(schema.attribute_count()):
        a = schema.attribute(i)
        if a.name() in names and a.is_nullable():
            attrs.append(ss.Attribute(a.name(), a.type(), ss.NOT_NULLABLE))
            ptrs.append((view._ptrs[i][0], 0))
        else:
            attrs.append(a)
            ptrs.append(view._ptrs[i])
    return ss.DeviceView(ss.TupleSchema(attrs), ptrs, view.row_count())


def _merge_spec(spec, with_residual=()):
    merged = ss.AggregationSpecification()
    counts = []
    for (aggregation, distinct, out_type, _input_name, output_name) in spec.elements:
        if distinct or aggregation == ss.CONCAT:   # FIRST / LAST merge as themselves: partial tables are gathered in rank (= row) order
            raise ss.SupersonicException(ss.ERROR_NOT_IMPLEMENTED, "aggregation cannot be merged across shards as a partial result (sharded_group_aggregate sends the distinct pairs / the rows instead)")
        if aggregation == ss.COUNT:
            merged.AddAggregation(ss.SUM, output_name, output_name)   # COUNT merges as SUM of the partial counts
            counts.append(output_name)
        else:
            merged.AddAggregation(aggregation, output_name, output_name)
            if output_name in with_residual:
                merged.AddAggregation(ss.SUM, output_name + RESIDUAL, output_name + RESIDUAL)
    return merged, counts


def _schema_of(operation):
    """Result schema of an operation tree (bound on a device-less context: binding needs no GPU); None if it does not bind
    there -- the job then runs without residual columns, as before."""
    try:
        return ss.Plan(operation, ss.Context(-1)).result_schema
    except ss.SupersonicException:
        return None


def job_strings(local_strings, group=None):
    """Every byte string any rank's plan can meet, in one sorted list that is the same on all ranks: plans created with
    it as `extra_strings` build identical order-preserving dictionaries, so the INT32 codes of STRING columns mean the
    same thing on every rank and can cross shards like any other column (set-up only: one all_gather_object)."""
    import torch.distributed as dist
    mine = sorted(set(v.encode() if isinstance(v, str) else bytes(v) for v in local_strings))
    everyone = [None] * dist.get_world_size(group)
    dist.all_gather_object(everyone, mine, group=group)
    return sorted(set().union(*[set(x) for x in everyone]))


def _has_strings(schema):
    return any(schema.attribute(i).type() == ss.STRING for i in range(schema.attribute_count()))


def _gather_objects(values, group):
    import torch.distributed as dist
    everyone = [None] * dist.get_world_size(group)
    dist.all_gather_object(everyone, list(values), group=group)
    return everyone


def _all_gather_view(view, group, device):
    """Concatenate every rank's View (same schema) in rank order."""
    import torch
    import torch.distributed as dist

    if dist.get_backend(group) == "gloo":
        device = "cpu"                     # (host Views through a host backend: no bounce over the device)
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
        if view.schema().attribute(i).type() == ss.STRING:   # host Views hold the byte strings themselves
            chunks = _gather_objects([bytes(v) if v is not None else b"" for v in col.data], group)
            data = np.empty(sum(counts), dtype=object)
            data[:] = [v for chunk in chunks for v in chunk]
            nulls = None
            if view.schema().attribute(i).is_nullable():
                zs = _gather_objects((col.is_null if col.is_null is not None else np.zeros(view.row_count(), bool)).tolist(), group)
                nulls = np.array([z for chunk in zs for z in chunk], dtype=bool)
            cols.append(ss.Column(data, nulls))
            continue
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


def _owner_of_rows(view, n_keys, world):
    """Owner rank of every row of a partial table: a hash of the key cells (NULL keys hash as a flag) modulo the world
    size -- any function of the key values works as long as every rank computes the same one (host form of the key-range
    exchange; the device form is ssgpu_result_route_images)."""
    h = np.full(view.row_count(), 0x243F6A8885A308D3, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for k in range(n_keys):
            col = view.column(k)
            if col.data.dtype == object:
                cells = np.array([int.from_bytes(bytes(v or b"")[:8].ljust(8, b"\0"), "little") ^ len(v or b"") for v in col.data], dtype=np.uint64)
            else:
                cells = np.zeros(view.row_count(), dtype=np.uint64)
                raw = np.ascontiguousarray(col.data)
                cells[:] = raw.view({1: np.uint8, 4: np.uint32, 8: np.uint64}[raw.dtype.itemsize])
            if col.is_null is not None:
                cells = np.where(col.is_null, np.uint64(0x51), cells)
            h = (h ^ cells) * np.uint64(0x9E3779B97F4A7C15)
            h ^= h >> np.uint64(29)
    return ((h >> np.uint64(32)) % np.uint64(world)).astype(np.int64)


def _take_rows(view, mask):
    return ss.View(view.schema(), [ss.Column(view.column(i).data[mask], None if view.column(i).is_null is None else view.column(i).is_null[mask])
                                   for i in range(view.column_count())], int(mask.sum()))


def sharded_group_aggregate(group_by, spec, local_child, executor, group=None, device="cpu", key_range=False):
    """GroupAggregate(group_by, spec, <all shards of local_child>) with one exchange step.

    group_by: list of key attribute names; spec: AggregationSpecification; local_child: this rank's
    Operation (e.g. Filter(...ScanView(shard))).  Returns the same View on every rank (group order
    unspecified, as in the reference).

    key_range=False: every partial table goes to every rank (one all-gather) and every rank merges all of them.
    key_range=True: a partial row goes only to the rank that owns its key (hash of the group keys modulo the world size:
    one all-to-all), every rank merges 1 / world of the groups, and the finished slices are gathered -- the form whose
    merge work and link traffic shrink with the number of ranks."""
    import torch.distributed as dist
    child_schema = _schema_of(local_child)
    mode = _exchange_mode(spec, child_schema)
    if mode == "rows":
        # CONCAT and the row-after-row SUM need a group's VALUES in input order, not a partial result: the shards send the rows
        # themselves -- keys and aggregated columns -- and the owner runs the specification as written over them (rank order =
        # row order of the job, kept by the gather; the answer, DOUBLE sums included, is the single-process one)
        used = list(group_by) + [n for n in dict.fromkeys(e[3] for e in spec.elements if e[3]) if n not in group_by]
        partial = executor(ss.Project(ss.ProjectNamedAttributes(used), local_child))
        merge = lambda rows: ss.GroupAggregate(ss.ProjectNamedAttributes(list(group_by)), spec, None, ss.ScanView(rows))   # noqa: E731
    else:
        plain = ss.AggregationSpecification()
        plain.elements = [e for e in spec.elements if not e[1]]
        shard_spec, with_residual = _shard_spec(plain, child_schema)
        if mode == "distinct":
            partial, merge = _distinct_blocks(group_by, spec, plain, shard_spec, with_residual, local_child, executor)
        else:
            merged_spec, counts = _merge_spec(spec, with_residual)
            partial = executor(ss.GroupAggregate(ss.ProjectNamedAttributes(list(group_by)), shard_spec, None, local_child))
            schema = partial.schema()
            merge = lambda rows: _merge_plan(group_by, merged_spec, counts, schema, rows)   # noqa: E731
    if not key_range:
        return executor(merge(_all_gather_view(partial, group, device)))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    owner = _owner_of_rows(partial, len(group_by), world)
    # the all-to-all, spelled with all-gathers of the per-destination slices (host form: any backend has all_gather)
    mine = []
    for d in range(world):
        arrived = _all_gather_view(_take_rows(partial, owner == d), group, device)    # every rank receives destination d's rows ...
        if d == rank:
            mine = arrived                                                           # ... and keeps its own
    return _all_gather_view(executor(merge(mine)), group, device)


def _exchange_mode(spec, child_schema):
    """What the shards of a GroupAggregate send: "partial" tables (every aggregate merges: SUM of sums ...), partial tables next to
    the "distinct" (keys, value) pairs of every DISTINCT aggregate, or -- for CONCAT and the row-after-row SUM of a floating input
    into an integer result (aggregation_operators.h:173-185) -- the "rows" themselves."""
    mode = "partial"
    for (aggregation, distinct, out_type, input_name, _output_name) in spec.elements:
        if aggregation == ss.CONCAT:
            return "rows"
        if aggregation == ss.SUM and child_schema is not None and out_type in (ss.INT32, ss.UINT32, ss.INT64, ss.UINT64):
            pos = child_schema.LookupAttributePosition(input_name)
            if pos >= 0 and child_schema.attribute(pos).type() in (ss.FLOAT, ss.DOUBLE):
                return "rows"
        if distinct:
            mode = "distinct"
    return mode


DISTINCT_OF = "$distinct:"   # prefix of the union table's column that carries a DISTINCT aggregate's input values


def _distinct_blocks(group_by, spec, plain, shard_spec, with_residual, local_child, executor):
    """DISTINCT aggregates across shards.  The distinct values of a group over the whole job are the distinct values of the shards'
    distinct values: every shard sends, next to its partial table of the other aggregates, the (keys, value) pairs of each DISTINCT
    input (a GroupAggregate by keys + value), all stacked into ONE table -- a block's rows are NULL in the other blocks' columns,
    and every merging aggregate skips NULLs -- so one exchange and one merge plan serve: SUM of sums ... over the partial columns,
    the DISTINCT aggregates as written over the pair columns.  -> (this rank's stacked table, rows -> merge plan)."""
    keys = list(group_by)
    blocks = []      # (view, {union column name: column name in the view})
    if plain.elements:
        part = executor(ss.GroupAggregate(ss.ProjectNamedAttributes(keys), shard_spec, None, local_child))
        blocks.append((part, {part.schema().attribute(i).name(): part.schema().attribute(i).name() for i in range(part.schema().attribute_count())}))
    inputs = list(dict.fromkeys(e[3] for e in spec.elements if e[1]))
    for name in inputs:
        by = keys + ([name] if name not in keys else [])
        pairs = executor(ss.GroupAggregate(ss.ProjectNamedAttributes(by), ss.AggregationSpecification().AddAggregation(ss.COUNT, "", "$n"), None, local_child))
        cols = {k: k for k in keys}
        cols[DISTINCT_OF + name] = name
        blocks.append((pairs, cols))
    # the stacked table: keys, the partial table's columns, one column per DISTINCT input; every non-key column NULLABLE
    attrs, seen = [], set()
    for view, cols in blocks:
        for union_name, local_name in cols.items():
            if union_name in seen:
                continue
            seen.add(union_name)
            a = view.schema().attribute(view.schema().LookupAttributePosition(local_name))
            attrs.append(ss.Attribute(union_name, a.type(), a.nullability() if union_name in keys else ss.NULLABLE))
    total = sum(view.row_count() for view, _ in blocks)
    columns = []
    for a in attrs:
        is_string = a.type() == ss.STRING
        data = np.empty(total, dtype=object) if is_string else np.zeros(total, dtype=ss.numpy_dtype(a.type()))
        if is_string:
            data[:] = [b""] * total
        nulls = np.zeros(total, dtype=bool) if a.is_nullable() else None
        at = 0
        for view, cols in blocks:
            n = view.row_count()
            if a.name() in cols:
                col = view.column(view.schema().LookupAttributePosition(cols[a.name()]))
                data[at:at + n] = col.data
                if col.is_null is not None:
                    nulls[at:at + n] = col.is_null
            else:
                nulls[at:at + n] = True
            at += n
        columns.append(ss.Column(data, nulls))
    stacked = ss.View(ss.TupleSchema(attrs), columns, total)
    merged_spec, counts = ss.AggregationSpecification(), []
    for (aggregation, distinct, out_type, input_name, output_name) in spec.elements:
        if distinct:
            merged_spec.elements.append((aggregation, 1, out_type, DISTINCT_OF + input_name, output_name))
        elif aggregation == ss.COUNT:
            merged_spec.AddAggregation(ss.SUM, output_name, output_name)
            counts.append(output_name)
        else:
            merged_spec.AddAggregation(aggregation, output_name, output_name)
            if output_name in with_residual:
                merged_spec.AddAggregation(ss.SUM, output_name + RESIDUAL, output_name + RESIDUAL)
    # the columns of the merged GroupAggregate, in its order, for _merge_plan's projection (COUNT back to NOT NULL, sum + residual)
    single = ss.Plan(ss.GroupAggregate(ss.ProjectNamedAttributes(keys), spec, None, local_child), ss.Context(-1)).result_schema
    out_attrs = []
    for i in range(single.attribute_count()):
        out_attrs.append(single.attribute(i))
        if single.attribute(i).name() in with_residual:
            out_attrs.append(ss.Attribute(single.attribute(i).name() + RESIDUAL, ss.DOUBLE, ss.NULLABLE))
    out_schema = ss.TupleSchema(out_attrs)
    return stacked, (lambda rows: _merge_plan(group_by, merged_spec, counts, out_schema, rows))


def _merge_plan(group_by, merged_spec, counts, schema, everyone, valid=None):
    """The second, local GroupAggregate over the gathered partial tables (+ COUNT columns back to NOT NULL, and every DOUBLE
    sum = its merged sum + its merged residual -- each of the two is itself accumulated in double-double -- with the
    residual columns projected away).
    valid: name of a BOOL column of `everyone` marking the real rows (padding rows of fixed-size images are 0)."""
    source = ss.ScanView(everyone)
    if valid is not None:
        source = ss.Filter(ss.NamedAttribute(valid), ss.ProjectAllAttributes(), source)
    merged = ss.GroupAggregate(ss.ProjectNamedAttributes(list(group_by)), merged_spec, None, source)
    names = [schema.attribute(i).name() for i in range(schema.attribute_count())]
    residuals = set(n for n in names if n.endswith(RESIDUAL))
    if counts or residuals:
        # SUM(...) is NULLABLE, COUNT is not: restore the schema of the single-process result
        e = ss.CompoundExpression()
        for i in range(schema.attribute_count()):
            a = schema.attribute(i)
            if a.name() in residuals:
                continue
            if a.name() in counts:
                zero = ss.ConstUint64(0) if a.type() == ss.UINT64 else ss.ConstUint32(0)
                e.AddAs(a.name(), ss.IfNull(ss.NamedAttribute(a.name()), zero))
            elif a.name() + RESIDUAL in residuals:
                e.AddAs(a.name(), ss.Plus(ss.NamedAttribute(a.name()), ss.NamedAttribute(a.name() + RESIDUAL)))
            else:
                e.Add(ss.NamedAttribute(a.name()))
        merged = ss.Compute(e, merged)
    return merged


# ---------------------------------------------------------------------------------------------
# Sort across shards: sample sort with ONE all-to-all (BASELINE config #5 scaled out)
# ---------------------------------------------------------------------------------------------
_CONST_OF = {ss.INT32: ss.ConstInt32, ss.INT64: ss.ConstInt64, ss.UINT32: ss.ConstUint32, ss.UINT64: ss.ConstUint64,
             ss.FLOAT: ss.ConstFloat, ss.DOUBLE: ss.ConstDouble, ss.DATE: ss.ConstDate, ss.DATETIME: ss.ConstDateTime}


def _choose_splitters(samples, world):
    """world - 1 ascending splitters from the gathered, sorted sample of first-key values (NaNs, which sort last in
    numpy and compare false with everything, are never splitters)."""
    if samples.dtype.kind == "f":
        samples = samples[~np.isnan(samples)]
    if len(samples) == 0:
        return []
    return [samples[min(len(samples) - 1, (i + 1) * len(samples) // world)] for i in range(world - 1)]


def _range_predicate(key, dtype, nullable, splitters, d, world, descending):
    """Rows whose first sort key belongs to destination rank d.  Ascending: rank 0 takes the
    smallest values and the NULLs (NULLs sort first, sort.cc:205-238); descending: mirrored, NULLs
    last.  All rows with equal first-key values land on the same rank, so the remaining keys and
    the stable tie order are settled by the destination's local sort."""
    const = _CONST_OF[dtype]
    b = (world - 1 - d) if descending else d          # bucket index in ascending value order
    pred = None
    if splitters:
        if b == world - 1:
            # the last bucket is the complement of the others, not "key > splitter": every comparison with a NaN is
            # false, so a NaN key would match no bucket and the row would vanish from the global result
            pred = ss.Not(ss.LessOrEqual(ss.NamedAttribute(key), const(splitters[b - 1])))
        else:
            if b > 0:
                pred = ss.Greater(ss.NamedAttribute(key), const(splitters[b - 1]))
            hi = ss.LessOrEqual(ss.NamedAttribute(key), const(splitters[b]))
            pred = hi if pred is None else ss.And(pred, hi)
    elif b != 0:
        pred = ss.ConstBool(False)                      # no non-NULL value anywhere: one bucket holds everything
    if pred is None:
        pred = ss.ConstBool(True)
    if nullable:
        null_rank = world - 1 if descending else 0
        isnull = ss.IsNull(ss.NamedAttribute(key))
        pred = ss.Or(isnull, pred) if d == null_rank else ss.And(ss.Not(isnull), pred)
    return pred


def _all_to_all_views(parts, schema, group, device):
    """parts[d] = this rank's rows for rank d (same schema).  Returns the rows every rank sent to
    this one, concatenated in source-rank order.  One all-gather of the count matrix, then one
    all_to_all_single per column buffer (values, NULL mask)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    send = torch.tensor([p.row_count() for p in parts], dtype=torch.int64, device=device)
    matrix = [torch.zeros(world, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(matrix, send, group=group)
    recv = [int(matrix[src][rank].item()) for src in range(world)]
    sent = [int(x) for x in send.tolist()]
    cols = []
    for i in range(schema.attribute_count()):
        np_dtype = parts[0].column(i).data.dtype
        if schema.attribute(i).type() == ss.STRING:
            # host Views hold the byte strings themselves: every rank publishes its per-destination lists and keeps its own
            mine = [[bytes(v) if v is not None else b"" for v in p.column(i).data] for p in parts]
            everyone = _gather_objects(mine, group)
            data = np.empty(sum(recv), dtype=object)
            data[:] = [v for src in range(world) for v in everyone[src][rank]]
            nulls = None
            if schema.attribute(i).is_nullable():
                zmine = [(p.column(i).is_null if p.column(i).is_null is not None else np.zeros(p.row_count(), bool)).tolist() for p in parts]
                zs = _gather_objects(zmine, group)
                nulls = np.array([z for src in range(world) for z in zs[src][rank]], dtype=bool)
            cols.append(ss.Column(data, nulls))
            continue
        bufs = []
        for which in (0, 1):
            if which == 1 and not schema.attribute(i).is_nullable():
                bufs.append(None)
                continue
            dt = np_dtype if which == 0 else np.dtype(np.bool_)
            pieces = []
            for p in parts:
                arr = p.column(i).data if which == 0 else p.column(i).is_null
                if arr is None:
                    arr = np.zeros(p.row_count(), dtype=np.bool_)
                pieces.append(np.ascontiguousarray(arr).view(np.uint8).reshape(-1))
            out_bytes = np.concatenate(pieces) if pieces else np.zeros(0, np.uint8)
            t_in = torch.from_numpy(out_bytes.copy()).to(device)
            t_out = torch.empty(sum(recv) * dt.itemsize, dtype=torch.uint8, device=device)
            dist.all_to_all_single(t_out, t_in, [r * dt.itemsize for r in recv], [s * dt.itemsize for s in sent], group=group)
            bufs.append(t_out.cpu().numpy().view(dt))
        cols.append(ss.Column(bufs[0], bufs[1]))
    return ss.View(schema, cols, sum(recv))


def sharded_sort(sort_order, local_child, executor, group=None, device="cpu", samples_per_rank=256):
    """Sort(sort_order, <all shards of local_child>) as a sample sort: local sort, splitters from a
    regular sample of the first key, ONE all-to-all of the rows, local sort of what arrived.

    Returns this rank's slice of the globally sorted rows: the concatenation of the results in
    rank order equals the single-process Sort (same stable order of ties when the shards are
    contiguous row ranges in rank order).  Rows are partitioned by the first key only, so a first
    key with few distinct values balances poorly (but stays correct)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    local = executor(ss.Sort(sort_order, None, 0, local_child))
    if world == 1:
        return local
    schema = local.schema()
    key, order = sort_order.keys[0]
    pos = [schema.attribute(i).name() for i in range(schema.attribute_count())].index(key)
    attr = schema.attribute(pos)
    if attr.type() not in _CONST_OF:
        raise ss.SupersonicException(ss.ERROR_NOT_IMPLEMENTED, "first sort key type cannot be range-partitioned across shards yet")
    descending = order == ss.DESCENDING
    # regular sample of the non-NULL first-key values of the (already sorted) shard
    col = local.column(pos)
    vals = col.data if col.is_null is None else col.data[~col.is_null]
    take = min(samples_per_rank, len(vals))
    sample = np.zeros(samples_per_rank, dtype=vals.dtype)
    if take:
        sample[:take] = vals[(np.arange(take) * len(vals)) // take]
    raw = torch.from_numpy(np.concatenate([np.array([take], dtype=np.int64).view(np.uint8), sample.view(np.uint8)]).copy()).to(device)
    gathered = [torch.empty_like(raw) for _ in range(world)]
    dist.all_gather(gathered, raw, group=group)
    everyone = []
    for g in gathered:
        b = g.cpu().numpy()
        n = int(b[:8].view(np.int64)[0])
        everyone.append(b[8:].view(vals.dtype)[:n])
    allv = np.sort(np.concatenate(everyone)) if everyone else np.zeros(0, vals.dtype)
    splitters = [v.item() for v in _choose_splitters(allv, world)]
    parts = [executor(ss.Filter(_range_predicate(key, attr.type(), attr.is_nullable(), splitters, d, world, descending),
                                ss.ProjectAllAttributes(), ss.ScanView(local))) for d in range(world)]
    arrived = _all_to_all_views(parts, schema, group, device)
    return executor(ss.Sort(sort_order, None, 0, ss.ScanView(arrived)))


class _DevPtr(object):
    """A raw device pointer as something torch.as_tensor can wrap without a copy."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _bytes_over(torch, device, ptr, nbytes):
    if nbytes == 0 or not ptr:
        return torch.empty(0, dtype=torch.uint8, device=device)
    return torch.as_tensor(_DevPtr(ptr, nbytes, "|u1"), device=device)


def device_sharded_sort(ctx, sort_order, local_view, group=None, samples_per_rank=256, always_exchange=False):
    """sharded_sort with every row staying in HBM: the local sorts and the range Filters run as device
    plans, their result buffers are wrapped as torch tensors (no copy) and exchanged with ONE RCCL
    all_to_all_single per column buffer, and the arrivals are sorted where they land.

    local_view: this rank's shard (host View or DeviceView).  Returns (plan, DeviceView): the rank's
    slice of the global order as device columns owned by `plan` (fetch with plan.fetch())."""
    import torch
    dist = _dist_for(group)

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = torch.device("cuda", torch.cuda.current_device())
    schema = local_view.schema()
    n_attrs = schema.attribute_count()
    widths = [4 if schema.attribute(i).type() == ss.STRING else np.dtype(ss.numpy_dtype(schema.attribute(i).type())).itemsize
              for i in range(n_attrs)]
    strings = None
    if _has_strings(schema):
        # STRING columns travel as the INT32 codes of ONE dictionary that every rank builds identically
        if not isinstance(local_view, ss.View):
            raise ss.SupersonicException(ss.ERROR_NOT_IMPLEMENTED, "STRING columns of a DeviceView carry their caller's dictionary: shard host Views")
        strings = job_strings(ss.collect_strings(ss.ScanView(local_view)), group)
    first = ss.Plan(ss.Sort(sort_order, None, 0, ss.ScanView(local_view)), ctx, strings)
    first.run()
    ctx.synchronize()
    local = first.result_device_view()
    if world == 1 and not always_exchange:
        return first, local
    key, order = sort_order.keys[0]
    pos = [schema.attribute(i).name() for i in range(n_attrs)].index(key)
    attr = schema.attribute(pos)
    if attr.type() not in _CONST_OF:
        raise ss.SupersonicException(ss.ERROR_NOT_IMPLEMENTED, "first sort key type cannot be range-partitioned across shards yet")
    descending = order == ss.DESCENDING
    np_dt = np.dtype(ss.numpy_dtype(attr.type()))
    rows = local.row_count()
    # the NULLs of the first key are one contiguous run of the sorted shard (front for ASCENDING, back for DESCENDING)
    n_null = 0
    if attr.is_nullable() and rows and local._ptrs[pos][1]:
        n_null = int(_bytes_over(torch, device, local._ptrs[pos][1], rows).sum(dtype=torch.int64).item())
    lo = 0 if descending else n_null
    n_val = rows - n_null
    take = min(samples_per_rank, n_val)
    sample = torch.zeros(samples_per_rank * np_dt.itemsize + 8, dtype=torch.uint8, device=device)
    sample[:8] = torch.tensor([take], dtype=torch.int64, device=device).view(torch.uint8)
    if take:
        keys = _bytes_over(torch, device, local._ptrs[pos][0], rows * np_dt.itemsize).view(-1, np_dt.itemsize)
        idx = lo + (torch.arange(take, device=device, dtype=torch.int64) * n_val) // take
        sample[8:8 + take * np_dt.itemsize] = keys[idx].reshape(-1)
    gathered = [torch.empty_like(sample) for _ in range(world)]
    dist.all_gather(gathered, sample, group=group)
    everyone = []
    for g in gathered:
        b = g.cpu().numpy()
        everyone.append(b[8:].view(np_dt)[: int(b[:8].view(np.int64)[0])])
    allv = np.sort(np.concatenate(everyone))
    splitters = [v.item() for v in _choose_splitters(allv, world)]
    # one range Filter per destination, straight over the sorted shard's device columns
    parts = []
    for d in range(world):
        pl = ss.Plan(ss.Filter(_range_predicate(key, attr.type(), attr.is_nullable(), splitters, d, world, descending),
                               ss.ProjectAllAttributes(), ss.ScanView(local)), ctx, strings)
        pl.run()
        parts.append((pl, pl.result_device_view()))
    ctx.synchronize()
    send = torch.tensor([v.row_count() for (_p, v) in parts], dtype=torch.int64, device=device)
    matrix = [torch.zeros(world, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(matrix, send, group=group)
    recv = [int(matrix[src][rank].item()) for src in range(world)]
    sent = [int(x) for x in send.tolist()]
    arrived, keep = [], []
    for i in range(n_attrs):
        ptrs = []
        for which, w in ((0, widths[i]), (1, 1)):
            if which == 1 and not schema.attribute(i).is_nullable():
                ptrs.append(0)
                continue
            pieces = [_bytes_over(torch, device, v._ptrs[i][which], v.row_count() * w) if v._ptrs[i][which]
                      else torch.zeros(v.row_count() * w, dtype=torch.uint8, device=device) for (_p, v) in parts]
            t_in = torch.cat(pieces) if pieces else torch.empty(0, dtype=torch.uint8, device=device)
            t_out = torch.empty(max(sum(recv) * w, 1), dtype=torch.uint8, device=device)
            dist.all_to_all_single(t_out[: sum(recv) * w], t_in, [r * w for r in recv], [s * w for s in sent], group=group)
            keep.append(t_out)
            ptrs.append(t_out.data_ptr())
        arrived.append((ptrs[0], ptrs[1]))
    torch.cuda.synchronize()
    final = ss.Plan(ss.Sort(sort_order, None, 0, ss.ScanView(ss.DeviceView(schema, arrived, sum(recv)))), ctx, strings)
    final.run()
    ctx.synchronize()
    del keep, parts
    return final, final.result_device_view()


class DeviceShardedGroupAggregate(object):
    """GroupAggregate over row-range shards with ONE RCCL collective per step (BASELINE config #4).

    Per step, all on the device and in stream order: the shard's GroupAggregate plan -> its partial
    table packed into one image (`Plan.pack_image`, row count in the header) -> ONE
    `all_gather_into_tensor` of the images -> `Plan.unpack_images` (contiguous columns + a validity
    column) -> the merge plan (GroupAggregate of the merge functions under Filter(__valid)), which
    was created once and is reused.  No device value is read on the host between the two plans'
    own runs; `check()` reads the 32-byte trailer once, after the caller's last step.

    The image capacity is agreed on at the first step (one extra all-reduce, set-up only) and
    regrown by `check()` if a later shard outgrows it.  `collectives` counts the collectives the
    most recent step issued (asserted to be 1 in the tests)."""

    def __init__(self, ctx, group_by, spec, local_child, group=None, capacity_rows=0, exchange="all_gather"):
        """exchange = "all_gather": every rank ends with the FULL result (every partial table goes everywhere, every rank
        merges all of them).  exchange = "key_range": a partial row goes only to the owner of its key (ssgpu_result_route_images
        + ONE all_to_all_single of equally sized images), every rank merges -- and ends with -- the groups it owns: merge
        work and link traffic per rank shrink with the world size (`gather_result()` collects the full table when wanted)."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, _dist_for(group)
        self.ctx, self.group = ctx, group
        assert exchange in ("all_gather", "key_range")
        self.exchange = exchange
        self.world = dist.get_world_size(group)
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.group_by = list(group_by)
        child_schema = _schema_of(local_child)
        shard_spec, with_residual = _shard_spec(spec, child_schema)
        self.merged_spec, self.counts = _merge_spec(spec, with_residual)
        self.never_null = _never_null(shard_spec, child_schema)
        op = ss.GroupAggregate(ss.ProjectNamedAttributes(self.group_by), shard_spec, None, local_child)
        # STRING keys / MIN / MAX results travel as the INT32 codes of ONE dictionary that every rank builds identically
        # (decided by the result schema, which is the same on all ranks: every rank issues the same collectives)
        self.strings = None
        probe = ss.Plan(op, ctx)
        if _has_strings(probe.result_schema) or _has_strings(probe.input.schema()):
            self.strings = job_strings(ss.collect_strings(op), group)
            probe = ss.Plan(op, ctx, self.strings)
        self.first = probe
        # a step never waits for the host; the job keeps its shard's columns alive (ssgpu.h: INPUT LIFETIME) -- an option of the job's
        # OWN plans: other plans on the (possibly shared) context keep the default contract
        self.first.set_option("lazy_feedback", 1)
        self.capacity = int(capacity_rows)
        self.merge = None
        self.collectives = 0
        self.setup_collectives = 0
        # kernels and the collective are ordered by stream: the library launches on the ctx stream, RCCL on torch's
        raw = ctx.stream()                                   # None / 0: the library launches on the legacy default stream
        self._lib_stream = torch.cuda.ExternalStream(raw) if raw else torch.cuda.default_stream(self.device)

    def _allocate(self):
        torch = self.torch
        self.image_bytes, self.unpacked_bytes, _offs = self.first.image_layout(self.capacity, self.world)
        # all_gather: one image out, world images in; key_range: world images out (one per destination), world images in
        n_out = self.world if self.exchange == "key_range" else 1
        self.image = torch.empty(n_out * self.image_bytes, dtype=torch.uint8, device=self.device)
        self.images = torch.empty(self.world * self.image_bytes, dtype=torch.uint8, device=self.device)
        self.unpacked = torch.empty(self.unpacked_bytes, dtype=torch.uint8, device=self.device)
        self.merge = None

    def _capacity_for(self, rows):
        if self.exchange == "key_range":     # rows one destination receives from one source: its share of the keys, with head room for the hash's spread
            return max(1024, (int(rows * 1.3 / self.world) + 2047) // 1024 * 1024)
        return max(1024, (int(rows) * 5 // 4 + 1023) // 1024 * 1024)

    def _agree_capacity(self, failed=0):
        """Set-up only: the largest partial table of any rank, with head room (one all-reduce, one host read)."""
        torch = self.torch
        rows = torch.tensor([0 if failed else self.first.lib.ssgpu_result_row_count(self.first._result)], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(rows, op=self.dist.ReduceOp.MAX, group=self.group)
        self.setup_collectives += 1
        self.capacity = self._capacity_for(int(rows.item()))

    def _failed_images(self, code):
        """This rank's shard run failed: its images leave EMPTY and flagged with the return code (header word 5), so that the
        step's collective still happens and every rank's check() raises -- a rank that returned here would leave the others
        waiting in the all-to-all / all-gather forever."""
        torch = self.torch
        header = torch.tensor([0, self.capacity, 0, 0, 0, int(code), 0, 0], dtype=torch.int64).view(torch.uint8).to(self.device)
        n_out = self.world if self.exchange == "key_range" else 1
        for d in range(n_out):
            self.image[d * self.image_bytes:d * self.image_bytes + 64].copy_(header)

    def step(self, view=None):
        torch = self.torch
        self.collectives = 0
        failed = 0
        try:
            self.first.run(view)
        except ss.SupersonicException as e:
            failed = e.return_code if e.return_code > 0 else ss.ERROR_UNKNOWN
        if not self.capacity:
            self._agree_capacity(failed)
        if getattr(self, "_cap_alloc", None) != self.capacity:
            self._allocate()
            self._cap_alloc = self.capacity
        cur = torch.cuda.current_stream(self.device)
        if failed:
            self._failed_images(failed)
        elif self.exchange == "key_range":
            self.first.route_images(len(self.group_by), self.world, self.capacity, self.image.data_ptr())
        else:
            self.first.pack_image(self.capacity, self.image.data_ptr())
        cur.wait_stream(self._lib_stream)                  # the collective reads what the routing / pack kernels wrote
        if self.exchange == "key_range":
            self.dist.all_to_all_single(self.images, self.image, group=self.group)     # image d -> rank d, equal sizes
        else:
            self.dist.all_gather_into_tensor(self.images, self.image, group=self.group)
        self.collectives += 1
        self._lib_stream.wait_stream(cur)
        everyone = _declare_not_null(self.first.unpack_images(self.images.data_ptr(), self.world, self.capacity, self.unpacked.data_ptr()), self.never_null)
        if self.merge is None:
            self.merge = ss.Plan(_merge_plan(self.group_by, self.merged_spec, self.counts, self.first.result_schema, everyone,
                                             valid="__valid"), self.ctx, self.strings)
            self.merge.set_option("lazy_feedback", 1)
        self.merge.run(everyone)
        return self.merge

    def check(self):
        """After the last step: True if every image fitted; otherwise the capacity was regrown (all ranks
        agree: they fold the same headers) and the step has to be repeated.  An evaluation error of any
        shard's run surfaces here."""
        self.ctx.synchronize()
        self.torch.cuda.synchronize(self.device)
        t = self.unpacked[self.unpacked_bytes - 32:].view(self.torch.int64).tolist()
        if self.exchange == "key_range" and self.world > 1:   # every rank unpacked other images: the failure / error word as well as the capacity verdict below
            w = self.torch.tensor([int(t[3])], dtype=self.torch.int64, device=self.device)
            self.dist.all_reduce(w, op=self.dist.ReduceOp.MAX, group=self.group)
            self.setup_collectives += 1
            t[3] = int(w.item())
        if t[3] >> 8:
            raise ss.SupersonicException(t[3] >> 8, "the GroupAggregate of a rank's shard failed (return code %d)" % (t[3] >> 8))
        if t[3] & 0xFF:
            raise ss.SupersonicException(ss.ERROR_EVALUATION_ERROR, "Evaluation error in a shard's GroupAggregate")
        if self.exchange == "key_range" and self.world > 1:
            # every rank saw other images: the verdict and the new capacity have to be the same everywhere (one tiny
            # all-reduce, after the timed steps only)
            v = self.torch.tensor([int(t[2] != 0), int(t[0])], dtype=self.torch.int64, device=self.device)
            self.dist.all_reduce(v, op=self.dist.ReduceOp.MAX, group=self.group)
            self.setup_collectives += 1
            t[2], t[0] = int(v[0].item()), int(v[1].item())
        if t[2]:
            self.capacity = max(self.capacity, (int(t[0]) * 5 // 4 + 1023) // 1024 * 1024) if self.exchange == "key_range" else \
                max(1024, (int(t[0]) * 5 // 4 + 1023) // 1024 * 1024)
            return False
        return True

    def result(self):
        """The merge plan and its result as device columns: the FULL table (all_gather) or this rank's groups (key_range)."""
        return self.merge, self.merge.result_device_view()

    def gather_result(self):
        """key_range: the full table on every rank -- the finished slices all-gathered through the host (not part of a step)."""
        local = self.merge.fetch()
        return local if self.exchange != "key_range" else _all_gather_view(local, self.group, self.device)


class PlanDenseBackend(object):
    """The device side of DenseShardedGroupAggregate: ONE plan -- the job's GroupAggregate itself, no shard / merge pair --
    run through ssgpu_plan_run_dense / ssgpu_plan_fold_dense, buffers as torch tensors on the plan's device."""

    def __init__(self, ctx, op, strings=None):
        import torch
        self.torch = torch
        ctx.set_option("group_dense", 1)
        self.ctx = ctx
        self.plan = ss.Plan(op, ctx, strings) if strings is not None else ss.Plan(op, ctx)
        self.plan.set_option("lazy_feedback", 1)      # (the job keeps its shard's columns alive between steps)
        self.device = torch.device("cuda", torch.cuda.current_device())
        raw = ctx.stream()
        self._lib_stream = torch.cuda.ExternalStream(raw) if raw else torch.cuda.default_stream(self.device)

    def key_ranges(self, view):
        return self.plan.key_ranges(view)

    def set_dense(self, ranges, n_chunks):
        return self.plan.set_dense(ranges, n_chunks)

    def alloc(self, nbytes):
        return self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device)

    def run_dense(self, view, table):
        self.plan.run_dense(view, table.data_ptr())

    def before_collective(self):      # the collective (torch's stream) reads what the library's stream wrote
        self.torch.cuda.current_stream(self.device).wait_stream(self._lib_stream)

    def after_collective(self):
        self._lib_stream.wait_stream(self.torch.cuda.current_stream(self.device))

    def fold_dense(self, chunks, n_chunks):
        self.plan.fold_dense(chunks.data_ptr(), n_chunks)

    def dense_flags(self):
        return self.plan.dense_flags()

    def dense_grow(self):
        self.plan.dense_grow()

    def dense_fail(self, table, code):
        self.plan.dense_fail(table.data_ptr(), code)

    def local_result(self):
        return self.plan.fetch()


class DenseShardedGroupAggregate(object):
    """GroupAggregate over row-range shards through DENSE SLOTS (include/ssgpu.h "dense-slot GroupAggregate
    across ranks"): the ranks agree once on the value ranges of the group keys; from then on every rank's partial table is
    the same array -- slot = the keys' mixed-radix number -- and a step is

        shard scan into the table (chunk r = the slots rank r owns)  ->  ONE all_to_all_single of the chunks  ->
        element-wise fold of the world images of the owned slot range + extraction

    No merge plan, no routing, no packing; DOUBLE sums cross as raw (hi, lo) accumulator pairs and are rounded once.  Every
    rank ends with the groups it owns (`gather_result()` collects the table).  What can go wrong in a step travels in the
    chunks' headers, which reach EVERY rank: a rank whose shard overflowed a segment, met a key outside the ranges or hit an
    evaluation error still sends its chunks, flagged -- nobody leaves a step early, and `check()` gives every rank the same
    verdict without a further collective: False = repeat the step (segments were enlarged / the ranks agreed on wider ranges).
    `backend`: PlanDenseBackend on a GPU; the CPU tests pass a host restatement of the table with the same interface.
    Raises SupersonicException(NOT_IMPLEMENTED / INVALID_ARGUMENT_VALUE) from `setup()` when the plan or the ranges do not
    take this form -- identically on every rank (same plan, same agreed ranges) -- and the caller uses the image exchange."""

    def __init__(self, backend, group=None):
        import torch.distributed as dist
        self.dist = _dist_for(group)      # (host tensors of the CPU tests pass through it unchanged)
        self.backend, self.group = backend, group
        self.world = dist.get_world_size(group)
        self.layout = None
        self.collectives = 0
        self.setup_collectives = 0
        self.ranges = None
        self._view = None

    def setup(self, view):
        """Set-up (and again when a step met a key outside the ranges): every rank's key ranges, united -- one all_gather_object."""
        mine = self.backend.key_ranges(view)
        everyone = [None] * self.world
        self.dist.all_gather_object(everyone, [list(r) for r in mine], group=self.group)
        self.setup_collectives += 1
        ranges = [(min(r[k][0] for r in everyone), max(r[k][1] for r in everyone)) for k in range(len(mine))]
        if self.ranges is not None:            # never narrower than before: a plan that alternates between inputs settles
            ranges = [(min(a[0], b[0]), max(a[1], b[1])) for a, b in zip(ranges, self.ranges)]
        self.ranges = ranges
        self.layout = self.backend.set_dense(ranges, self.world)
        nbytes = self.world * int(self.layout["chunk_bytes"])
        self.table = self.backend.alloc(nbytes)
        self.chunks = self.backend.alloc(nbytes)

    def step(self, view=None):
        self._view = view
        self.collectives = 0
        if self.layout is None:
            self.setup(view)
        try:
            self.backend.run_dense(view, self.table)
        except ss.SupersonicException as e:
            # this rank's shard failed (memory quota, interrupt ...): it still takes part in the collective -- its chunks leave flagged
            # with the code, every rank's check() raises it; returning here would leave the other ranks waiting in the all-to-all
            self.backend.dense_fail(self.table, e.return_code if e.return_code > 0 else ss.ERROR_UNKNOWN)
        self.backend.before_collective()
        self.dist.all_to_all_single(self.chunks, self.table, group=self.group)     # chunk r of every rank -> rank r
        self.collectives += 1
        self.backend.after_collective()
        self.backend.fold_dense(self.chunks, self.world)
        return self.backend

    def check(self):
        flags, error = self.backend.dense_flags()      # the OR over every rank's headers: the same words on every rank
        if flags & 8:
            raise ss.SupersonicException(flags >> 8, "the GroupAggregate of a rank's shard failed (return code %d)" % (flags >> 8))
        if error:
            raise ss.SupersonicException(ss.ERROR_EVALUATION_ERROR, "Evaluation error in a shard's GroupAggregate")
        if flags & 4:                                   # some rank met a key outside the ranges: agree on wider ones
            self.setup(self._view)
            return False
        if flags & 2:                                   # some rank's record segments ran full: every rank enlarges its own
            self.backend.dense_grow()
            return False
        if flags & 1:
            raise ss.SupersonicException(ss.ERROR_MEMORY_EXCEEDED, "a dense partition outgrew its table")
        return True

    def gather_result(self):
        """The full table on every rank: the owners' finished slices concatenated in rank order (not part of a step)."""
        local = self.backend.local_result()
        device = getattr(self.backend, "device", None)
        return _all_gather_view(local, self.group, device if device is not None else "cpu")


def device_sharded_group_aggregate(ctx, group_by, spec, local_child, group=None):
    """One-shot form of DeviceShardedGroupAggregate: returns (plan, DeviceView) -- the full result on every
    rank, as device columns owned by `plan`."""
    job = DeviceShardedGroupAggregate(ctx, group_by, spec, local_child, group)
    job.step()
    while not job.check():
        job.step()
    plan, view = job.result()
    plan._sharded_job = job     # the merge plan reads the job's buffers: keep them alive with it
    return plan, view
