"""The dense-slot exchange of a sharded GroupAggregate on CPU processes (gloo, world_size 2): SURVEY 8(e), BASELINE config #4.

`supersonic_amd.distributed.DenseShardedGroupAggregate` is the protocol -- key ranges agreed once (one all_gather_object),
then per step: shard table -> ONE all_to_all_single of slot slices -> element-wise fold -> the owner's groups; flags of
every rank (a key outside the ranges, a failed shard run, an evaluation error) ride in the chunk headers, so every rank
reaches the same verdict and nobody leaves a step while the others wait in the collective.  No kernel can run here: the
backend handed to it is a host restatement of the device table (`HostDenseTable`, below) whose LAYOUT -- slots,
partitions, entries per partition, slots per chunk -- comes from the real `ssgpu_plan_set_dense` on a device-less context
and whose partial aggregates come from the CPU oracle.  What is under test: the range union across ranks, slot ownership
(slot -> partition -> chunk, the same arithmetic as csrc/launch.h: ssgpu_dense_entry), the collective on two real ranks,
the fold, and the three repeat / failure flows.  The device backend (PlanDenseBackend) runs the same protocol class over
RCCL: tests/test_00_configs_gpu.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import supersonic_amd as ss
from supersonic_amd.distributed import DenseShardedGroupAggregate
from oracle import oracle
from helpers import sort_rows, assert_cols_equal

NA = ss.NamedAttribute
KEYS = ["k1", "k2"]                 # k1: NULLABLE INT32 (33 packed bits), k2: BOOL -- together one 64-bit key word
KEY_SIGNED = [True, False]          # the ranges' order-preserving unsigned domain: signed columns with their sign bit flipped (ssgpu.h)
KEY_NULLABLE = [True, False]
SIGN = 1 << 63


def make_view(n, seed=11, k1_lo=0, k1_hi=40):
    rng = np.random.default_rng(seed)
    schema = ss.TupleSchema([ss.Attribute("a", ss.INT64), ss.Attribute("k1", ss.INT32, ss.NULLABLE), ss.Attribute("k2", ss.BOOL),
                             ss.Attribute("v", ss.INT64, ss.NULLABLE), ss.Attribute("d", ss.DOUBLE)])
    return ss.View(schema, [rng.integers(0, 1000, n), ss.Column(rng.integers(k1_lo, k1_hi, n).astype(np.int32), rng.random(n) < 0.1),
                            rng.integers(0, 2, n).astype(bool),
                            ss.Column(rng.integers(-1000, 1000, n), rng.random(n) < 0.3),
                            rng.integers(-4000, 4000, n) * 0.25])


def spec():
    return (ss.AggregationSpecification().AddAggregation(ss.SUM, "v", "sv").AddAggregation(ss.MIN, "v", "mnv")
            .AddAggregation(ss.MAX, "d", "mxd").AddAggregation(ss.SUM, "d", "sd")
            .AddAggregation(ss.COUNT, "v", "cv").AddAggregation(ss.COUNT, "", "n"))


MERGE = ["sum", "min", "max", "sum", "sum", "sum"]       # how a partial column of spec() combines across shards


def job_op(view, with_filter=True):
    child = ss.ScanView(view)
    if with_filter:
        child = ss.Filter(ss.Greater(NA("a"), ss.ConstInt64(299)), ss.ProjectAllAttributes(), child)
    return ss.GroupAggregate(ss.ProjectNamedAttributes(KEYS), spec(), None, child)


class HostDenseTable(object):
    """Host restatement of the device's chunked dense table (include/ssgpu.h: header | slots), one 8-byte word + one NULL byte
    per aggregate and slot, an `occupied` byte per slot.  Layout numbers from the REAL ssgpu_plan_set_dense."""

    def __init__(self, make_op, fail_at_step=None, fail_code=ss.ERROR_MEMORY_EXCEEDED):
        self.make_op = make_op
        self.ctx = ss.Context(-1)                         # bind-only: layout and binding need no GPU
        self.ctx.set_option("group_dense", 1)
        self.plan = None
        self.step_no = 0
        self.fail_at_step, self.fail_code = fail_at_step, fail_code
        self.device = None

    @staticmethod
    def _ordered(col, k):
        if not KEY_SIGNED[k]:
            return col.astype(np.uint64)
        return (col.astype(np.int64).astype(np.uint64)) ^ np.uint64(SIGN)      # INT32 keys: sign-extended, sign bit flipped

    def key_ranges(self, view):
        out = []
        for k, name in enumerate(KEYS):
            c = view.column(view.schema().LookupAttributePosition(name))
            live = c.data if c.is_null is None else c.data[~c.is_null]
            o = self._ordered(live, k)
            out.append((int(o.min()), int(o.max())) if len(o) else ((1 << 64) - 1, 0))
        return out

    def set_dense(self, ranges, n_chunks):
        self.plan = ss.Plan(self.make_op(make_view(8)), self.ctx)            # (any input of the job's schema: only the plan's shape matters)
        self.layout = self.plan.set_dense(ranges, n_chunks)                  # raises like the device plan for unusable ranges
        self.ranges, self.n_chunks = ranges, n_chunks
        self.spans = []
        for (lo, hi), name, nullable in zip(ranges, KEYS, KEY_NULLABLE):
            self.spans.append((hi - lo + 1 if hi >= lo else 0) + (1 if nullable else 0))
        self.n_aggs = len(MERGE)
        s1 = int(self.layout["chunk_slots"]) + 1
        self.chunk_host_bytes = (64 + s1 * (1 + 9 * self.n_aggs) + 63) // 64 * 64
        return {"chunk_bytes": self.chunk_host_bytes, "slots": self.layout["slots"], "n_parts": self.layout["n_parts"]}

    def alloc(self, nbytes):
        return torch.zeros(nbytes, dtype=torch.uint8)

    def _slot_of(self, idx):
        """dense index -> (chunk, slot in chunk): partition = idx % n_parts, entry = idx // n_parts (csrc/launch.h)"""
        npart, cap = int(self.layout["n_parts"]), int(self.layout["part_cap"])
        ppc = npart // self.n_chunks
        part, entry = idx % npart, idx // npart
        return part // ppc, (part % ppc) * cap + entry

    def _chunk_views(self, buf, c):
        s1 = int(self.layout["chunk_slots"]) + 1
        raw = buf.numpy()[c * self.chunk_host_bytes:(c + 1) * self.chunk_host_bytes]
        header = raw[:64].view(np.uint32)
        occ = raw[64:64 + s1]
        vals = raw[64 + s1:64 + s1 + 8 * self.n_aggs * s1].view(np.uint64).reshape(self.n_aggs, s1)
        nulls = raw[64 + s1 + 8 * self.n_aggs * s1:64 + s1 + 9 * self.n_aggs * s1].reshape(self.n_aggs, s1)
        return header, occ, vals, nulls

    def run_dense(self, view, table):
        self.step_no += 1
        if self.fail_at_step is not None and self.step_no == self.fail_at_step:
            raise ss.SupersonicException(self.fail_code, "this rank's shard run failed (test)")
        table.zero_()
        schema, cols = oracle.run(self.make_op(view))                      # this shard's partial table, by the CPU oracle
        nk = len(KEYS)
        idx = np.zeros(len(cols[0][0]), dtype=np.int64)
        miss = np.zeros(len(idx), dtype=bool)
        stride = 1
        for k in range(nk - 1, -1, -1):
            data, z = cols[k]
            lo, hi = self.ranges[k]
            o = self._ordered(data, k)
            off = (o - np.uint64(lo)).astype(np.int64) if hi >= lo else np.zeros(len(idx), np.int64)
            isnull = z if z is not None else np.zeros(len(idx), bool)
            bad = ~isnull & (((o < np.uint64(lo)) | (o > np.uint64(hi))) if hi >= lo else ~isnull)
            miss |= bad
            off = np.where(isnull, self.spans[k] - 1, np.where(bad, 0, off))
            idx += off * stride
            stride *= self.spans[k]
        flags = 4 if miss.any() else 0
        for c in range(self.n_chunks):
            header, _o, _v, _n = self._chunk_views(table, c)
            header[0], header[1] = flags, 0
        for row in np.nonzero(~miss)[0]:
            c, s = self._slot_of(int(idx[row]))
            _h, occ, vals, nulls = self._chunk_views(table, c)
            occ[s] = 1
            for j in range(self.n_aggs):
                data, z = cols[nk + j]
                nulls[j, s] = 1 if (z is not None and z[row]) else 0
                vals[j, s] = np.array([data[row]]).view(np.uint64)[0] if data.dtype.itemsize == 8 else np.uint64(int(data[row]))
        self._schema, self._dtypes = schema, [cols[nk + j][0].dtype for j in range(self.n_aggs)]

    def dense_fail(self, table, code):
        for c in range(self.n_chunks):
            header, _o, _v, _n = self._chunk_views(table, c)
            header[0], header[1] = 8 | (int(code) << 8), 0

    def before_collective(self):
        pass

    def after_collective(self):
        pass

    def fold_dense(self, chunks, n):
        s1 = int(self.layout["chunk_slots"]) + 1
        self.flags = self.error = 0
        failed = 0
        occ = np.zeros(s1, bool)
        vals = [None] * self.n_aggs
        nulls = [np.ones(s1, bool) for _ in range(self.n_aggs)]
        for c in range(n):
            header, o, v, z = self._chunk_views(chunks, c)
            self.flags |= int(header[0]) & 0xFF
            failed = max(failed, int(header[0]) >> 8)
            self.error |= int(header[1])
            occ |= o.astype(bool)
            for j, how in enumerate(MERGE):
                dt = getattr(self, "_dtypes", [np.dtype(np.int64)] * self.n_aggs)[j]
                x = v[j].view(dt if dt.itemsize == 8 else np.uint64).copy()
                present = o.astype(bool) & ~z[j].astype(bool)
                if vals[j] is None:
                    vals[j] = np.zeros(s1, dtype=x.dtype)
                both = present & ~nulls[j]
                first = present & nulls[j]
                with np.errstate(over="ignore"):
                    vals[j][both] = {"sum": vals[j][both] + x[both], "min": np.minimum(vals[j][both], x[both]), "max": np.maximum(vals[j][both], x[both])}[how]
                vals[j][first] = x[first]
                nulls[j] &= ~present
        self.flags |= failed << 8
        self._folded = (occ, vals, nulls)

    def dense_flags(self):
        return self.flags, self.error & 0xFF

    def dense_grow(self):
        pass

    def local_result(self):
        """The occupied slots of the owned range as a host View of the job's result schema (keys rebuilt from the slot numbers)."""
        occ, vals, nulls = self._folded
        rank = dist.get_rank()
        npart, cap = int(self.layout["n_parts"]), int(self.layout["part_cap"])
        ppc = npart // self.n_chunks
        slots = np.nonzero(occ[:-1])[0]
        idx = (slots % cap) * npart + (rank * ppc + slots // cap)
        cols, stride = [], 1
        strides = []
        for k in range(len(KEYS) - 1, -1, -1):
            strides.insert(0, stride)
            stride *= self.spans[k]
        for k, name in enumerate(KEYS):
            off = (idx // strides[k]) % self.spans[k]
            lo, hi = self.ranges[k]
            isnull = (off == self.spans[k] - 1) if KEY_NULLABLE[k] else np.zeros(len(off), bool)
            o = np.uint64(lo) + np.where(isnull, 0, off).astype(np.uint64)
            if KEY_SIGNED[k]:
                cols.append(ss.Column(np.where(isnull, 0, (o ^ np.uint64(SIGN)).astype(np.int64)).astype(np.int32), isnull if KEY_NULLABLE[k] else None))
            else:
                cols.append(ss.Column(o.astype(bool), None))
        for j in range(self.n_aggs):
            nullable = self._schema[len(KEYS) + j][2]
            cols.append(ss.Column(vals[j][slots], nulls[j][slots] if nullable else None))
        ts = ss.TupleSchema([ss.Attribute(n, t, ss.NULLABLE if nl else ss.NOT_NULLABLE) for (n, t, nl) in self._schema])
        return ss.View(ts, cols)


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def shard_views(scenario, rank):
    """Per step, this rank's shard.  Rank 0 and rank 1 see different k1 ranges (the union is what the table must cover)."""
    lo, hi = ((0, 20), (15, 40))[rank]
    n = (9000, 4000)[rank]
    views = [make_view(n, seed=11 + rank, k1_lo=lo, k1_hi=hi)]
    if scenario == "wider_later":            # step 2: rank 1's keys leave the agreed ranges -> flag 4 on BOTH ranks -> wider ranges, repeat
        views.append(make_view(n, seed=21 + rank, k1_lo=lo, k1_hi=hi + (25 if rank == 1 else 0)))
    return views


def worker(rank, world, port, scenario, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fail = 2 if (scenario == "rank1_fails" and rank == 1) else None
        code = ss.INTERRUPTED if scenario == "rank1_fails" else ss.ERROR_MEMORY_EXCEEDED
        backend = HostDenseTable(job_op, fail_at_step=fail, fail_code=code)
        job = DenseShardedGroupAggregate(backend)
        views = shard_views(scenario, rank)
        if scenario == "rank1_fails":
            views = views * 2
        if scenario == "too_wide":
            views = [make_view(5000, seed=31 + rank, k1_lo=-(1 << 30) * rank, k1_hi=(1 << 30))]
        outcome = []
        for view in views:
            try:
                job.step(view)
                repeats = 0
                while not job.check():
                    repeats += 1
                    job.step(view)
                got = job.gather_result()
                outcome.append(("ok", repeats, job.collectives, job.setup_collectives,
                                [(got.column(i).data, got.column(i).is_null) for i in range(got.column_count())]))
            except ss.SupersonicException as e:
                outcome.append(("raised", e.return_code))
                break
        q.put((rank, outcome))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def run_two_ranks(scenario):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, scenario, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(2))          # (a rank that hung in the collective would time out here)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return results


def whole(step):
    parts = [shard_views("wider_later", r)[step] for r in range(2)]
    full = ss.View(parts[0].schema(), [ss.Column(np.concatenate([p.column(i).data for p in parts]),
                                                  None if parts[0].column(i).is_null is None else np.concatenate([p.column(i).is_null for p in parts]))
                                       for i in range(parts[0].column_count())])
    return oracle.run(job_op(full))[1]


def test_dense_exchange_two_ranks_agree_on_ranges_and_own_disjoint_slot_slices():
    results = run_two_ranks("plain")
    want = whole(0)
    for rank in (0, 1):
        (status, repeats, collectives, setup, cols), = results[rank]
        assert status == "ok" and repeats == 0 and collectives == 1 and setup == 1, results[rank][0][:4]
        assert_cols_equal(sort_rows(cols), sort_rows(want), context="rank %d: the gathered owners' slices" % rank)


def test_a_key_outside_the_ranges_on_one_rank_makes_every_rank_widen_and_repeat():
    results = run_two_ranks("wider_later")
    for step in (0, 1):
        want = whole(step)
        for rank in (0, 1):
            status, repeats, _c, setup, cols = results[rank][step]
            assert status == "ok" and repeats == (0, 1)[step] and setup == (1, 2)[step], (rank, step, results[rank][step][:4])
            assert_cols_equal(sort_rows(cols), sort_rows(want), context="rank %d, step %d" % (rank, step))


def test_a_failed_shard_run_on_one_rank_fails_the_step_on_every_rank_without_a_hang():
    # rank 1's second run raises INTERRUPTED before the collective: it still sends its (flagged) chunks, rank 0 does not wait forever,
    # and both ranks raise the SAME code from check()
    results = run_two_ranks("rank1_fails")
    for rank in (0, 1):
        assert results[rank][0][0] == "ok"
        assert results[rank][1] == ("raised", ss.INTERRUPTED), results[rank][1]


def test_ranges_too_wide_for_a_table_are_refused_identically_on_every_rank():
    results = run_two_ranks("too_wide")                 # the union spans 2^31 values: ssgpu_plan_set_dense refuses (the caller takes the image exchange)
    for rank in (0, 1):
        assert results[rank] == [("raised", ss.ERROR_INVALID_ARGUMENT_VALUE)], results[rank]
