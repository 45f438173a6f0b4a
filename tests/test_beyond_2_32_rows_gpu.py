"""More than 2^32 rows in one run.  288 GB of HBM hold column blocks whose row indices do not fit 32 bits (an INT32 column of 2^32 + 1537
rows is 17 GB): tile numbers, row offsets, byte offsets and the row ids FIRST / LAST order by have to be 64-bit all the way.  Checked
against torch reductions over the same device memory; torch is plumbing here, every query runs through the C ABI."""
import pytest

import supersonic_amd as ss

pytestmark = pytest.mark.gpu
NA = ss.NamedAttribute
ROWS = (1 << 32) + 1537


@pytest.fixture(scope="module")
def column():
    import torch
    free, _total = torch.cuda.mem_get_info(0)
    if free < (48 << 30):
        pytest.skip("needs 48 GB of free device memory")
    device = torch.device("cuda", 0)
    x = torch.empty(ROWS, dtype=torch.int32, device=device)
    piece = 1 << 28
    for lo in range(0, ROWS, piece):          # x[i] = (i * 7 + i // 1000003) mod 1009 - 4, without a 34 GB index tensor
        n = min(piece, ROWS - lo)
        i = torch.arange(lo, lo + n, dtype=torch.int64, device=device)
        x[lo:lo + n] = ((i * 7 + i // 1000003) % 1009 - 4).to(torch.int32)
        del i
    x[ROWS - 1] = 777                          # what LAST must find, at a row id beyond 2^32
    x[0] = -3
    torch.cuda.synchronize()
    return torch, device, x


def test_scalar_aggregate_over_more_than_2_32_rows(column):
    torch, device, x = column
    ctx = ss.Context(0)
    schema = ss.TupleSchema([ss.Attribute("x", ss.INT32)])
    view = ss.DeviceView(schema, [(x.data_ptr(), 0)], ROWS)
    spec = (ss.AggregationSpecification().AddAggregation(ss.COUNT, "", "n").AddAggregationWithDefinedOutputType(ss.SUM, "x", "s", ss.INT64)
            .AddAggregation(ss.MIN, "x", "mn").AddAggregation(ss.MAX, "x", "mx").AddAggregation(ss.FIRST, "x", "f").AddAggregation(ss.LAST, "x", "l"))
    op = ss.ScalarAggregate(spec, ss.Filter(ss.NotEqual(NA("x"), ss.ConstInt32(5)), ss.ProjectAllAttributes(), ss.ScanView(view)))
    plan = ss.Plan(op, ctx)
    plan.run(view)
    got = plan.fetch()
    row = [got.column(i).data[0].item() for i in range(got.column_count())]
    n = s = 0
    piece = 1 << 29
    for lo in range(0, ROWS, piece):
        part = x[lo:lo + piece]
        m = part != 5
        n += int(m.sum().item())
        s += int(part[m].to(torch.int64).sum().item())
    assert row == [n, s, -4, 1004, -3, 777], row
    assert n > (1 << 32) - (1 << 24)           # (the count itself does not fit 32 bits)


def test_materialising_filter_keeps_rows_beyond_2_32(column):
    # survivors: every row whose value is 1003 or more (about 0.2 %), and the last row: the compacting store's offsets are 64-bit
    torch, device, x = column
    ctx = ss.Context(0)
    schema = ss.TupleSchema([ss.Attribute("x", ss.INT32)])
    view = ss.DeviceView(schema, [(x.data_ptr(), 0)], ROWS)
    e = ss.CompoundExpression().Add(NA("x")).AddAs("y", ss.Plus(NA("x"), ss.ConstInt32(1)))
    op = ss.Compute(e, ss.Filter(ss.Or(ss.GreaterOrEqual(NA("x"), ss.ConstInt32(1003)), ss.Equal(NA("x"), ss.ConstInt32(777))), ss.ProjectAllAttributes(), ss.ScanView(view)))
    plan = ss.Plan(op, ctx)
    plan.run(view)
    dv = plan.result_device_view()
    want_n = 0
    piece = 1 << 29
    tail = None
    for lo in range(0, ROWS, piece):
        part = x[lo:lo + piece]
        m = (part >= 1003) | (part == 777)
        want_n += int(m.sum().item())
        tail = part[m][-4:].clone() if int(m.sum().item()) >= 4 else tail
    assert dv.row_count() == want_n

    class _DevPtr(object):
        def __init__(self, ptr, count, typestr):
            self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2}
    out_x = torch.as_tensor(_DevPtr(dv._ptrs[0][0], want_n, "<i4"), device=device)
    out_y = torch.as_tensor(_DevPtr(dv._ptrs[1][0], want_n, "<i4"), device=device)
    assert torch.equal(out_x[-4:], tail) and int(out_x[-1].item()) == 777      # input order kept up to the very last row
    assert bool(((out_x >= 1003) | (out_x == 777)).all().item()) and torch.equal(out_y, out_x + 1)


@pytest.mark.parametrize("dense", [0, 1])
def test_group_aggregate_over_more_than_2_32_rows(column, dense):
    # 1009 groups of ~4.3 M rows each; the hashed shapes (dense = 0) and the dense slots both count rows beyond 32 bits per run
    torch, device, x = column
    ctx = ss.Context(0)
    ctx.set_option("group_dense", dense)
    schema = ss.TupleSchema([ss.Attribute("x", ss.INT32)])
    view = ss.DeviceView(schema, [(x.data_ptr(), 0)], ROWS)
    spec = (ss.AggregationSpecification().AddAggregation(ss.COUNT, "", "n").AddAggregationWithDefinedOutputType(ss.SUM, "x", "s", ss.INT64).AddAggregation(ss.MAX, "x", "mx"))
    plan = ss.Plan(ss.GroupAggregate(ss.ProjectNamedAttribute("x"), spec, None, ss.ScanView(view)), ctx)
    plan.run(view)
    got = plan.fetch()
    keys, counts, sums, maxes = (got.column(i).data for i in range(4))
    want = torch.zeros(1009 + 4 + 1, dtype=torch.int64, device=device)
    piece = 1 << 29
    for lo in range(0, ROWS, piece):
        want += torch.bincount((x[lo:lo + piece] + 4).to(torch.int64), minlength=1009 + 4 + 1)
    want = want.cpu().numpy()
    assert int(counts.sum()) == ROWS and len(keys) == int((want > 0).sum())
    for k, n, s, m in zip(keys.tolist(), counts.tolist(), sums.tolist(), maxes.tolist()):
        assert n == want[k + 4] and s == k * n and m == k, (k, n, s, m)


@pytest.mark.parametrize("dense", [0, 1])
def test_partitioned_group_aggregate_over_more_than_2_32_rows(column, dense):
    # ~98 k groups, more rows than the two-pass shapes' 32-bit record indices reach
    torch, device, x = column
    free, _total = torch.cuda.mem_get_info(0)
    if free < (64 << 30):
        pytest.skip("needs 64 GB of free device memory")
    k2 = torch.empty(ROWS, dtype=torch.int32, device=device)
    piece = 1 << 28
    for lo in range(0, ROWS, piece):
        n = min(piece, ROWS - lo)
        k2[lo:lo + n] = (torch.arange(lo, lo + n, dtype=torch.int64, device=device) % 97).to(torch.int32)
    ctx = ss.Context(0)
    ctx.set_option("group_dense", dense)
    schema = ss.TupleSchema([ss.Attribute("x", ss.INT32), ss.Attribute("k2", ss.INT32)])
    view = ss.DeviceView(schema, [(x.data_ptr(), 0), (k2.data_ptr(), 0)], ROWS)
    spec = ss.AggregationSpecification().AddAggregation(ss.COUNT, "", "n").AddAggregationWithDefinedOutputType(ss.SUM, "x", "s", ss.INT64)
    plan = ss.Plan(ss.GroupAggregate(ss.ProjectNamedAttributes(["x", "k2"]), spec, None, ss.ScanView(view)), ctx)
    plan.run(view)
    got = plan.fetch()
    kx, kk, counts, sums = (got.column(i).data for i in range(4))
    # (the two-pass shapes index their partition records with 32 bits: beyond 2^32 records the stage takes the direct shape -- LDS
    #  pre-aggregation + the global table -- whichever form was asked for; what matters here is that nothing wraps on the way there)
    assert plan.stage_info()[0]["group_shape"] == 0
    assert int(counts.sum()) == ROWS and bool((sums == kx.astype("int64") * counts.astype("int64")).all())
    want = torch.zeros(1014 * 97, dtype=torch.int64, device=device)
    piece = 1 << 28
    for lo in range(0, ROWS, piece):
        want += torch.bincount((x[lo:lo + piece] + 4).to(torch.int64) * 97 + k2[lo:lo + piece].to(torch.int64), minlength=1014 * 97)
    want = want.cpu().numpy()
    assert len(kx) == int((want > 0).sum())
    idx = (kx.astype("int64") + 4) * 97 + kk.astype("int64")
    assert (want[idx] == counts.astype("int64")).all()
