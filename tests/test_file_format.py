"""The reference's View file format (cursor/infrastructure/file_io.cc:176-193,377-440): chunks of
<= 8192 rows, each `uint64 row_count` followed, per column, by `row_count` bool bytes of is_null
(NULLABLE attributes only) and `row_count * sizeof(type)` bytes of raw data.

CPU: the Python writer / host reader against a byte image built by hand from that description.
GPU: file -> device Block through the pinned staging path (ssgpu_block_create_from_file) -> query,
and result -> file (ssgpu_result_write_file) -> host reader, both against the oracle."""
import struct

import numpy as np
import pytest

import supersonic_amd as ss
from oracle import oracle
from helpers import to_cols, assert_cols_equal

NA = ss.NamedAttribute


def schema3():
    return ss.TupleSchema([ss.Attribute("k", ss.INT32, ss.NULLABLE), ss.Attribute("v", ss.DOUBLE), ss.Attribute("b", ss.BOOL, ss.NULLABLE)])


def test_writer_matches_hand_built_image(tmp_path):
    view = ss.View(schema3(), [ss.Column(np.array([7, -1, 5], np.int32), np.array([False, True, False])),
                               np.array([1.5, -2.0, 0.25]), ss.Column(np.array([True, False, True]), np.array([False, False, True]))])
    path = str(tmp_path / "v.ssv")
    out = ss.FileOutput(path)
    assert out.Write(view) == 3
    out.Finalize()
    image = (struct.pack("<Q", 3)
             + bytes([0, 1, 0]) + struct.pack("<3i", 7, -1, 5)           # k: is_null bytes, then data
             + struct.pack("<3d", 1.5, -2.0, 0.25)                       # v: NOT NULL -> data only
             + bytes([0, 0, 1]) + bytes([1, 0, 1]))                      # b: is_null bytes, then bool bytes
    assert open(path, "rb").read() == image
    back = ss.read_view_file(schema3(), path)
    assert_cols_equal(to_cols(back), to_cols(view), context="file round trip")


def test_chunking_and_empty(tmp_path):
    n = 20000     # 8192 + 8192 + 3616
    rng = np.random.default_rng(5)
    view = ss.View(schema3(), [ss.Column(rng.integers(-50, 50, n).astype(np.int32), rng.random(n) < 0.2), rng.standard_normal(n),
                               ss.Column(rng.integers(0, 2, n).astype(bool), rng.random(n) < 0.2)])
    path = str(tmp_path / "big.ssv")
    out = ss.FileOutput(path); out.Write(view); out.Finalize()
    raw = open(path, "rb").read()
    row_bytes = (1 + 4) + 8 + (1 + 1)
    assert len(raw) == 3 * 8 + n * row_bytes
    assert struct.unpack_from("<Q", raw, 0)[0] == 8192
    assert struct.unpack_from("<Q", raw, 8 + 8192 * row_bytes)[0] == 8192
    assert struct.unpack_from("<Q", raw, 16 + 16384 * row_bytes)[0] == n - 16384
    assert_cols_equal(to_cols(ss.read_view_file(schema3(), path)), to_cols(view), context="chunked round trip")
    empty = str(tmp_path / "empty.ssv")
    out = ss.FileOutput(empty); out.Write(ss.View(schema3(), [ss.Column(np.zeros(0, np.int32), np.zeros(0, bool)), np.zeros(0), ss.Column(np.zeros(0, bool), np.zeros(0, bool))])); out.Finalize()
    assert open(empty, "rb").read() == b""
    assert ss.read_view_file(schema3(), empty).row_count() == 0
    with open(str(tmp_path / "trunc.ssv"), "wb") as f:
        f.write(raw[:-5])
    with pytest.raises(ss.SupersonicException):
        ss.read_view_file(schema3(), str(tmp_path / "trunc.ssv"))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 5, 8192, 30011])
def test_file_to_device_block_to_query_to_file(gpu_ctx, tmp_path, n):
    rng = np.random.default_rng(9)
    view = ss.View(schema3(), [ss.Column(rng.integers(0, 13, n).astype(np.int32), rng.random(n) < 0.1), rng.integers(-4000, 4000, n) * 0.25,
                               ss.Column(rng.integers(0, 2, n).astype(bool), rng.random(n) < 0.1)])
    src = str(tmp_path / "in.ssv")
    out = ss.FileOutput(src); out.Write(view); out.Finalize()
    dev = ss.FileInput(schema3(), src, gpu_ctx)            # pinned staging -> HBM
    assert dev.row_count() == n
    spec = ss.AggregationSpecification().AddAggregation(ss.SUM, "v", "s").AddAggregation(ss.COUNT, "b", "c")

    def query(v):
        return ss.GroupAggregate(ss.ProjectNamedAttributes(["k"]), spec, None,
                                 ss.Filter(ss.IfNull(NA("b"), ss.ConstBool(True)), ss.ProjectAllAttributes(), ss.ScanView(v)))
    plan = ss.Plan(query(dev), gpu_ctx)
    plan.run(dev)
    dst = str(tmp_path / "out.ssv")
    plan.write_file(dst)                                   # result -> file in the same format
    got = ss.read_view_file(plan.result_schema, dst)
    _schema, want = oracle.run(query(view))
    from helpers import sort_rows
    assert_cols_equal(sort_rows(to_cols(got)), sort_rows(want), context="file -> block -> group -> file")
    # and the block itself writes back byte-identically
    copy = str(tmp_path / "copy.ssv")
    dev.write_file(copy)
    assert open(copy, "rb").read() == open(src, "rb").read()


def schema_s():
    return ss.TupleSchema([ss.Attribute("s", ss.STRING, ss.NULLABLE), ss.Attribute("k", ss.INT32), ss.Attribute("t", ss.STRING)])


def test_variable_length_columns_match_hand_built_image(tmp_path):
    # file_io.cc:22-27,122-147: [is_null bytes][uint64 length per row, 0 for NULL and empty][bytes of the rest in one run]
    view = ss.View(schema_s(), [ss.Column(np.array([b"ab", b"ignored", b"", b"xyz"], dtype=object), np.array([False, True, False, False])),
                                np.array([1, 2, 3, 4], np.int32), np.array([b"", b"q", b"\x00\x01", b"end"], dtype=object)])
    path = str(tmp_path / "s.ssv")
    out = ss.FileOutput(path); out.Write(view); out.Finalize()
    image = (struct.pack("<Q", 4)
             + bytes([0, 1, 0, 0]) + struct.pack("<4Q", 2, 0, 0, 3) + b"abxyz"
             + struct.pack("<4i", 1, 2, 3, 4)
             + struct.pack("<4Q", 0, 1, 2, 3) + b"q\x00\x01end")
    assert open(path, "rb").read() == image
    back = ss.read_view_file(schema_s(), path)
    assert list(back.column(0).is_null) == [False, True, False, False]
    assert list(back.column(0).data) == [b"ab", b"", b"", b"xyz"]            # a NULL row reads back as an empty piece
    assert list(back.column(2).data) == [b"", b"q", b"\x00\x01", b"end"]
    with open(str(tmp_path / "trunc.ssv"), "wb") as f:
        f.write(image[:-2])
    with pytest.raises(ss.SupersonicException):
        ss.read_view_file(schema_s(), str(tmp_path / "trunc.ssv"))


def test_variable_length_chunking(tmp_path):
    n = 8192 + 100
    rng = np.random.default_rng(2)
    words = np.array([bytes(rng.integers(97, 123, rng.integers(0, 9)).astype(np.uint8)) for _ in range(n)], dtype=object)
    view = ss.View(schema_s(), [ss.Column(words, rng.random(n) < 0.2), np.arange(n, dtype=np.int32), words[::-1].copy()])
    path = str(tmp_path / "c.ssv")
    out = ss.FileOutput(path); out.Write(view); out.Finalize()
    back = ss.read_view_file(schema_s(), path)
    nulls = view.column(0).is_null
    assert list(back.column(0).is_null) == list(nulls)
    assert [b"" if z else w for w, z in zip(words, nulls)] == list(back.column(0).data)
    assert list(back.column(2).data) == list(words[::-1])
    assert list(back.column(1).data) == list(range(n))


@pytest.mark.gpu
def test_string_file_to_query_to_file(gpu_ctx, tmp_path):
    n = 20000
    rng = np.random.default_rng(4)
    pool = np.array([b"", b"a", b"ab", b"b", b"zz", b"a\x00", b"m" * 40], dtype=object)
    view = ss.View(schema_s(), [ss.Column(pool[rng.integers(0, len(pool), n)], rng.random(n) < 0.1), rng.integers(-9, 9, n).astype(np.int32),
                                pool[rng.integers(0, len(pool), n)]])
    src = str(tmp_path / "in.ssv")
    out = ss.FileOutput(src); out.Write(view); out.Finalize()
    host = ss.FileInput(schema_s(), src, gpu_ctx)
    spec = ss.AggregationSpecification().AddAggregation(ss.SUM, "k", "sk").AddAggregation(ss.MAX, "s", "ms").AddAggregation(ss.COUNT, "s", "c")

    def query(v):
        return ss.GroupAggregate(ss.ProjectNamedAttributes(["t"]), spec, None,
                                 ss.Filter(ss.NotEqual(NA("t"), ss.ConstString("b")), ss.ProjectAllAttributes(), ss.ScanView(v)))
    plan = ss.Plan(query(host), gpu_ctx)
    plan.run()
    dst = str(tmp_path / "out.ssv")
    plan.write_file(dst)
    got = ss.read_view_file(plan.result_schema, dst)
    oschema, want = oracle.run(query(view), 1 << 20)
    from helpers import sort_rows
    assert_cols_equal(sort_rows(to_cols(got)), sort_rows(want), context="STRING file round trip")


def test_corrupt_chunk_headers_are_io_errors(tmp_path):
    # FileInputCursor::Next, file_io.cc:398-409: a chunk of 0 rows or of more than kMaxChunkRowCount rows is
    # ERROR_GENERAL_IO_ERROR (101) with the reference's message -- never a giant read or a silent success
    schema = ss.TupleSchema([ss.Attribute("a", ss.INT64)])
    for rows, text in ((0, "Chunk of size 0."), (8193, "Input chunk too large."), (1 << 60, "Input chunk too large.")):
        path = str(tmp_path / ("bad_%d.view" % (rows % 100000)))
        with open(path, "wb") as f:
            f.write(np.array([rows], np.uint64).tobytes())
            f.write(np.arange(16, dtype=np.int64).tobytes())
        with pytest.raises(ss.SupersonicException) as e:
            ss.read_view_file(schema, path)
        assert e.value.return_code == ss.ERROR_GENERAL_IO_ERROR and text in str(e.value)
    short = str(tmp_path / "short.view")
    with open(short, "wb") as f:
        f.write(np.array([4], np.uint64).tobytes())
        f.write(np.arange(2, dtype=np.int64).tobytes())
    with pytest.raises(ss.SupersonicException) as e:
        ss.read_view_file(schema, short)
    assert e.value.return_code == ss.ERROR_GENERAL_IO_ERROR
