"""The host-side seams of the C ABI that need no GPU: the BufferAllocator-shaped allocator with its MemoryLimit quota
(reference: supersonic/base/memory/memory.h:100-233,465-520 and base/memory/memory_test.cc's quota cases), the
order-preserving STRING dictionary (StringPiece order, types_infrastructure.h:238-246) and Expression::Bind
(expression/base/expression.h:158-160) on a bind-only context."""
import ctypes as C

import numpy as np
import pytest

import supersonic_amd as ss
from supersonic_amd import _lib as L


@pytest.fixture(scope="module")
def ctx():
    return ss.Context(-1)          # bind-only: no device


def test_allocator_unlimited(ctx):
    a = ss.HeapBufferAllocator(ctx)
    assert a.Available() > (1 << 62)
    p, g = a.Allocate(1000)
    assert p and g == 1000 and p % 256 == 0
    C.memset(p, 0xAB, 1000)
    assert a.GetUsage() == 1000
    a.Free(p)
    assert a.GetUsage() == 0


def test_allocator_zero_size_is_not_null(ctx):
    # memory.h:112-117: a zero-size request succeeds and data() is not NULL
    a = ss.MemoryLimit(0, ctx)
    p, g = a.Allocate(0)
    assert p and g == 0
    assert a.Allocate(1) is None
    a.Free(p)


def test_allocator_quota_and_best_effort(ctx):
    a = ss.MemoryLimit(1000, ctx)
    p1, g1 = a.Allocate(600)
    assert g1 == 600 and a.Available() == 400
    assert a.Allocate(500) is None                 # over the quota: a NULL Buffer
    assert a.GetUsage() == 600                      # a refused request charges nothing
    p2, g2 = a.BestEffortAllocate(500, 100)         # as much as the quota leaves, at least `minimal`
    assert g2 == 400 and a.Available() == 0
    assert a.BestEffortAllocate(10, 1) is None
    a.Free(p1)
    assert a.Available() == 600
    a.Free(p2)
    assert a.GetUsage() == 0


def test_allocator_reallocate_keeps_contents_and_old_buffer_on_failure(ctx):
    a = ss.MemoryLimit(4096, ctx)
    p, g = a.Allocate(1024)
    src = (C.c_ubyte * 1024).from_address(p)
    for i in range(1024):
        src[i] = i & 255
    q, g2 = a.Reallocate(p, 3000)                    # 3000 fit next to the old 1024
    assert g2 == 3000 and a.GetUsage() == 3000
    assert bytes((C.c_ubyte * 1024).from_address(q)) == bytes(i & 255 for i in range(1024))
    assert a.Reallocate(q, 5000) is None             # does not fit ...
    assert a.GetUsage() == 3000                       # ... and the old buffer is still owned
    assert bytes((C.c_ubyte * 16).from_address(q)) == bytes(range(16))
    r, g3 = a.Reallocate(q, 5000, 100)               # best effort: what the quota leaves NEXT TO the old buffer
    assert g3 == 1096 and a.GetUsage() == 1096
    a.Free(r)


def test_allocator_follows_the_reference_memory_limit_tests(ctx):
    # base/memory/memory_test.cc:176-181 AvailableInMemoryLimit
    limit = ss.MemoryLimit(1000, ctx)
    assert limit.Available() == 1000
    p, _ = limit.Allocate(300)
    assert limit.Available() == 700
    # :183-198 ReallocShouldAdjustQuota: the mediator reserves the whole requested size while the old buffer is held
    p, g = limit.Reallocate(p, 500)
    assert g == 500 and limit.GetUsage() == 500 and limit.Available() == 500
    assert limit.Reallocate(p, 700, 600) is None
    assert limit.GetUsage() == 500 and limit.Available() == 500
    limit.Free(p)
    # :200-210 AllocatingEmptyAlwaysSucceeds
    zero = ss.MemoryLimit(0, ctx)
    assert zero.BestEffortAllocate(300, 10) is None
    p2, g2 = zero.BestEffortAllocate(300, 0)
    assert p2 and g2 == 0
    p3, g3 = zero.Allocate(0)
    assert p3 and g3 == 0
    # :212-223 ReallocatingEmptyAlwaysSucceeds
    lim = ss.MemoryLimit(300, ctx)
    b, gb = lim.BestEffortAllocate(400, 300)
    assert b and gb == 300
    b, gb = lim.Reallocate(b, 100, 0)                # nothing is left next to the 300 bytes held: the buffer shrinks to 0
    assert b and gb == 0
    b, gb = lim.Reallocate(b, 0)
    assert b and gb == 0
    # :53-99 SharedQuotaAllocatorShouldAllocate (one allocator, the same arithmetic)
    q = ss.MemoryLimit(1000, ctx)
    b1, s1 = q.BestEffortAllocate(450, 100)
    assert s1 == 450 and q.GetUsage() == 450 and q.Available() == 550
    b2, s2 = q.BestEffortAllocate(350, 100)
    assert s2 == 350 and q.GetUsage() == 800 and q.Available() == 200
    b3, s3 = q.BestEffortAllocate(500, 100)
    assert s3 == 200 and q.GetUsage() == 1000 and q.Available() == 0
    assert q.BestEffortAllocate(500, 100) is None and q.Available() == 0
    q.Free(b3); q.Free(b2)
    assert q.GetUsage() == 450 and q.Available() == 550
    q.Free(b1)
    assert q.GetUsage() == 0 and q.Available() == 1000
    b5, s5 = q.BestEffortAllocate(1200, 800)
    assert s5 == 1000 and q.GetUsage() == 1000 and q.Available() == 0


def test_allocator_rejects_bad_arguments(ctx):
    a = ss.MemoryLimit(100, ctx)
    with pytest.raises(ss.SupersonicException):
        a.BestEffortAllocate(10, 20)                  # minimal > requested


def test_dictionary_is_order_preserving():
    words = [b"b", b"", b"ab", b"a", b"abc", b"a\x00", b"\xff", b"a", b"b", "zażółć".encode(), b"B"]
    d = ss.StringDictionary(words)
    vals = d.values
    assert vals == sorted(set(words))                # bytes order == memcmp, then length (StringPiece operator<)
    assert len(d) == len(set(words))
    codes = d.encode(words, None)
    for i, w in enumerate(words):
        assert vals[codes[i]] == w
    for x in words:
        for y in words:
            assert (d.code_of(x) < d.code_of(y)) == (x < y)
    back = d.decode(codes, None)
    assert list(back) == words


def test_dictionary_nulls_and_unknown_values():
    d = ss.StringDictionary(["x", "y"])
    codes = d.encode(["x", "nope", "y"], np.array([False, True, False]))
    assert list(codes) == [0, 0, 1]                   # a NULL row is not looked up
    with pytest.raises(ss.SupersonicException) as e:
        d.encode(["x", "nope"], None)
    assert e.value.return_code == L.ERROR_INVALID_ARGUMENT_VALUE
    with pytest.raises(ss.SupersonicException):
        d.value(2)
    assert list(d.decode(np.array([1, 0]), np.array([True, False]))) == [b"", b"x"]


def test_dictionary_owns_its_bytes():
    lib = L.load()
    src = [bytearray(b"hello"), bytearray(b"world")]
    bufs = [(C.c_char * len(b)).from_buffer(b) for b in src]
    arr = (C.c_char_p * 2)(*[C.cast(b, C.c_char_p) for b in bufs])
    lens = (C.c_int32 * 2)(5, 5)
    h = C.c_void_p()
    assert lib.ssgpu_dict_create(arr, lens, 2, C.byref(h)) == L.OK
    src[0][:] = b"XXXXX"                              # the caller's memory changes; the dictionary copied it (Arena rule)
    ptr, n = C.c_void_p(), C.c_int32()
    assert lib.ssgpu_dict_decode(h, 0, C.byref(ptr), C.byref(n)) == L.OK
    assert C.string_at(ptr, n.value) == b"hello"
    lib.ssgpu_dict_destroy(h)


def test_expression_bind_result_schema_and_errors(ctx):
    schema = ss.TupleSchema([ss.Attribute("a", ss.INT32, ss.NOT_NULLABLE), ss.Attribute("b", ss.DOUBLE, ss.NULLABLE)])
    bound = ss.Plus(ss.NamedAttribute("a"), ss.NamedAttribute("b")).Bind(schema, ss.HeapBufferAllocator(ctx), 1024, ctx)
    rs = bound.result_schema
    assert rs.attribute_count() == 1
    assert rs.attribute(0).type() == ss.DOUBLE and rs.attribute(0).is_nullable()
    assert bound.row_capacity() == 1024
    with pytest.raises(ss.SupersonicException) as e:
        ss.NamedAttribute("missing").Bind(schema, None, 16, ctx)
    assert e.value.return_code == L.ERROR_ATTRIBUTE_MISSING
    with pytest.raises(ss.SupersonicException) as e:
        ss.And(ss.NamedAttribute("a"), ss.NamedAttribute("b")).Bind(schema, None, 16, ctx)
    assert 400 <= e.value.return_code < 500
    # a compound expression binds to one attribute per element (expression.h:268)
    comp = ss.CompoundExpression().AddAs("x", ss.NamedAttribute("a")).AddAs("y", ss.Negate(ss.NamedAttribute("b")))
    b2 = comp.Bind(schema, None, 0, ctx)
    assert [b2.result_schema.attribute(i).name() for i in range(2)] == ["x", "y"]
    assert b2.row_capacity() > (1 << 62)


def test_evaluate_on_a_bind_only_context_fails_loudly(ctx):
    schema = ss.TupleSchema([ss.Attribute("a", ss.INT32, ss.NOT_NULLABLE)])
    bound = ss.Negate(ss.NamedAttribute("a")).Bind(schema, None, 4, ctx)
    view = ss.View(schema, [ss.Column(np.arange(8, dtype=np.int32))])
    r = bound.Evaluate(view)
    assert r.is_failure()                              # no device, and no CPU fallback behind it


def test_memory_limit_api(ctx):
    schema = ss.TupleSchema([ss.Attribute("a", ss.INT32, ss.NOT_NULLABLE)])
    view = ss.View(schema, [ss.Column(np.arange(8, dtype=np.int32))])
    op = ss.Compute(ss.Negate(ss.NamedAttribute("a")), ss.ScanView(view))
    op.SetBufferAllocator(ss.MemoryLimit(1 << 20, ctx), True)
    plan = ss.Plan(op, ctx)
    assert plan.memory_in_use() == 0
    plan.set_memory_limit(None)
    plan.set_memory_limit(4096)
