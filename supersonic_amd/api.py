"""Host-side mirror of Supersonic's Expression / Operation / Cursor builder API
(supersonic/supersonic.h) over the C ABI of libssgpu.so.

Names, argument meaning and error behaviour follow the reference so that the
parity tests read like the reference's own tests:

    schema = TupleSchema([Attribute("a", INT64, NOT_NULLABLE), ...])
    op = ScalarAggregate(AggregationSpecification().AddAggregation(SUM, "s", "sum_s"),
             Filter(Greater(NamedAttribute("a"), ConstInt64(499)), ProjectAllAttributes(),
                 Compute(CompoundExpression().Add(NamedAttribute("a"))
                                             .AddAs("s", Plus(NamedAttribute("a"), NamedAttribute("b"))),
                         ScanView(view))))
    cursor = op.CreateCursor()          # binds: raises SupersonicException(return_code, message)
    result = cursor.Next(1024)          # ResultView: has_data() / is_eos() / view()

Execution is whole-shard on the GPU; Next() only slices the
finished result, as ViewIterator does (cursor/infrastructure/iterators.h:64).
There is no CPU execution path here: without libssgpu.so or without a device,
running a plan raises.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import (INT32, INT64, UINT32, UINT64, FLOAT, DOUBLE, BOOL, DATE, DATETIME, STRING, BINARY,  # noqa: F401
                   NOT_NULLABLE, NULLABLE, SUM, MIN, MAX, COUNT, CONCAT, FIRST, LAST, SUM_RESIDUAL, ASCENDING, DESCENDING,
                   OK, ERROR_UNKNOWN, ERROR_GENERAL_IO_ERROR, ERROR_MEMORY_EXCEEDED, ERROR_NOT_IMPLEMENTED, ERROR_EVALUATION_ERROR,
                   ERROR_TOO_MANY_ROWS, ERROR_ATTRIBUTE_COUNT_MISMATCH, ERROR_ATTRIBUTE_TYPE_MISMATCH,
                   ERROR_ATTRIBUTE_MISSING, ERROR_ATTRIBUTE_EXISTS, ERROR_INVALID_ARGUMENT_TYPE,
                   ERROR_INVALID_ARGUMENT_VALUE, INTERRUPTED, ERROR_NO_DEVICE, ERROR_HIP)

kDefaultRowCount = 1024  # Cursor::kDefaultRowCount, cursor/base/cursor.h:133

_NP = {INT32: np.int32, INT64: np.int64, UINT32: np.uint32, UINT64: np.uint64, FLOAT: np.float32,
       DOUBLE: np.float64, BOOL: np.bool_, DATE: np.int32, DATETIME: np.int64}
_TYPE_NAMES = {INT32: "INT32", INT64: "INT64", UINT32: "UINT32", UINT64: "UINT64", FLOAT: "FLOAT",
               DOUBLE: "DOUBLE", BOOL: "BOOL", DATE: "DATE", DATETIME: "DATETIME", STRING: "STRING",
               BINARY: "BINARY"}


def numpy_dtype(data_type):
    return np.dtype(object) if data_type == STRING else _NP[data_type]


def _as_bytes(v):
    return v.encode() if isinstance(v, str) else bytes(v)


class StringDictionary(object):
    """STRING columns cross the C ABI as INT32 codes of ONE order-preserving dictionary per plan
    (include/ssgpu.h, "STRING columns"): sorted unique byte strings of every STRING column of the
    scanned View and of every ConstString of the plan.  The dictionary itself lives behind the ABI
    (ssgpu_dict_create/encode/decode): code order is the reference's StringPiece order (memcmp, then
    length), so comparisons, MIN/MAX, group keys and sort order of the codes are those of the strings,
    and the library owns a deep copy of the bytes (the reference's Arena rule)."""

    def __init__(self, strings):
        self.lib = L.load()
        vals = [_as_bytes(v) for v in strings]
        self._arr, self._len = self._pack(vals)
        h = C.c_void_p()
        rc = self.lib.ssgpu_dict_create(self._arr, self._len, len(vals), C.byref(h))
        if rc != L.OK:
            raise SupersonicException(rc, "cannot build the STRING dictionary")
        self.handle = h
        del self._arr, self._len          # the library copied the bytes

    @staticmethod
    def _pack(vals):
        arr = (C.c_char_p * max(len(vals), 1))()
        lens = (C.c_int32 * max(len(vals), 1))()
        for i, v in enumerate(vals):
            arr[i] = v                    # c_char_p keeps the bytes object; the explicit length covers embedded NULs
            lens[i] = len(v)
        return arr, lens

    def __len__(self):
        return self.lib.ssgpu_dict_size(self.handle)

    @property
    def values(self):
        return [self.value(i) for i in range(len(self))]

    def value(self, code):
        ptr, n = C.c_void_p(), C.c_int32()
        if self.lib.ssgpu_dict_decode(self.handle, int(code), C.byref(ptr), C.byref(n)) != L.OK:
            raise SupersonicException(L.ERROR_INVALID_ARGUMENT_VALUE, "STRING code %d outside the dictionary" % int(code))
        return C.string_at(ptr, n.value)

    def code_of(self, v):
        return int(self.encode([v], None)[0])

    def encode(self, data, nulls):
        n = len(data)
        out = np.zeros(n, np.int32)
        if n == 0:
            return out
        vals = [b"" if (nulls is not None and nulls[i]) else _as_bytes(v) for i, v in enumerate(data)]
        arr, lens = self._pack(vals)
        nn = None if nulls is None else np.ascontiguousarray(nulls, dtype=np.uint8)
        rc = self.lib.ssgpu_dict_encode(self.handle, arr, lens, None if nn is None else nn.ctypes.data_as(C.c_void_p), n,
                                        out.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc != L.OK:
            raise SupersonicException(L.ERROR_INVALID_ARGUMENT_VALUE, "STRING value not in the plan's dictionary: the plan was created over another View")
        return out

    def decode(self, codes, nulls):
        out = np.empty(len(codes), dtype=object)
        cache = {}
        for i, c in enumerate(codes):
            if nulls is not None and nulls[i]:
                out[i] = b""
                continue
            c = int(c)
            v = cache.get(c)
            if v is None:
                v = cache[c] = self.value(c)
            out[i] = v
        return out

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ssgpu_dict_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class SupersonicException(Exception):
    """Exception{return_code, message} (supersonic/base/exception/exception.h:53)."""

    def __init__(self, return_code, message):
        Exception.__init__(self, "[%d] %s" % (return_code, message))
        self.return_code = return_code
        self.message = message


# --------------------------------------------------------------------------- context
class Context(object):
    _default = None

    def __init__(self, device=0):
        self.lib = L.load()
        h = C.c_void_p()
        rc = self.lib.ssgpu_ctx_create(device, C.byref(h))
        if rc != L.OK:
            raise SupersonicException(rc, "cannot create a context on device %d" % device)
        self.handle = h
        self.device = device

    @classmethod
    def default(cls):
        """Device 0 if there is a GPU, else a bind-only context (plans bind, runs fail)."""
        if cls._default is None:
            try:
                cls._default = Context(0)
            except SupersonicException:
                cls._default = Context(-1)
        return cls._default

    def set_option(self, key, value):
        rc = self.lib.ssgpu_ctx_set_option(self.handle, key.encode(), int(value))
        self.check(rc)

    def stream(self):
        return self.lib.ssgpu_ctx_stream(self.handle)

    def set_stream(self, hip_stream):
        self.check(self.lib.ssgpu_ctx_set_stream(self.handle, C.c_void_p(hip_stream)))

    def synchronize(self):
        self.check(self.lib.ssgpu_ctx_synchronize(self.handle))

    def last_error(self):
        return (self.lib.ssgpu_last_error(self.handle) or b"").decode()

    def check(self, rc):
        if rc != L.OK:
            raise SupersonicException(rc, self.last_error())

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ssgpu_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def specialized_kernels_trim(keep=0):
    """Unload specialised kernels that no plan uses, down to `keep` of them (ssgpu_specialized_kernels_trim)."""
    L.load().ssgpu_specialized_kernels_trim(int(keep))


def pool_trim(device=-1):
    """ssgpu_pool_trim: free the device blocks the library keeps between plans (all devices by default); returns bytes freed."""
    return L.load().ssgpu_pool_trim(int(device))


def memory_stats():
    """ssgpu_memory_stats: what the library holds in this process right now (device / pinned bytes, live plans, blocks and
    events, loaded specialised-kernel modules), as a dict."""
    st = L.MemoryStats()
    rc = L.load().ssgpu_memory_stats(C.byref(st))
    if rc != L.OK:
        raise SupersonicException(rc, "ssgpu_memory_stats failed")
    return {n: getattr(st, n) for n, _t in L.MemoryStats._fields_}


# ---------------------------------------------------------------------------- schema
class Attribute(object):
    def __init__(self, name, data_type, nullability=NOT_NULLABLE):
        self._name, self._type, self._nullability = name, data_type, nullability

    def name(self):
        return self._name

    def type(self):
        return self._type

    def nullability(self):
        return self._nullability

    def is_nullable(self):
        return self._nullability == NULLABLE

    def __eq__(self, o):
        return (self._name, self._type, self._nullability) == (o._name, o._type, o._nullability)

    def __repr__(self):
        return "%s: %s%s" % (self._name, _TYPE_NAMES.get(self._type, "?"), "" if self.is_nullable() else " NOT NULL")


class TupleSchema(object):
    def __init__(self, attributes=()):
        self._attrs = []
        for a in attributes:
            if not self.add_attribute(a):
                raise SupersonicException(L.ERROR_ATTRIBUTE_EXISTS, "duplicate attribute %s" % a.name())

    @staticmethod
    def Singleton(name, data_type, nullability):
        return TupleSchema([Attribute(name, data_type, nullability)])

    def add_attribute(self, attribute):
        if self.LookupAttributePosition(attribute.name()) >= 0:
            return False
        self._attrs.append(attribute)
        return True

    def attribute_count(self):
        return len(self._attrs)

    def attribute(self, i):
        return self._attrs[i]

    def LookupAttributePosition(self, name):
        for i, a in enumerate(self._attrs):
            if a.name() == name:
                return i
        return -1

    def __eq__(self, o):
        return self._attrs == o._attrs

    def __repr__(self):
        return ", ".join(repr(a) for a in self._attrs)


class Column(object):
    """One column of a host View: typed data + optional bool is_null (block.h:55-192)."""

    def __init__(self, data, is_null=None):
        self.data = data
        self.is_null = is_null


class View(object):
    """Host-side View: N (data, is_null) pairs + row_count (block.h:288-402)."""

    def __init__(self, schema, columns, row_count=None):
        self._schema = schema
        cols = []
        for i, c in enumerate(columns):
            if not isinstance(c, Column):
                c = Column(*c) if isinstance(c, tuple) else Column(c)
            nulls = None if c.is_null is None else np.ascontiguousarray(c.is_null, dtype=np.bool_)
            if schema.attribute(i).type() == STRING:
                data = np.empty(len(c.data), dtype=object)
                for j, v in enumerate(c.data):
                    data[j] = b"" if v is None else _as_bytes(v)
            else:
                data = np.ascontiguousarray(c.data, dtype=_NP[schema.attribute(i).type()])
            cols.append(Column(data, nulls))
        self._cols = cols
        self._rows = int(row_count if row_count is not None else (len(cols[0].data) if cols else 0))

    def schema(self):
        return self._schema

    def row_count(self):
        return self._rows

    def column_count(self):
        return len(self._cols)

    def column(self, i):
        return self._cols[i]


class DeviceView(object):
    """Columns already resident in HBM: (data_ptr, is_null_ptr or 0) integers."""

    def __init__(self, schema, pointers, row_count):
        self._schema, self._ptrs, self._rows = schema, list(pointers), int(row_count)

    def schema(self):
        return self._schema

    def row_count(self):
        return self._rows


class DeviceBlock(object):
    """A device-resident Block the LIBRARY lays out (ssgpu_block_create: one arena, column bases skewed against HBM channel
    conflicts -- include/ssgpu.h "device-resident Block") for columns that are produced on the device: `column_ptr(i)` is where
    the caller writes column i (row_capacity x width bytes; `null_ptr(i)` its NULL mask or 0), `view()` the DeviceView plans scan."""

    def __init__(self, schema, row_capacity, context):
        self.schema, self.ctx, self.rows = schema, context, int(row_capacity)
        attrs = [L.Attr(schema.attribute(i).name().encode(), schema.attribute(i).type(), schema.attribute(i).nullability())
                 for i in range(schema.attribute_count())]
        self._attrs = _array(L.Attr, attrs)
        h = C.c_void_p()
        context.check(context.lib.ssgpu_block_create(context.handle, self._attrs, len(attrs), self.rows, C.byref(h)))
        self.handle = h
        context.check(context.lib.ssgpu_block_set_row_count(h, self.rows))
        self._ptrs = []
        for i in range(len(attrs)):
            col = L.Column()
            context.check(context.lib.ssgpu_block_column(h, i, C.byref(col)))
            self._ptrs.append((col.data or 0, col.is_null or 0))

    def column_ptr(self, i):
        return self._ptrs[i][0]

    def null_ptr(self, i):
        return self._ptrs[i][1]

    def view(self, rows=None):
        v = DeviceView(self.schema, self._ptrs, self.rows if rows is None else int(rows))
        v._owner = self          # (the block lives as long as a view of it)
        return v

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.ctx.lib.ssgpu_block_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class BlockView(DeviceView):
    """A DeviceView over an owned device Block (kept alive with the view)."""

    def __init__(self, schema, ctx, block):
        self._ctx, self._block = ctx, block
        lib = ctx.lib
        ptrs = []
        for i in range(schema.attribute_count()):
            col = L.Column()
            ctx.check(lib.ssgpu_block_column(block, i, C.byref(col)))
            ptrs.append((col.data or 0, col.is_null or 0))
        DeviceView.__init__(self, schema, ptrs, lib.ssgpu_block_row_count(block))

    def write_file(self, path):
        self._ctx.check(self._ctx.lib.ssgpu_block_write_file(self._block, path.encode()))

    def __del__(self):
        try:
            if getattr(self, "_block", None):
                self._ctx.lib.ssgpu_block_destroy(self._block)
                self._block = None
        except Exception:
            pass


# ---- the reference's View file format (cursor/infrastructure/file_io.cc:176-193,377-440) ----------
FILE_CHUNK_ROWS = 8192   # kMaxChunkRowCount, file_io.cc:70


class FileOutput(object):
    """Sink that appends host Views to a file: per chunk (<= 8192 rows) a uint64 row count, then
    per column [row_count bool bytes of is_null if the attribute is NULLABLE][raw data]."""

    def __init__(self, path):
        self._f = open(path, "wb")

    def Write(self, view):
        schema = view.schema()
        for off in range(0, view.row_count(), FILE_CHUNK_ROWS):
            rc = min(FILE_CHUNK_ROWS, view.row_count() - off)
            self._f.write(np.uint64(rc).tobytes())
            for i in range(view.column_count()):
                col = view.column(i)
                if schema.attribute(i).is_nullable():
                    nulls = col.is_null[off:off + rc] if col.is_null is not None else np.zeros(rc, np.bool_)
                    self._f.write(np.ascontiguousarray(nulls, dtype=np.bool_).tobytes())
                else:
                    nulls = None
                if schema.attribute(i).type() in (STRING, BINARY):
                    # WriteVariableLengthData, file_io.cc:122-147: a uint64 length per row (0 for NULL and empty), then the
                    # bytes of every non-NULL, non-empty value in one run
                    vals = [b"" if (nulls is not None and nulls[j]) else _as_bytes(v) for j, v in enumerate(col.data[off:off + rc])]
                    self._f.write(np.array([len(v) for v in vals], np.uint64).tobytes())
                    self._f.write(b"".join(vals))
                else:
                    self._f.write(np.ascontiguousarray(col.data[off:off + rc]).tobytes())
        return view.row_count()

    def Finalize(self):
        self._f.close()


def read_view_file(schema, path):
    """FileInput drained on the host (numpy): the whole file as one View."""
    data = [[] for _ in range(schema.attribute_count())]
    nulls = [[] for _ in range(schema.attribute_count())]
    with open(path, "rb") as f:
        while True:
            head = f.read(8)
            if not head:
                break
            if len(head) != 8:
                raise SupersonicException(L.ERROR_GENERAL_IO_ERROR, "Reading cursor's data from the input file failed.")
            rc = int(np.frombuffer(head, np.uint64)[0])
            # FileInputCursor::Next, file_io.cc:398-409: a chunk holds 1 .. kMaxChunkRowCount rows; anything else is a corrupt header
            if rc == 0:
                raise SupersonicException(L.ERROR_GENERAL_IO_ERROR, "Reading cursor's data from the input file failed. Chunk of size 0.")
            if rc > FILE_CHUNK_ROWS:
                raise SupersonicException(L.ERROR_GENERAL_IO_ERROR, "Reading cursor's data from the input file failed. Input chunk too large.")
            for i in range(schema.attribute_count()):
                a = schema.attribute(i)
                dt = None if a.type() in (STRING, BINARY) else np.dtype(_NP[a.type()])
                if a.is_nullable():
                    raw = f.read(rc)
                    if len(raw) != rc:
                        raise SupersonicException(L.ERROR_GENERAL_IO_ERROR, "Reading cursor's data from the input file failed.")
                    nulls[i].append(np.frombuffer(raw, np.bool_))
                if a.type() in (STRING, BINARY):
                    # ReadVariableLengthData, file_io.cc:442-474: lengths, then one run of bytes cut by them
                    raw = f.read(rc * 8)
                    if len(raw) != rc * 8:
                        raise SupersonicException(L.ERROR_GENERAL_IO_ERROR, "Reading cursor's data from the input file failed.")
                    lens = np.frombuffer(raw, np.uint64).astype(np.int64)
                    total = int(lens.sum())
                    blob = f.read(total)
                    if len(blob) != total:
                        raise SupersonicException(L.ERROR_GENERAL_IO_ERROR, "Reading cursor's data from the input file failed.")
                    ends = np.cumsum(lens)
                    vals = np.empty(rc, dtype=object)
                    for j in range(rc):
                        vals[j] = blob[int(ends[j] - lens[j]):int(ends[j])]
                    data[i].append(vals)
                    continue
                raw = f.read(rc * dt.itemsize)
                if len(raw) != rc * dt.itemsize:
                    raise SupersonicException(L.ERROR_GENERAL_IO_ERROR, "Reading cursor's data from the input file failed.")
                data[i].append(np.frombuffer(raw, dt))
    cols = []
    for i in range(schema.attribute_count()):
        dt = np.dtype(object if schema.attribute(i).type() in (STRING, BINARY) else _NP[schema.attribute(i).type()])
        d = np.concatenate(data[i]) if data[i] else np.zeros(0, dt)
        z = (np.concatenate(nulls[i]) if nulls[i] else np.zeros(0, np.bool_)) if schema.attribute(i).is_nullable() else None
        cols.append(Column(d, z))
    return View(schema, cols)


def FileInput(schema, path, context=None):
    """FileInput(schema, file) drained straight into a device Block: chunks go through pinned
    staging buffers on the copy stream while the next chunk is read (ssgpu_block_create_from_file).
    Returns a device-resident view usable with ScanView.  A schema with STRING columns is read on the host instead
    (the values have to meet the plan's dictionary before they can be codes in HBM): the returned host View goes
    through ScanView's ordinary upload."""
    ctx = context or Context.default()
    if any(schema.attribute(i).type() in (STRING, BINARY) for i in range(schema.attribute_count())):
        return read_view_file(schema, path)
    attrs = (L.Attr * schema.attribute_count())()
    keep = []
    for i in range(schema.attribute_count()):
        a = schema.attribute(i)
        name = a.name().encode(); keep.append(name)
        attrs[i] = L.Attr(name, a.type(), 1 if a.is_nullable() else 0)
    block = C.c_void_p()
    ctx.check(ctx.lib.ssgpu_block_create_from_file(ctx.handle, attrs, schema.attribute_count(), path.encode(), C.byref(block)))
    return BlockView(schema, ctx, block)


# ------------------------------------------------------------------------ expressions
class Expression(object):
    def __init__(self, kind, op=0, dtype=0, args=(), i64=0, f64=0.0, name=None):
        self.kind, self.op, self.dtype, self.args, self.i64, self.f64, self.name = kind, op, dtype, list(args), i64, f64, name

    def Bind(self, input_schema, allocator=None, max_row_count=0, context=None):
        """Expression::Bind(input_schema, allocator, max_row_count) -> BoundExpressionTree
        (expression/base/expression.h:158-160); bind errors raise SupersonicException (400-499)."""
        return BoundExpressionTree(self, input_schema, allocator, max_row_count, context or Context.default())


def NamedAttribute(name):
    return Expression(L.EXPR_ATTR_NAMED, name=name)


def AttributeAt(position):
    return Expression(L.EXPR_ATTR_AT, i64=position)


def _const(dtype, i64=0, f64=0.0):
    return Expression(L.EXPR_CONST, dtype=dtype, i64=int(i64), f64=float(f64))


def ConstInt32(v): return _const(INT32, i64=v)
def ConstInt64(v): return _const(INT64, i64=v)
def ConstUint32(v): return _const(UINT32, i64=v)
def ConstUint64(v): return _const(UINT64, i64=np.array(v, dtype=np.uint64).astype(np.int64))
def ConstFloat(v): return _const(FLOAT, f64=v)


def ConstString(v):
    e = _const(STRING)
    e.sval = _as_bytes(v)
    return e
def ConstDouble(v): return _const(DOUBLE, f64=v)
def ConstBool(v): return _const(BOOL, i64=1 if v else 0)
def ConstDate(v): return _const(DATE, i64=v)
def ConstDateTime(v): return _const(DATETIME, i64=v)
def Null(data_type): return Expression(L.EXPR_NULL, dtype=data_type)


def _op(op, *args):
    return Expression(L.EXPR_OP, op=op, args=args)


# OperatorId values: supersonic/expression/proto/operators.proto
def Plus(a, b): return _op(0, a, b)
def Multiply(a, b): return _op(4, a, b)
def Minus(a, b): return _op(8, a, b)
def DivideQuiet(a, b): return _op(13, a, b)
def DivideNulling(a, b): return _op(14, a, b)
def DivideSignaling(a, b): return _op(15, a, b)
def Divide(a, b): return DivideSignaling(a, b)  # arithmetic_expressions.cc:104-107
def CppDivideNulling(a, b): return _op(18, a, b)
def CppDivideSignaling(a, b): return _op(19, a, b)
def CppDivide(a, b): return CppDivideSignaling(a, b)
def ModulusNulling(a, b): return _op(26, a, b)
def ModulusSignaling(a, b): return _op(27, a, b)
def Modulus(a, b): return ModulusSignaling(a, b)
def Negate(a): return _op(36, a)
def And(a, b): return _op(40, a, b)
def Or(a, b): return _op(44, a, b)
def AndNot(a, b): return _op(48, a, b)
def Not(a): return _op(52, a)
def Xor(a, b): return _op(56, a, b)
def BitwiseAnd(a, b): return _op(60, a, b)
def BitwiseOr(a, b): return _op(64, a, b)
def BitwiseNot(a): return _op(68, a)
def BitwiseXor(a, b): return _op(72, a, b)
def ShiftLeft(a, b): return _op(76, a, b)
def ShiftRight(a, b): return _op(80, a, b)
def BitwiseAndNot(a, b): return _op(84, a, b)
def Equal(a, b): return _op(100, a, b)
def NotEqual(a, b): return _op(104, a, b)
def Less(a, b): return _op(116, a, b)
def LessOrEqual(a, b): return _op(120, a, b)
def Greater(a, b): return _op(L.OP_GREATER, a, b)
def GreaterOrEqual(a, b): return _op(L.OP_GREATER_OR_EQUAL, a, b)
# exact math family (expression/core/math_expressions.h:78-126, comparison_expressions.h IsOdd/IsEven)
def Abs(a): return _op(360, a)
def Round(a): return _op(300, a)
def Ceil(a): return _op(342, a)
def Floor(a): return _op(346, a)
def Trunc(a): return _op(304, a)
def RoundWithPrecision(a, precision): return _op(L.OP_ROUND_WITH_PRECISION, a, precision)   # round(a * 10^p) / 10^p
def RoundToInt(a): return _op(316, a)
def CeilToInt(a): return _op(308, a)
def FloorToInt(a): return _op(312, a)
def SqrtQuiet(a): return _op(333, a)
def SqrtNulling(a): return _op(334, a)
def SqrtSignaling(a): return _op(335, a)
# libm family (expression/core/math_expressions.h:30-140): results agree with the host libm to a few ULP
def Exp(a): return _op(320, a)
def LnQuiet(a): return _op(325, a)
def LnNulling(a): return _op(326, a)
def Log10Quiet(a): return _op(329, a)
def Log10Nulling(a): return _op(330, a)
def Log2Quiet(a): return _op(357, a)
def Log2Nulling(a): return _op(358, a)
def LogNulling(base, a): return DivideNulling(LnNulling(a), LnNulling(base))    # math_bound_expressions.cc:94-108
def LogQuiet(base, a): return DivideQuiet(LnQuiet(a), LnQuiet(base))            # :110-124
def PowerQuiet(a, b): return _op(353, a, b)
def PowerNulling(a, b): return _op(354, a, b)
def PowerSignaling(a, b): return _op(355, a, b)
def Sin(a): return _op(800, a)
def Cos(a): return _op(804, a)
def Tan(a): return _op(808, a)
def Cot(a): return DivideQuiet(ConstDouble(1.0), Tan(a))                        # :196-211
def Asin(a): return _op(812, a)
def Acos(a): return _op(816, a)
def Atan(a): return _op(820, a)
def Atan2(x, y): return _op(824, x, y)
def Sinh(a): return _op(828, a)
def Cosh(a): return _op(832, a)
def Tanh(a): return _op(836, a)
def Asinh(a): return _op(840, a)
def Acosh(a): return _op(844, a)
def Atanh(a): return _op(848, a)
def ToDegrees(a): return Multiply(a, ConstDouble(180.0 / 3.141592653589793))     # :285-296
def ToRadians(a): return Multiply(a, ConstDouble(3.141592653589793 / 180.0))     # :298-309
def Pi(): return ConstDouble(3.141592653589793)
def IsFinite(a): return _op(148, a)
def IsInf(a): return _op(152, a)
def IsNaN(a): return _op(156, a)
def IsNormal(a): return _op(160, a)
def IsOdd(a): return _op(140, a)
def IsEven(a): return _op(144, a)


class ExpressionList(object):
    """expression/base/expression.h: owning list of expressions (Case / In arguments)."""

    def __init__(self, *expressions):
        self.expressions = list(expressions)

    def add(self, e):
        self.expressions.append(e)
        return self


def _list(arguments):
    return list(arguments.expressions) if isinstance(arguments, ExpressionList) else list(arguments)


def Case(arguments):
    """CASE arg0 WHEN arg2 THEN arg3 [...] ELSE arg1 (elementary_expressions.h:91-93)."""
    return Expression(L.EXPR_OP, op=200, args=_list(arguments))


def In(needle, haystack):
    """needle IN (haystack...) with SQL NULL semantics (comparison_expressions.h:75-89)."""
    return Expression(L.EXPR_OP, op=208, args=[needle] + _list(haystack))


def If(c, t, e): return _op(204, c, t, e)
def NullingIf(c, t, e): return _op(L.OP_NULLING_IF, c, t, e)   # a NULL condition gives NULL (elementary_expressions.h:55-61)
def IfNull(a, b): return _op(220, a, b)
def IsNull(a): return _op(224, a)
def CastTo(data_type, a): return Expression(L.EXPR_CAST, dtype=data_type, args=[a])
def Alias(new_name, a): return Expression(L.EXPR_ALIAS, name=new_name, args=[a])


class CompoundExpression(Expression):
    def __init__(self):
        Expression.__init__(self, L.EXPR_COMPOUND)

    def Add(self, argument):
        self.args.append(argument)
        return self

    def AddAs(self, alias, argument):
        self.args.append(Alias(alias, argument))
        return self


# ------------------------------------------------------------------------- projectors
class SingleSourceProjector(object):
    def __init__(self, entries):
        self.entries = list(entries)  # (kind, position, name, alias)


def ProjectAllAttributes(prefix=None): return SingleSourceProjector([(L.PROJ_ALL, 0, None, prefix)])
def ProjectNamedAttribute(name): return SingleSourceProjector([(L.PROJ_NAMED, 0, name, None)])
def ProjectNamedAttributeAs(name, alias): return SingleSourceProjector([(L.PROJ_NAMED_AS, 0, name, alias)])
def ProjectAttributeAt(position): return SingleSourceProjector([(L.PROJ_AT, position, None, None)])
def ProjectNamedAttributes(names): return SingleSourceProjector([(L.PROJ_NAMED, 0, n, None) for n in names])


class CompoundSingleSourceProjector(SingleSourceProjector):
    def __init__(self):
        SingleSourceProjector.__init__(self, [])

    def add(self, projector):
        self.entries.extend(projector.entries)
        return self


class CompoundMultiSourceProjector(object):
    """base/infrastructure/projector.h:422-441: (source index, SingleSourceProjector) pairs."""

    def __init__(self):
        self.entries = []      # (source, kind, position, name, alias)

    def add(self, source_index, projector):
        for (k, pos, name, alias) in projector.entries:
            self.entries.append((int(source_index), k, pos, name, alias))
        return self


INNER, LEFT_OUTER = L.JOIN_INNER, L.JOIN_LEFT_OUTER
NOT_UNIQUE, UNIQUE = L.KEYS_NOT_UNIQUE, L.KEYS_UNIQUE


class AggregationSpecification(object):
    """cursor/core/aggregate.h:28-130."""

    def __init__(self):
        self.elements = []

    def AddAggregation(self, aggregation, input_name, output_name):
        self.elements.append((aggregation, 0, -1, input_name, output_name))
        return self

    def AddDistinctAggregation(self, aggregation, input_name, output_name):
        self.elements.append((aggregation, 1, -1, input_name, output_name))
        return self

    def AddAggregationWithDefinedOutputType(self, aggregation, input_name, output_name, output_type):
        self.elements.append((aggregation, 0, output_type, input_name, output_name))
        return self

    def AddDistinctAggregationWithDefinedOutputType(self, aggregation, input_name, output_name, output_type):
        self.elements.append((aggregation, 1, output_type, input_name, output_name))
        return self


class GroupAggregateOptions(object):
    """cursor/core/aggregate.h:160-205.  max_unique_keys_in_result: the result keeps the first (limit + 1) distinct keys in
    first-seen order; every row whose key is not among them is aggregated into the LAST of those rows
    (row_hash_set.cc:500-511).  Default kint64max = no limit.  The memory quota / estimated row count have no device
    counterpart (tables are sized by run feedback)."""
    NO_LIMIT = (1 << 63) - 1

    def __init__(self):
        self.max_unique_keys_in_result = self.NO_LIMIT
        self.memory_quota = None          # bytes; None = no quota (the reference's default: numeric_limits<size_t>::max())

    def set_max_unique_keys_in_result_(self, n):     # (the reference's spelling, trailing underscore included)
        self.max_unique_keys_in_result = int(n)
        return self

    def set_memory_quota(self, nbytes):
        """aggregate.h:170-175.  BestEffortGroupAggregate: the result block of a view holds quota / (bytes of a result row) groups;
        GroupAggregate tables are sized by run feedback and ignore it."""
        self.memory_quota = int(nbytes)
        return self

    def set_enforce_quota(self, _flag):
        return self

    def set_estimated_result_row_count(self, _n):
        return self

    def _option0(self):
        """ssgpu_op.option0: 0 = no limit, n > 0 = limit n, -1 = limit 0."""
        n = self.max_unique_keys_in_result
        return 0 if n >= self.NO_LIMIT else (-1 if n == 0 else n)


class SortOrder(object):
    """cursor/infrastructure/ordering.h:48-101."""

    def __init__(self):
        self.keys = []

    def add(self, projector_or_name, column_order):
        if isinstance(projector_or_name, SingleSourceProjector):
            for (_k, _p, name, _a) in projector_or_name.entries:
                self.keys.append((name, column_order))
        else:
            self.keys.append((projector_or_name, column_order))
        return self


# ------------------------------------------------------------------------- operations
class _Builder(object):
    """Flattens an operation tree into the ssgpu_plan_desc arrays."""

    def __init__(self):
        self.exprs, self.expr_args, self.projs, self.aggs, self.sortkeys, self.ops = [], [], [], [], [], []
        self.keep = []
        self.scan = None
        self.scan_aux = None         # rhs table of a HashJoin (auxiliary input)
        self.aux = False             # emitting the rhs subtree
        self.strings = StringDictionary([])

    def s(self, text):
        if text is None:
            return None
        b = text.encode() if isinstance(text, str) else text
        self.keep.append(b)
        return b

    def expr(self, e):
        child = [self.expr(a) for a in e.args]
        first = len(self.expr_args)
        self.expr_args.extend(child)
        i64 = int(e.i64)
        if e.kind == L.EXPR_CONST and e.dtype == STRING:
            i64 = self.strings.code_of(e.sval)       # dictionary code of the constant
        x = L.Expr(e.kind, e.op, e.dtype, first, len(child), 0, i64, float(e.f64), self.s(e.name))
        self.exprs.append(x)
        return len(self.exprs) - 1

    def proj(self, p):
        first = len(self.projs)
        for entry in p.entries:
            source = 0
            if len(entry) == 5:
                source, entry = entry[0], entry[1:]
            (k, pos, name, alias) = entry
            self.projs.append(L.Proj(k, pos, self.s(name), self.s(alias), source, 0))
        return first, len(p.entries)

    def aggspec(self, spec):
        first = len(self.aggs)
        for (agg, distinct, otype, inp, outp) in spec.elements:
            self.aggs.append(L.Agg(agg, distinct, otype, 0, self.s(inp), self.s(outp)))
        return first, len(spec.elements)

    def sort(self, order):
        first = len(self.sortkeys)
        for (name, o) in order.keys:
            self.sortkeys.append(L.SortKey(self.s(name), o, 0))
        return first, len(order.keys)

    def op(self, **kw):
        o = L.Op(kw.get("kind"), kw.get("child", -1), kw.get("expr", -1), kw.get("proj_first", 0), kw.get("proj_n", 0),
                 kw.get("agg_first", 0), kw.get("agg_n", 0), kw.get("sort_first", 0), kw.get("sort_n", 0),
                 kw.get("child2", -1), kw.get("option0", 0), kw.get("proj2_first", 0), kw.get("proj2_n", 0),
                 kw.get("proj3_first", 0), kw.get("proj3_n", 0))
        self.ops.append(o)
        return len(self.ops) - 1


class Operation(object):
    buffer_allocator = None

    def _emit(self, b):
        raise NotImplementedError

    def SetBufferAllocator(self, allocator, cascade_to_children=True):
        """Operation::SetBufferAllocator (cursor/base/operation.h:66-76): the cursor created from this operation
        allocates its device buffers against the allocator's quota (a MemoryLimit); running past it fails with
        ERROR_MEMORY_EXCEEDED.  One plan = one device quota, so `cascade_to_children` has no separate meaning here."""
        self.buffer_allocator = allocator
        return self

    def CreateCursor(self, context=None):
        """Operation::CreateCursor (cursor/base/operation.h:62): binds the whole tree."""
        return Cursor(self, context or Context.default())


class ScanView(Operation):
    def __init__(self, view):
        self.view = view

    def _emit(self, b):
        if b.aux:
            b.scan_aux = self.view
            return b.op(kind=L.OP_SCAN, option0=1)
        b.scan = self.view
        return b.op(kind=L.OP_SCAN)


class Compute(Operation):
    def __init__(self, expression, child):
        self.expression, self.child = expression, child

    def _emit(self, b):
        c = self.child._emit(b)
        return b.op(kind=L.OP_COMPUTE, child=c, expr=b.expr(self.expression))


class Project(Operation):
    def __init__(self, projector, child):
        self.projector, self.child = projector, child

    def _emit(self, b):
        c = self.child._emit(b)
        pf, pn = b.proj(self.projector)
        return b.op(kind=L.OP_PROJECT, child=c, proj_first=pf, proj_n=pn)


class Filter(Operation):
    def __init__(self, predicate, projector, child):
        self.predicate, self.projector, self.child = predicate, projector, child

    def _emit(self, b):
        c = self.child._emit(b)
        pf, pn = b.proj(self.projector)
        return b.op(kind=L.OP_FILTER, child=c, expr=b.expr(self.predicate), proj_first=pf, proj_n=pn)


class ScalarAggregate(Operation):
    def __init__(self, aggregation_specification, child):
        self.spec, self.child = aggregation_specification, child

    def _emit(self, b):
        c = self.child._emit(b)
        af, an = b.aggspec(self.spec)
        return b.op(kind=L.OP_SCALAR_AGGREGATE, child=c, agg_first=af, agg_n=an)


class GroupAggregate(Operation):
    def __init__(self, group_by, aggregation, options, child):
        self.group_by, self.spec, self.options, self.child = group_by, aggregation, options, child

    def _emit(self, b):
        c = self.child._emit(b)
        pf, pn = b.proj(self.group_by)
        af, an = b.aggspec(self.spec)
        return b.op(kind=L.OP_GROUP_AGGREGATE, child=c, proj_first=pf, proj_n=pn, agg_first=af, agg_n=an,
                    option0=(self.options._option0() if self.options else 0))


class BestEffortGroupAggregate(Operation):
    """cursor/core/aggregate.h:230-250: groups and aggregates as many input rows as the result block holds, returns them, and starts
    anew with the input it had not consumed -- rows are key-unique within each returned view, not across views; an input of any
    size is processed (no ERROR_MEMORY_EXCEEDED).  options.memory_quota bounds the block (include/ssgpu.h
    ssgpu_plan_run_best_effort); without one the result is the GroupAggregate's."""

    def __init__(self, group_by, aggregation, options, child):
        self.group_by, self.spec, self.options, self.child = group_by, aggregation, options, child

    def _emit(self, b):
        c = self.child._emit(b)
        pf, pn = b.proj(self.group_by)
        af, an = b.aggspec(self.spec)
        quota = self.options.memory_quota if (self.options is not None and self.options.memory_quota is not None) else 0
        return b.op(kind=L.OP_BEST_EFFORT_GROUP_AGGREGATE, child=c, proj_first=pf, proj_n=pn, agg_first=af, agg_n=an, option0=max(0, min(int(quota), (1 << 62))))


class AggregateClusters(Operation):
    def __init__(self, clustered_by_columns, aggregation, child):
        self.group_by, self.spec, self.child = clustered_by_columns, aggregation, child

    def _emit(self, b):
        c = self.child._emit(b)
        pf, pn = b.proj(self.group_by)
        af, an = b.aggspec(self.spec)
        return b.op(kind=L.OP_AGGREGATE_CLUSTERS, child=c, proj_first=pf, proj_n=pn, agg_first=af, agg_n=an)


class Sort(Operation):
    def __init__(self, sort_order, result_projector, memory_limit, child):
        self.order, self.projector, self.memory_limit, self.child = sort_order, result_projector, memory_limit, child

    def _emit(self, b):
        c = self.child._emit(b)
        pf, pn = b.proj(self.projector if self.projector is not None else ProjectAllAttributes())
        sf, sn = b.sort(self.order)
        return b.op(kind=L.OP_SORT, child=c, proj_first=pf, proj_n=pn, sort_first=sf, sort_n=sn,
                    option0=int(self.memory_limit or 0))


class HashJoinOperation(Operation):
    """cursor/core/hash_join.h:37-56.  On the device the rhs must be ScanView(table) -- a resident
    dimension table -- and its keys UNIQUE; the probe is fused into the lhs pipeline."""

    def __init__(self, join_type, lhs_key_selector, rhs_key_selector, result_projector, rhs_key_uniqueness, lhs_child, rhs_child):
        self.join_type, self.lhs_keys, self.rhs_keys = join_type, lhs_key_selector, rhs_key_selector
        self.result_projector, self.uniqueness = result_projector, rhs_key_uniqueness
        self.child, self.rhs_child = lhs_child, rhs_child

    def _emit(self, b):
        c = self.child._emit(b)
        b.aux = True
        r = self.rhs_child._emit(b)
        b.aux = False
        pf, pn = b.proj(self.lhs_keys)
        p2f, p2n = b.proj(self.rhs_keys)
        p3f, p3n = b.proj(self.result_projector)
        return b.op(kind=L.OP_HASH_JOIN, child=c, child2=r, proj_first=pf, proj_n=pn, proj2_first=p2f, proj2_n=p2n,
                    proj3_first=p3f, proj3_n=p3n, option0=int(self.join_type) | (int(self.uniqueness) << 8))


def HashJoin(join_type, lhs_key_selector, rhs_key_selector, result_projector, rhs_key_uniqueness, lhs_child, rhs_child):
    return HashJoinOperation(join_type, lhs_key_selector, rhs_key_selector, result_projector, rhs_key_uniqueness, lhs_child, rhs_child)


def _array(ctype, items):
    arr = (ctype * max(len(items), 1))()
    for i, it in enumerate(items):
        arr[i] = it
    return arr


def collect_strings(operation):
    """Every STRING value a plan can meet: ConstString payloads and the STRING columns of the host
    Views it scans."""
    found = []

    def walk_expr(e):
        if e is None:
            return
        if getattr(e, "sval", None) is not None:
            found.append(e.sval)
        for a in getattr(e, "args", ()):
            walk_expr(a)

    def walk_op(o):
        if o is None:
            return
        for attr in ("expression", "predicate"):
            walk_expr(getattr(o, attr, None))
        view = getattr(o, "view", None)
        if isinstance(view, View):
            schema = view.schema()
            for i in range(schema.attribute_count()):
                if schema.attribute(i).type() == STRING:
                    col = view.column(i)
                    for j, v in enumerate(col.data):
                        if col.is_null is None or not col.is_null[j]:
                            found.append(v)
        walk_op(getattr(o, "child", None))
        walk_op(getattr(o, "rhs_child", None))
    walk_op(operation)
    return found


def _find_allocator(operation):
    """The allocator set on the nearest operation from the root (SetBufferAllocator cascades downwards)."""
    o = operation
    while o is not None:
        if getattr(o, "buffer_allocator", None) is not None:
            return o.buffer_allocator
        o = getattr(o, "child", None)
    return None


class BufferAllocator(object):
    """BufferAllocator over the C ABI's pinned-host allocator (base/memory/memory.h:100-233): Allocate /
    BestEffortAllocate / Reallocate / Free / Available, with an optional soft quota (MemoryLimit, memory.h:465).
    Buffers are pinned host memory when the context has a device (what the host<->device copies of a View want),
    plain 256-byte aligned host memory on a bind-only context."""

    def __init__(self, quota=None, context=None):
        self.ctx = context or Context.default()
        self.lib = self.ctx.lib
        h = C.c_void_p()
        self.ctx.check(self.lib.ssgpu_allocator_create(self.ctx.handle, -1 if quota is None else int(quota), C.byref(h)))
        self.handle = h

    def BestEffortAllocate(self, requested, minimal):
        """-> (address, granted bytes) or None when `minimal` does not fit the quota (a NULL Buffer)."""
        p, g = C.c_void_p(), C.c_size_t()
        rc = self.lib.ssgpu_allocator_allocate(self.handle, int(requested), int(minimal), C.byref(p), C.byref(g))
        if rc == L.ERROR_MEMORY_EXCEEDED:
            return None
        self.ctx.check(rc)
        return (p.value, g.value)

    def Allocate(self, requested):
        return self.BestEffortAllocate(requested, requested)

    def Reallocate(self, address, requested, minimal=None):
        p, g = C.c_void_p(), C.c_size_t()
        rc = self.lib.ssgpu_allocator_reallocate(self.handle, C.c_void_p(address), int(requested),
                                                 int(requested if minimal is None else minimal), C.byref(p), C.byref(g))
        if rc == L.ERROR_MEMORY_EXCEEDED:
            return None
        self.ctx.check(rc)
        return (p.value, g.value)

    def Free(self, address):
        self.lib.ssgpu_allocator_free(self.handle, C.c_void_p(address))

    def Available(self):
        return self.lib.ssgpu_allocator_available(self.handle)

    def GetUsage(self):
        return self.lib.ssgpu_allocator_allocated(self.handle)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ssgpu_allocator_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def HeapBufferAllocator(context=None):
    """HeapBufferAllocator::Get() (memory.h:240): no quota."""
    return BufferAllocator(None, context)


def MemoryLimit(quota, context=None):
    """MemoryLimit(quota) (memory.h:465-520): a soft quota in bytes."""
    return BufferAllocator(quota, context)


class Plan(object):
    """A bound plan (ssgpu_plan): owns the device programs and result buffers."""

    def __init__(self, operation, context, extra_strings=None):
        """extra_strings: further byte strings for the plan's dictionary -- what a sharded job passes so that every rank
        builds the SAME dictionary (codes of STRING columns are then comparable across ranks)."""
        self.ctx = context
        self.lib = context.lib
        b = _Builder()
        b.strings = self.strings = StringDictionary(list(collect_strings(operation)) + [_as_bytes(v) for v in (extra_strings or [])])
        operation._emit(b)
        if b.scan is None:
            raise SupersonicException(L.ERROR_INVALID_ARGUMENT_VALUE, "plan has no ScanView")
        self.input = b.scan
        schema = b.scan.schema()
        attrs = [L.Attr(b.s(schema.attribute(i).name()), schema.attribute(i).type(), schema.attribute(i).nullability())
                 for i in range(schema.attribute_count())]
        self._keep = b
        d = L.PlanDesc()
        self._arrays = [_array(L.Attr, attrs), _array(L.Op, b.ops), _array(L.Expr, b.exprs),
                        _array(C.c_int32, b.expr_args), _array(L.Proj, b.projs), _array(L.Agg, b.aggs),
                        _array(L.SortKey, b.sortkeys)]
        d.input_schema, d.n_attrs = self._arrays[0], len(attrs)
        d.ops, d.n_ops = self._arrays[1], len(b.ops)
        d.exprs, d.n_exprs = self._arrays[2], len(b.exprs)
        d.expr_args, d.n_expr_args = self._arrays[3], len(b.expr_args)
        d.projs, d.n_projs = self._arrays[4], len(b.projs)
        d.aggs, d.n_aggs = self._arrays[5], len(b.aggs)
        d.sortkeys, d.n_sortkeys = self._arrays[6], len(b.sortkeys)
        self.aux_input = b.scan_aux
        if b.scan_aux is not None:
            asch = b.scan_aux.schema()
            aux_attrs = [L.Attr(b.s(asch.attribute(i).name()), asch.attribute(i).type(), asch.attribute(i).nullability())
                         for i in range(asch.attribute_count())]
            self._arrays.append(_array(L.Attr, aux_attrs))
            d.aux_schema, d.n_aux_attrs = self._arrays[-1], len(aux_attrs)
        h = C.c_void_p()
        rc = self.lib.ssgpu_plan_create(context.handle, C.byref(d), C.byref(h))
        context.check(rc)
        self._adopt(h)
        self.lib.ssgpu_plan_set_dict(h, self.strings.handle)      # CONCAT prints STRING inputs through the plan's dictionary
        alloc = _find_allocator(operation)
        if alloc is not None:
            self.set_buffer_allocator(alloc)

    def _adopt(self, h):
        self.handle = h
        n = self.lib.ssgpu_plan_attr_count(h)
        out = []
        for i in range(n):
            a = L.Attr()
            self.lib.ssgpu_plan_attr(h, i, C.byref(a))
            out.append(Attribute(a.name.decode(), a.dtype, a.nullable))
        self.result_schema = TupleSchema(out)
        self._block = None
        self._block_key = None
        self._aux_block = None
        self._aux_block_key = None

    def describe(self):
        return self.lib.ssgpu_plan_describe(self.handle).decode()

    def set_option(self, key, value):
        """ssgpu_plan_set_option: an option of this plan alone ("lazy_feedback")."""
        self.ctx.check(self.lib.ssgpu_plan_set_option(self.handle, key.encode(), int(value)))
        return self

    def set_memory_limit(self, nbytes):
        """Soft quota on the device memory this plan holds (ssgpu_plan_set_memory_limit); None or < 0 = unlimited."""
        self.ctx.check(self.lib.ssgpu_plan_set_memory_limit(self.handle, -1 if nbytes is None else int(nbytes)))

    def set_buffer_allocator(self, allocator):
        q = allocator.Available() if allocator is not None else None
        self.set_memory_limit(None if (q is None or q >= (1 << 62)) else q)

    def memory_in_use(self):
        return self.lib.ssgpu_plan_memory_in_use(self.handle)

    def specialized(self):
        """Specialised kernels (runtime compilation, csrc/rtc.cpp) this plan currently holds."""
        return self.lib.ssgpu_plan_specialized(self.handle)

    def stage_info(self):
        """ssgpu_plan_stage_info of every stage: what the last run did (execution shape, reruns, radix passes), as dicts."""
        out = []
        for i in range(self.lib.ssgpu_plan_stage_count(self.handle)):
            st = L.StageInfo()
            self.ctx.check(self.lib.ssgpu_plan_stage_info(self.handle, i, C.byref(st)))
            out.append({n: getattr(st, n) for n, _t in L.StageInfo._fields_ if n != "reserved"})
        return out

    def specialize(self):
        """ssgpu_plan_specialize: run kernels compiled for this plan; what can be compiled without a run is compiled now."""
        self.ctx.check(self.lib.ssgpu_plan_specialize(self.handle))
        return self

    def specialize_reason(self):
        """Why a stage that asked for a specialised kernel did not get one ("" if none was refused)."""
        return (self.lib.ssgpu_plan_specialize_reason(self.handle) or b"").decode()

    def program(self, stage=0):
        """Raw VM instructions of a stage (debug hook used by tests/vm_emulator.py)."""
        ptr, n, nb = C.c_void_p(), C.c_int32(), C.c_int32()
        self.ctx.check(self.lib.ssgpu_plan_program(self.handle, stage, C.byref(ptr), C.byref(n), C.byref(nb)))
        return C.string_at(ptr, n.value * nb.value), n.value, nb.value

    # -- input staging ------------------------------------------------------------
    def _columns_for(self, view, slot="_block"):
        if isinstance(view, DeviceView):
            cols = (L.Column * max(len(view._ptrs), 1))()
            for i, (dp, npn) in enumerate(view._ptrs):
                cols[i].data = dp
                cols[i].is_null = npn or None
            return cols, len(view._ptrs), view.row_count()
        # host View: stage into a device Block on the copy stream (pinned staging is the
        # caller's choice; numpy memory is pageable, which only makes the copy synchronous)
        # the staged block is reused only for the SAME view object (a strong reference is kept: the id of a freed
        # temporary View is readily reused by the next one, which must not inherit its device data)
        if getattr(self, slot) is None or getattr(self, slot + "_key") is not view:
            if getattr(self, slot) is not None:
                self.lib.ssgpu_block_destroy(getattr(self, slot))
                setattr(self, slot, None)
            schema = view.schema()
            attrs = _array(L.Attr, [L.Attr(schema.attribute(i).name().encode(), schema.attribute(i).type(),
                                           schema.attribute(i).nullability()) for i in range(schema.attribute_count())])
            blk = C.c_void_p()
            self.ctx.check(self.lib.ssgpu_block_create(self.ctx.handle, attrs, schema.attribute_count(),
                                                       max(view.row_count(), 1), C.byref(blk)))
            for i in range(view.column_count()):
                col = view.column(i)
                nulls = col.is_null
                if schema.attribute(i).type() == STRING:
                    col = Column(self.strings.encode(col.data, nulls), nulls)
                if view.row_count():
                    self.ctx.check(self.lib.ssgpu_block_upload(
                        blk, i, col.data.ctypes.data_as(C.c_void_p),
                        None if nulls is None else nulls.ctypes.data_as(C.c_void_p), 0, view.row_count()))
            self.lib.ssgpu_block_set_row_count(blk, view.row_count())
            self.ctx.synchronize()
            setattr(self, slot, blk); setattr(self, slot + "_key", view)
        n = view.schema().attribute_count()
        cols = (L.Column * max(n, 1))()
        for i in range(n):
            self.lib.ssgpu_block_column(getattr(self, slot), i, C.byref(cols[i]))
        return cols, n, view.row_count()

    # -- execution ------------------------------------------------------------------
    def _bind_aux(self):
        if self.aux_input is not None:
            cols, n, rows = self._columns_for(self.aux_input, "_aux_block")
            self._aux_cols = cols    # keep the ctypes array alive
            self.ctx.check(self.lib.ssgpu_plan_set_aux_input(self.handle, cols, n, rows))

    def run(self, view=None):
        self._bind_aux()
        cols, n, rows = self._columns_for(view if view is not None else self.input)
        res = C.c_void_p()
        self.ctx.check(self.lib.ssgpu_plan_run(self.handle, cols, n, rows, C.byref(res)))
        self._result = res
        return res

    def run_best_effort(self, start_row=0, view=None):
        """ssgpu_plan_run_best_effort: one view of a BestEffortGroupAggregate -- the aggregate over the longest run of input rows from
        start_row on whose keys fit the result block.  Returns the row the next view starts at (== the input's row count: done)."""
        self._bind_aux()
        cols, n, rows = self._columns_for(view if view is not None else self.input)
        res, nxt = C.c_void_p(), C.c_int64()
        self.ctx.check(self.lib.ssgpu_plan_run_best_effort(self.handle, cols, n, rows, int(start_row), C.byref(nxt), C.byref(res)))
        self._result = res
        return nxt.value

    def run_host(self, view=None, chunk_rows=0):
        """Chunked staging (ssgpu_plan_run_host): the HOST columns of `view` (default: the plan's input) travel through two alternating
        sets of device columns of chunk_rows rows, chunk k + 1 being copied while chunk k is read -- inputs larger than device
        memory run, and the copy overlaps the kernels (pinned numpy memory, e.g. torch's pin_memory, for the copies to be
        asynchronous).  ScalarAggregate plans, row-local plans (Filter / Compute / Project / HashJoin) and plans whose first blocking
        operation is a GroupAggregate of mergeable aggregates (include/ssgpu.h, "CHUNKED STAGING"); NOT_IMPLEMENTED otherwise."""
        view = view if view is not None else self.input
        if isinstance(view, DeviceView):
            raise SupersonicException(ERROR_INVALID_ARGUMENT_VALUE, "run_host takes a host View (device columns: run)")
        self._bind_aux()                         # (a HashJoin's rhs table: device-resident, bound as for run)
        schema = view.schema()
        n = schema.attribute_count()
        cols = (L.Column * max(n, 1))()
        keep = []                                # the arrays whose memory the asynchronous copies read
        for i in range(n):
            col = view.column(i)
            nulls = col.is_null
            data = col.data
            if schema.attribute(i).type() == STRING:
                data = self.strings.encode(data, nulls)
            data = np.ascontiguousarray(data, dtype=np.int32 if schema.attribute(i).type() == STRING else _NP[schema.attribute(i).type()])
            nulls = None if nulls is None else np.ascontiguousarray(nulls, dtype=np.bool_)
            keep.append((data, nulls))
            cols[i].data = data.ctypes.data
            cols[i].is_null = None if nulls is None else nulls.ctypes.data
        res = C.c_void_p()
        self.ctx.check(self.lib.ssgpu_plan_run_host(self.handle, cols, n, view.row_count(), int(chunk_rows), C.byref(res)))
        self.ctx.synchronize()                   # (the host arrays may go once the streams have drained)
        self._result = res
        return res

    def chunked_form(self):
        """ssgpu_plan_chunked_form: how run_host / stream would take this plan -- (1, ...) ScalarAggregate state fold, (2, ...) row-local plan,
        results appended, (3, per-chunk plan, merging plan) GroupAggregate partial tables + one merge; raises what they would raise."""
        kind, head, tail = C.c_int32(0), C.c_char_p(), C.c_char_p()
        self.ctx.check(self.lib.ssgpu_plan_chunked_form(self.handle, C.byref(kind), C.byref(head), C.byref(tail)))
        return kind.value, (head.value or b"").decode(), (tail.value or b"").decode()

    def stream(self, views, chunk_rows=0):
        """The push form of chunked staging (ssgpu_plan_stream_begin / _push / _finish): `views` is an iterable of host Views with the
        plan's input schema -- the blocks a child cursor hands out -- each of which may be overwritten as soon as the next is asked for."""
        self._bind_aux()
        self.ctx.check(self.lib.ssgpu_plan_stream_begin(self.handle, int(chunk_rows)))
        for view in views:
            schema = view.schema()
            n = schema.attribute_count()
            cols = (L.Column * max(n, 1))()
            keep = []
            for i in range(n):
                col = view.column(i)
                data, nulls = col.data, col.is_null
                if schema.attribute(i).type() == STRING:
                    data = self.strings.encode(data, nulls)
                data = np.ascontiguousarray(data, dtype=np.int32 if schema.attribute(i).type() == STRING else _NP[schema.attribute(i).type()])
                nulls = None if nulls is None else np.ascontiguousarray(nulls, dtype=np.bool_)
                keep.append((data, nulls))
                cols[i].data = data.ctypes.data
                cols[i].is_null = None if nulls is None else nulls.ctypes.data
            self.ctx.check(self.lib.ssgpu_plan_stream_push(self.handle, cols, n, view.row_count()))
        res = C.c_void_p()
        self.ctx.check(self.lib.ssgpu_plan_stream_finish(self.handle, C.byref(res)))
        self._result = res
        return res

    def run_partial(self, view, global_row_offset=0):
        cols, n, rows = self._columns_for(view)
        self.ctx.check(self.lib.ssgpu_plan_run_partial(self.handle, cols, n, rows, global_row_offset))
        segs = (L.PartialSegment * 16)()
        k = self.lib.ssgpu_plan_partial_segments(self.handle, segs, 16)
        return [(segs[i].device_ptr, segs[i].count, segs[i].dtype, segs[i].reduce) for i in range(k)]

    # ---- dense-slot GroupAggregate across ranks (ssgpu.h: ssgpu_plan_key_ranges ... ssgpu_plan_dense_grow) ----------------
    def key_ranges(self, view):
        """Value ranges of the group keys over `view` (united with what this plan has seen before): [(lo, hi)] in the order-
        preserving unsigned domain of ssgpu.h (lo > hi: no value).  NOT_IMPLEMENTED for plans the dense form cannot take."""
        cols, n, rows = self._columns_for(view)
        nk = C.c_int32(0)
        lo = (C.c_uint64 * 8)()
        hi = (C.c_uint64 * 8)()
        self.ctx.check(self.lib.ssgpu_plan_key_ranges(self.handle, cols, n, rows, C.byref(nk), lo, hi))
        return [(int(lo[k]), int(hi[k])) for k in range(nk.value)]

    def set_dense(self, ranges, n_chunks):
        """Lay the group table out for these key ranges, in n_chunks slot ranges (one per rank); returns the layout."""
        n = len(ranges)
        lo = (C.c_uint64 * max(n, 1))(*[r[0] for r in ranges])
        hi = (C.c_uint64 * max(n, 1))(*[r[1] for r in ranges])
        lay = L.DenseLayout()
        self.ctx.check(self.lib.ssgpu_plan_set_dense(self.handle, n, lo, hi, n_chunks, C.byref(lay)))
        return {f: getattr(lay, f) for f, _t in L.DenseLayout._fields_}

    def run_dense(self, view, table_ptr):
        """This shard's partial table into the caller's chunked device buffer (n_chunks * chunk_bytes); nothing is read back."""
        cols, n, rows = self._columns_for(view)
        self.ctx.check(self.lib.ssgpu_plan_run_dense(self.handle, cols, n, rows, C.c_void_p(table_ptr)))

    def fold_dense(self, chunks_ptr, n_chunks):
        """Fold the n_chunks images of the slot range this rank owns (what the all-to-all delivered) and extract its groups."""
        res = C.c_void_p()
        self.ctx.check(self.lib.ssgpu_plan_fold_dense(self.handle, C.c_void_p(chunks_ptr), n_chunks, C.byref(res)))
        self._result = res
        return res

    def dense_flags(self):
        """(flags, error) the last fold's chunk headers carried; synchronises.  flags: 2 = a record segment ran full, 4 = a key
        outside the ranges -- on SOME rank; every rank reads the same words."""
        f = C.c_uint32(0)
        e = C.c_uint32(0)
        self.ctx.check(self.lib.ssgpu_plan_dense_flags(self.handle, C.byref(f), C.byref(e)))
        return f.value, e.value

    def dense_grow(self):
        self.ctx.check(self.lib.ssgpu_plan_dense_grow(self.handle))

    def dense_fail(self, table_ptr, code):
        """This rank's run failed with `code`: flag every chunk of its table, so that the step's collective still happens and every
        rank learns of the failure from the headers."""
        self.ctx.check(self.lib.ssgpu_plan_dense_fail(self.handle, C.c_void_p(table_ptr), int(code)))

    def fold_partials(self, images_ptr, n_images):
        """Fold n_images all-gathered images of the partial state (device pointer) into this plan's state."""
        self.ctx.check(self.lib.ssgpu_plan_fold_partials(self.handle, C.c_void_p(images_ptr), n_images))

    def finalize(self):
        res = C.c_void_p()
        self.ctx.check(self.lib.ssgpu_plan_finalize(self.handle, C.byref(res)))
        self._result = res
        return res

    def fold_finalize(self, images_ptr, n_images):
        """fold_partials + finalize as ONE kernel launch (ssgpu_plan_fold_finalize): the step of a sharded scalar aggregate
        after its collective."""
        res = C.c_void_p()
        self.ctx.check(self.lib.ssgpu_plan_fold_finalize(self.handle, C.c_void_p(images_ptr), n_images, C.byref(res)))
        self._result = res
        return res

    def write_file(self, path, res=None):
        """FileOutput(path)->Write(result view): the finished result in the reference's file format."""
        rs = self.result_schema
        if any(rs.attribute(i).type() in (STRING, BINARY) for i in range(rs.attribute_count())):
            out = FileOutput(path)          # variable-length values leave through the dictionary: fetched, then written
            out.Write(self.fetch(res))
            out.Finalize()
            return
        self.ctx.check(self.lib.ssgpu_result_write_file(res or self._result, path.encode()))

    def result_row_count(self, res=None):
        """ssgpu_result_row_count of the last run's result (waits for the run; nothing is copied)."""
        rows = self.lib.ssgpu_result_row_count(res or self._result)
        if rows < 0:
            raise SupersonicException(L.ERROR_HIP, self.ctx.last_error())
        return int(rows)

    def fetch(self, res=None, nulls_of_result=False):
        """Copy the result to host: a View over numpy arrays.  nulls_of_result: a column has a NULL mask when the RESULT has one (a
        result of another plan with this plan's columns: DoEvaluate's skip form makes every column NULLABLE)."""
        res = res or self._result
        rows = self.lib.ssgpu_result_row_count(res)
        if rows < 0:
            raise SupersonicException(L.ERROR_HIP, self.ctx.last_error())
        cols = []
        for i in range(self.result_schema.attribute_count()):
            a = self.result_schema.attribute(i)
            dp, npn = C.c_void_p(), C.c_void_p()
            self.ctx.check(self.lib.ssgpu_result_column(res, i, C.byref(dp), C.byref(npn)))
            dt = np.dtype(np.int32 if a.type() == STRING else _NP[a.type()])
            data = np.frombuffer(C.string_at(dp, rows * dt.itemsize), dtype=dt).copy() if rows else np.zeros(0, dt)
            nulls = None
            if a.is_nullable() or (nulls_of_result and npn.value):
                nulls = (np.frombuffer(C.string_at(npn, rows), dtype=np.uint8).copy() != 0) if rows else np.zeros(0, bool)
            if a.type() == STRING:
                own = self.lib.ssgpu_result_column_dict(res, i)      # a CONCAT column: codes of the result's own dictionary
                if own:
                    dec = np.empty(len(data), dtype=object)
                    ptr, n = C.c_void_p(), C.c_int32()
                    for j, code in enumerate(data.tolist()):
                        if nulls is not None and nulls[j]:
                            dec[j] = b""
                        else:
                            self.ctx.check(self.lib.ssgpu_dict_decode(own, code, C.byref(ptr), C.byref(n)))
                            dec[j] = C.string_at(ptr, n.value)
                    data = dec
                else:
                    data = self.strings.decode(data, nulls)
            cols.append(Column(data, nulls))
        return View(self.result_schema, cols, rows)

    def result_device_view(self, res=None):
        """The result as a DeviceView over the plan's own output buffers (no copy): valid until the
        plan runs again or is destroyed -- keep the Plan alive while the view is in use."""
        res = res or self._result
        rows = self.lib.ssgpu_result_row_count(res)
        if rows < 0:
            raise SupersonicException(L.ERROR_HIP, self.ctx.last_error())
        ptrs = []
        for i in range(self.result_schema.attribute_count()):
            col = L.Column()
            self.ctx.check(self.lib.ssgpu_result_device_column(res, i, C.byref(col)))
            ptrs.append((col.data or 0, col.is_null or 0))
        return DeviceView(self.result_schema, ptrs, rows)

    # -- result images (ONE-collective exchange of materialised results, ssgpu.h) -----------
    def image_layout(self, capacity_rows, n_images=1):
        """(image_bytes, unpacked_bytes, offsets) -- offsets[i] = (image data, image NULL mask, unpacked data,
        unpacked NULL mask) byte offsets of attribute i (-1 = none); the last entry is the validity column."""
        n = self.result_schema.attribute_count()
        ib, ub = C.c_int64(), C.c_int64()
        offs = (C.c_int64 * (4 * (n + 1)))()
        self.ctx.check(self.lib.ssgpu_plan_image_layout(self.handle, capacity_rows, n_images, C.byref(ib), C.byref(ub), offs))
        return ib.value, ub.value, [tuple(offs[4 * i: 4 * i + 4]) for i in range(n + 1)]

    def pack_image(self, capacity_rows, image_ptr, res=None):
        """Pack the (device-resident) result into one image at the device pointer `image_ptr` (async)."""
        self.ctx.check(self.lib.ssgpu_result_pack_image(res or self._result, capacity_rows, C.c_void_p(image_ptr)))

    def route_images(self, n_keys, n_dest, capacity_rows, images_ptr, res=None):
        """Key-range exchange: route the (device-resident) result's rows into n_dest images at `images_ptr` by a hash of their
        first n_keys columns (async)."""
        self.ctx.check(self.lib.ssgpu_result_route_images(res or self._result, n_keys, n_dest, capacity_rows, C.c_void_p(images_ptr)))

    def unpack_images(self, images_ptr, n_images, capacity_rows, unpacked_ptr):
        """n_images gathered images of THIS plan's result schema -> the columns of one
        (n_images * capacity_rows)-row table at `unpacked_ptr` (async).  Returns a DeviceView whose
        schema is the result schema plus a trailing NOT NULL BOOL column "__valid"."""
        n = self.result_schema.attribute_count()
        cols = (L.Column * (n + 1))()
        self.ctx.check(self.lib.ssgpu_images_unpack(self.handle, C.c_void_p(images_ptr), n_images, capacity_rows,
                                                    C.c_void_p(unpacked_ptr), cols))
        attrs = [self.result_schema.attribute(i) for i in range(n)] + [Attribute("__valid", BOOL, NOT_NULLABLE)]
        return DeviceView(TupleSchema(attrs), [(cols[i].data or 0, cols[i].is_null or 0) for i in range(n + 1)],
                          n_images * capacity_rows)

    def counters(self):
        c = L.Counters()
        self.ctx.check(self.lib.ssgpu_plan_counters(self.handle, C.byref(c)))
        return c

    def recent_kernel_ms(self, max_runs=256):
        """Dominant-kernel durations (ms) of the most recent runs, oldest first (waits for the stream)."""
        buf = (C.c_double * max_runs)()
        n = self.lib.ssgpu_plan_recent_kernel_ms(self.handle, buf, max_runs)
        return [buf[i] for i in range(n)]

    def interrupt(self):
        self.lib.ssgpu_interrupt(self.handle)

    def _release(self):
        if getattr(self, "_aux_block", None):
            self.lib.ssgpu_block_destroy(self._aux_block)
        if getattr(self, "_block", None):
            self.lib.ssgpu_block_destroy(self._block)
        self._block = self._block_key = self._aux_block = self._aux_block_key = None
        if getattr(self, "handle", None):
            self.lib.ssgpu_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass


def _expr_strings(e, found):
    if getattr(e, "sval", None) is not None:
        found.append(e.sval)
    for a in getattr(e, "args", ()):
        _expr_strings(a, found)
    return found


class BoundExpressionTree(Plan):
    """BoundExpressionTree (expression/base/expression.h:96-145) over ssgpu_expr_bind / ssgpu_expr_evaluate:
    result_schema(), row_capacity(), Evaluate(view) -> ResultView."""

    def __init__(self, expression, input_schema, allocator, max_row_count, context):
        self.ctx, self.lib = context, context.lib
        self.expression, self.input_schema = expression, input_schema
        self.max_row_count = int(max_row_count or 0)
        self.allocator = allocator
        self.handle = None
        self._block = self._block_key = self._aux_block = self._aux_block_key = None
        self.aux_input = None
        self._has_strings = any(input_schema.attribute(i).type() == STRING for i in range(input_schema.attribute_count()))
        self._bind(_expr_strings(expression, []))

    def _bind(self, strings):
        """(Re)binds against a dictionary holding `strings`; a tree over STRING attributes is re-bound per View,
        because the codes of its constants depend on the View's values (one dictionary per evaluated View)."""
        if self.handle:
            self._release()
        b = _Builder()
        b.strings = self.strings = StringDictionary(strings)
        root = b.expr(self.expression)
        schema = self.input_schema
        attrs = _array(L.Attr, [L.Attr(b.s(schema.attribute(i).name()), schema.attribute(i).type(), schema.attribute(i).nullability())
                                for i in range(schema.attribute_count())])
        self._keep = (b, attrs, _array(L.Expr, b.exprs), _array(C.c_int32, b.expr_args))
        h = C.c_void_p()
        self.ctx.check(self.lib.ssgpu_expr_bind(self.ctx.handle, attrs, schema.attribute_count(), self._keep[2], len(b.exprs),
                                                self._keep[3], len(b.expr_args), root, self.max_row_count, C.byref(h)))
        self._adopt(h)
        if self.allocator is not None:
            self.set_buffer_allocator(self.allocator)

    def row_capacity(self):
        return self.lib.ssgpu_expr_row_capacity(self.handle)

    def Evaluate(self, view):
        """BoundExpressionTree::Evaluate(const View&) -> EvaluationResult (expression.cc:57-76): a ResultView holding
        the result View or the failure (ERROR_TOO_MANY_ROWS beyond row_capacity(), evaluation errors 300-399)."""
        try:
            if self._has_strings and isinstance(view, View):
                found = _expr_strings(self.expression, [])
                for i in range(self.input_schema.attribute_count()):
                    if self.input_schema.attribute(i).type() == STRING:
                        col = view.column(i)
                        found.extend(v for j, v in enumerate(col.data) if col.is_null is None or not col.is_null[j])
                self._bind(found)
            cols, n, rows = self._columns_for(view)
            res = C.c_void_p()
            self.ctx.check(self.lib.ssgpu_expr_evaluate(self.handle, cols, n, rows, C.byref(res)))
            self._result = res
            return ResultView(view=self.fetch(res))
        except SupersonicException as e:
            return ResultView(exception=e)


    def DoEvaluate(self, view, skip_vectors):
        """BoundExpression::DoEvaluate(const View& input, const BoolView& skip_vectors) (expression/base/expression.h:46-92) over
        ssgpu_expr_evaluate_skip: one skip vector per result attribute -- a numpy bool array (uploaded; updated IN PLACE with the result's
        NULLs on return), a device pointer (int; the library writes the NULLs back to it), or None (nothing skipped).  A row whose skip byte
        is set is not evaluated: NULL result, no failure from a signalling operator."""
        try:
            n_out = self.result_schema.attribute_count()
            if len(skip_vectors) != n_out:
                raise SupersonicException(L.ERROR_ATTRIBUTE_COUNT_MISMATCH, "one skip vector per result attribute")
            cols, n, rows = self._columns_for(view)
            host = [i for i, v in enumerate(skip_vectors) if isinstance(v, np.ndarray)]
            ptrs = (C.c_void_p * max(n_out, 1))()
            blk = None
            if host:
                attrs = _array(L.Attr, [L.Attr(("s%d" % i).encode(), BOOL, NOT_NULLABLE) for i in host])
                blk = C.c_void_p()
                self.ctx.check(self.lib.ssgpu_block_create(self.ctx.handle, attrs, len(host), max(rows, 1), C.byref(blk)))
                for k, i in enumerate(host):
                    arr = np.ascontiguousarray(skip_vectors[i], dtype=np.bool_)
                    if len(arr) != rows:
                        raise SupersonicException(L.ERROR_INVALID_ARGUMENT_VALUE, "a skip vector has one byte per input row")
                    if rows:
                        self.ctx.check(self.lib.ssgpu_block_upload(blk, k, arr.ctypes.data_as(C.c_void_p), None, 0, rows))
                    col = L.Column()
                    self.ctx.check(self.lib.ssgpu_block_column(blk, k, C.byref(col)))
                    ptrs[i] = col.data
                self.ctx.synchronize()
            for i, v in enumerate(skip_vectors):
                if v is not None and not isinstance(v, np.ndarray):
                    ptrs[i] = int(v)
            try:
                res = C.c_void_p()
                self.ctx.check(self.lib.ssgpu_expr_evaluate_skip(self.handle, cols, n, rows, ptrs, n_out, C.byref(res)))
                self._result = res
                out = self.fetch(res, nulls_of_result=True)
            finally:
                if blk is not None:
                    self.ctx.synchronize()
                    self.lib.ssgpu_block_destroy(blk)
            for i in host:
                z = out.column(i).is_null
                if z is not None:
                    skip_vectors[i][:] = z
            return ResultView(view=out)
        except SupersonicException as e:
            return ResultView(exception=e)


class ResultView(object):
    """cursor/base/cursor.h:42-122."""

    def __init__(self, view=None, eos=False, exception=None):
        self._view, self._eos, self._exc = view, eos, exception

    def has_data(self):
        return self._view is not None

    def is_eos(self):
        return self._eos

    def is_failure(self):
        return self._exc is not None

    def view(self):
        return self._view

    def exception(self):
        return self._exc


class Cursor(object):
    """cursor/base/cursor.h:131-226.  The pipeline runs on the first Next()."""

    def __init__(self, operation, context):
        self.plan = Plan(operation, context)
        self._result = None
        self._pos = 0
        self._interrupted = False
        self._best_effort = isinstance(operation, BestEffortGroupAggregate)
        self._next_row, self._input_rows = 0, 0

    def schema(self):
        return self.plan.result_schema

    def Interrupt(self):
        self._interrupted = True
        self.plan.interrupt()

    def Next(self, max_row_count=kDefaultRowCount):
        if self._best_effort:
            # GroupAggregateCursor::Next with best_effort_ (aggregate_groups.cc:211-222): serve the current result; when it has been
            # read and the input is not exhausted, ProcessInput again.  A view never mixes rows of two results.
            while self._result is None or self._pos >= self._result.row_count():
                if self._result is not None and self._next_row >= self._input_rows:
                    return ResultView(eos=True)
                try:
                    self._next_row = self.plan.run_best_effort(self._next_row)
                    self._result = self.plan.fetch()
                    self._input_rows = self.plan.input.row_count()
                    self._pos = 0
                except SupersonicException as e:
                    return ResultView(exception=e)
                if self._result.row_count() == 0 and self._next_row >= self._input_rows:
                    return ResultView(eos=True)
        elif self._result is None:
            try:
                self.plan.run()
                self._result = self.plan.fetch()
            except SupersonicException as e:
                return ResultView(exception=e)
        total = self._result.row_count()
        if self._pos >= total:
            return ResultView(eos=True)
        # rowcount_t is an UNSIGNED 64-bit integer in the reference (base/infrastructure/types.h:252-256): Next(-1), the guide's
        # "as many rows as you have" (test/guide/primer.cc:321), is Next(2**64 - 1); 0 still returns at least one row
        want = int(max_row_count) & 0xFFFFFFFFFFFFFFFF
        n = min(max(want, 1), total - self._pos)
        lo, hi = self._pos, self._pos + n
        self._pos = hi
        cols = [Column(c.data[lo:hi], None if c.is_null is None else c.is_null[lo:hi]) for c in self._result._cols]
        return ResultView(view=View(self._result.schema(), cols, n))


def drain(cursor, max_row_count=kDefaultRowCount):
    """Pull a cursor to EOS and concatenate the views (test helper, cf. ViewCopier)."""
    parts = []
    while True:
        r = cursor.Next(max_row_count)
        if r.is_failure():
            raise r.exception()
        if r.is_eos():
            break
        parts.append(r.view())
    schema = cursor.schema()
    cols = []
    for i in range(schema.attribute_count()):
        dt = numpy_dtype(schema.attribute(i).type())
        data = np.concatenate([p.column(i).data for p in parts]) if parts else np.zeros(0, dt)
        nulls = None
        if schema.attribute(i).is_nullable():
            nulls = np.concatenate([p.column(i).is_null for p in parts]) if parts else np.zeros(0, bool)
        cols.append(Column(data, nulls))
    return View(schema, cols, sum(p.row_count() for p in parts))
