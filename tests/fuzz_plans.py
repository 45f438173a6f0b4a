"""Random plan generator for the fuzz parity tests (test_fuzz_cpu.py, test_fuzz_gpu.py).

Seeded and deterministic.  Expressions are drawn from the operator families on the device path with
random types and nullability, so most trees bind and a fraction do not (implicit downcasts,
irreconcilable types ...): the binder must agree with the oracle on BOTH outcomes.  Operators whose
results are not bit-defined across CPUs and GPUs are left out on purpose: quiet division / square root
(they produce NaNs whose sign and payload differ), variable shift counts, SUM over floating point
(order-dependent rounding)."""
import numpy as np

import supersonic_amd as ss

NA = ss.NamedAttribute

COLUMNS = [("a", ss.INT64, True), ("b", ss.INT64, False), ("k1", ss.INT32, True), ("k2", ss.INT32, False),
           ("u", ss.UINT32, False), ("w", ss.UINT64, True), ("d0", ss.DOUBLE, True), ("d1", ss.DOUBLE, False),
           ("f", ss.FLOAT, False), ("t", ss.BOOL, True), ("s", ss.BOOL, False), ("name", ss.STRING, True), ("day", ss.DATE, False)]
WORDS = ["", "a", "ab", "abc", "b", "ba", "zebra", "Zebra", "alpha", "beta", "gamma", "delta", "x y", "x", "xyz"]
INTS = [c for c in COLUMNS if c[1] in (ss.INT64, ss.INT32, ss.UINT32, ss.UINT64)]
FLOATS = [c for c in COLUMNS if c[1] in (ss.DOUBLE, ss.FLOAT)]
BOOLS = [c for c in COLUMNS if c[1] == ss.BOOL]


def make_view(n, seed):
    rng = np.random.default_rng(seed)

    def nulls(flag):
        return (rng.random(n) < 0.15) if flag else None
    data = {
        "a": rng.integers(-1000, 1000, n), "b": rng.integers(-50, 50, n),
        "k1": rng.integers(-100, 100, n).astype(np.int32), "k2": rng.integers(0, 7, n).astype(np.int32),
        "u": rng.integers(0, 1 << 32, n).astype(np.uint32), "w": rng.integers(0, 1 << 63, n).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, n).astype(np.uint64),
        "d0": rng.integers(-4000, 4000, n) * 0.25, "d1": rng.integers(-64, 64, n).astype(np.float64),
        "f": (rng.integers(-64, 64, n) * 0.5).astype(np.float32), "t": rng.integers(0, 2, n).astype(bool), "s": rng.integers(0, 2, n).astype(bool),
        "name": np.array([WORDS[i] for i in rng.integers(0, len(WORDS), n)], dtype=object), "day": rng.integers(-400, 400, n).astype(np.int32)}
    schema = ss.TupleSchema([ss.Attribute(name, t, ss.NULLABLE if nl else ss.NOT_NULLABLE) for (name, t, nl) in COLUMNS])
    return ss.View(schema, [ss.Column(data[name], nulls(nl)) for (name, _t, nl) in COLUMNS])


class Gen(object):
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)

    def pick(self, xs):
        return xs[int(self.rng.integers(0, len(xs)))]

    def const_int(self):
        v = int(self.rng.integers(-20, 20))
        return self.pick([ss.ConstInt32, ss.ConstInt64, ss.ConstInt64])(v)

    def const_float(self):
        return self.pick([ss.ConstDouble, ss.ConstDouble, ss.ConstFloat])(float(self.rng.integers(-16, 16)) * 0.5)

    def integer(self, depth):
        r = self.rng.random()
        if depth <= 0 or r < 0.25:
            return NA(self.pick(INTS)[0]) if self.rng.random() < 0.8 else self.const_int()
        x, y = self.integer(depth - 1), self.integer(depth - 1)
        if self.rng.random() < 0.04:     # ill-typed on purpose: both binders must refuse it with the same code
            bad = int(self.rng.integers(0, 5))
            if bad == 0:
                return ss.Plus(x, self.boolean(0))
            if bad == 1:
                return ss.BitwiseAnd(x, self.floating(0))
            if bad == 2:
                return ss.If(y, x, y)
            if bad == 3:
                return ss.ShiftLeft(x, self.floating(0))
            return ss.ModulusNulling(self.floating(0), x)
        choice = int(self.rng.integers(0, 15))
        if choice < 3:
            return self.pick([ss.Plus, ss.Minus, ss.Multiply])(x, y)
        if choice == 3:
            return ss.Negate(x)
        if choice == 4:
            return self.pick([ss.CppDivideNulling, ss.ModulusNulling])(x, y)
        if choice == 5:
            return self.pick([ss.If, ss.NullingIf])(self.boolean(depth - 1), x, y)
        if choice == 6:
            return ss.IfNull(x, y)
        if choice == 7:
            return ss.CastTo(self.pick([ss.INT64, ss.INT32, ss.UINT32, ss.UINT64]), x)
        if choice == 8:
            return self.pick([ss.BitwiseAnd, ss.BitwiseOr, ss.BitwiseXor, ss.BitwiseAndNot])(x, y)
        if choice == 9:
            return ss.BitwiseNot(x)
        if choice == 10:
            return self.pick([ss.ShiftLeft, ss.ShiftRight])(x, ss.ConstInt32(int(self.rng.integers(0, 9))))
        if choice == 11:
            return self.pick([ss.RoundToInt, ss.CeilToInt, ss.FloorToInt])(self.floating(depth - 1))
        if choice == 12:
            return ss.Abs(x)
        if choice == 13:
            return ss.Case([self.integer(depth - 1), x, self.const_int(), y, self.const_int(), self.integer(0)])
        return x

    def floating(self, depth):
        r = self.rng.random()
        if depth <= 0 or r < 0.25:
            return NA(self.pick(FLOATS)[0]) if self.rng.random() < 0.8 else self.const_float()
        x = self.floating(depth - 1)
        y = self.floating(depth - 1) if self.rng.random() < 0.7 else self.integer(depth - 1)
        choice = int(self.rng.integers(0, 10))
        if choice < 3:
            return self.pick([ss.Plus, ss.Minus, ss.Multiply])(x, y)
        if choice == 3:
            return ss.DivideNulling(x, y)
        if choice == 4:
            return self.pick([ss.Round, ss.Ceil, ss.Floor, ss.Trunc, ss.Abs, ss.Negate])(x)
        if choice == 5:
            return ss.SqrtNulling(x)
        if choice == 6:
            return self.pick([ss.If, ss.NullingIf])(self.boolean(depth - 1), x, y)
        if choice == 7:
            return ss.IfNull(x, y)
        if choice == 8:
            return ss.CastTo(self.pick([ss.DOUBLE, ss.FLOAT]), y)
        return x

    def numeric(self, depth):
        return self.integer(depth) if self.rng.random() < 0.6 else self.floating(depth)

    def boolean(self, depth):
        r = self.rng.random()
        if depth <= 0 or r < 0.15:
            return NA(self.pick(BOOLS)[0])
        choice = int(self.rng.integers(0, 9))
        if choice < 3:
            cmp = self.pick([ss.Less, ss.LessOrEqual, ss.Greater, ss.GreaterOrEqual, ss.Equal, ss.NotEqual])
            return cmp(self.numeric(depth - 1), self.numeric(depth - 1))
        if choice == 3:
            return self.pick([ss.And, ss.Or, ss.Xor, ss.AndNot])(self.boolean(depth - 1), self.boolean(depth - 1))
        if choice == 4:
            return ss.Not(self.boolean(depth - 1))
        if choice == 5:
            return ss.IsNull(self.numeric(depth - 1))
        if choice == 6:
            return self.pick([ss.IsOdd, ss.IsEven])(self.integer(depth - 1))
        if choice == 7:
            r = self.rng.random()
            if r < 0.3:     # STRING comparisons run on order-preserving dictionary codes
                cmp = self.pick([ss.Less, ss.LessOrEqual, ss.Greater, ss.GreaterOrEqual, ss.Equal, ss.NotEqual])
                return cmp(NA("name"), ss.ConstString(self.pick(WORDS + ["aa", "zz"])))
            if r < 0.45:
                return ss.In(NA("name"), [ss.ConstString(self.pick(WORDS + ["nope"])) for _ in range(int(self.rng.integers(1, 4)))])
            if r < 0.6:     # DATE against DATE, and DATETIME through the explicit cast
                if self.rng.random() < 0.5:
                    return self.pick([ss.Less, ss.GreaterOrEqual, ss.Equal])(NA("day"), ss.ConstDate(int(self.rng.integers(-400, 400))))
                return ss.Less(ss.CastTo(ss.DATETIME, NA("day")), ss.ConstDateTime(int(self.rng.integers(-400, 400)) * 86400000000 + 5))
            return ss.In(self.integer(depth - 1), [self.const_int() for _ in range(int(self.rng.integers(1, 5)))])
        return ss.If(self.boolean(depth - 1), self.boolean(depth - 1), self.boolean(depth - 1))

    def any_expr(self, depth):
        r = self.rng.random()
        return self.integer(depth) if r < 0.45 else self.floating(depth) if r < 0.75 else self.boolean(depth)

    # ---- plans ------------------------------------------------------------------------------
    def compute_plan(self, view):
        e = ss.CompoundExpression()
        for i in range(int(self.rng.integers(1, 6))):
            e.AddAs("e%d" % i, self.any_expr(int(self.rng.integers(1, 5))))
        child = ss.ScanView(view)
        if self.rng.random() < 0.5:
            child = ss.Filter(self.boolean(int(self.rng.integers(1, 4))), ss.ProjectAllAttributes(), child)
        return ss.Compute(e, child)

    def aggregate_plan(self, view, grouped):
        e = ss.CompoundExpression().Add(NA("k2")).Add(NA("s")).Add(NA("name")).Add(NA("day")).Add(NA("a")).Add(NA("b")).Add(NA("k1")).Add(NA("w"))
        spec = ss.AggregationSpecification()
        if self.rng.random() < 0.4:
            spec.AddAggregation(self.pick([ss.MIN, ss.MAX, ss.COUNT, ss.FIRST, ss.LAST]), "name", "rname")
        if self.rng.random() < 0.3:
            spec.AddAggregation(self.pick([ss.MIN, ss.MAX, ss.FIRST, ss.LAST]), "day", "rday")
        for i in range(int(self.rng.integers(1, 7))):
            name = "x%d" % i
            kind = int(self.rng.integers(0, 4))
            if kind == 0:
                e.AddAs(name, self.integer(int(self.rng.integers(0, 4))))
                spec.AddAggregation(self.pick([ss.SUM, ss.MIN, ss.MAX, ss.COUNT, ss.FIRST, ss.LAST]), name, "r%d" % i)
            elif kind == 1:
                # MIN / MAX of -0.0 and +0.0 depends on the visiting order in the reference;
                # x + 0.0 turns -0.0 into +0.0 and leaves every other value alone
                e.AddAs(name, ss.Plus(self.floating(int(self.rng.integers(0, 4))), ss.ConstDouble(0.0)))
                spec.AddAggregation(self.pick([ss.MIN, ss.MAX, ss.COUNT]), name, "r%d" % i)
            elif kind == 2:
                e.AddAs(name, self.boolean(int(self.rng.integers(0, 3))))
                spec.AddAggregation(self.pick([ss.MIN, ss.MAX, ss.COUNT, ss.FIRST, ss.LAST]), name, "r%d" % i)
            else:
                e.AddAs(name, self.integer(0))
                spec.AddAggregation(ss.COUNT, "", "r%d" % i)
        child = ss.ScanView(view)
        if self.rng.random() < 0.6:
            child = ss.Filter(self.boolean(int(self.rng.integers(1, 4))), ss.ProjectAllAttributes(), child)
        child = ss.Compute(e, child)
        if grouped:
            # [k1], [a], [k1, k2] ... do not pack into one 64-bit key word: the sort-based fallback
            keys = self.pick([["k2"], ["k2", "s"], ["s"], ["k1"], ["k1", "k2"], ["a"], ["a", "k1", "s"], ["name"], ["day", "s"], ["name", "k2"]])
            if "name" in keys:     # an aggregate's output may not reuse a key's name
                spec.elements = [x for x in spec.elements if x[-1] != "rname"] or spec.elements
            return ss.GroupAggregate(ss.ProjectNamedAttributes(keys), spec, None, child)
        return ss.ScalarAggregate(spec, child)

    def ordered_aggregate_plan(self, view):
        """Aggregates whose device shapes carry an explicit order along (round 4): DISTINCT next to FIRST / LAST, key limits with
        FIRST / LAST and wide keys, DISTINCT inside AggregateClusters.  Returns (operation, result order is defined)."""
        e = ss.CompoundExpression().Add(NA("k2")).Add(NA("s")).Add(NA("k1")).Add(NA("a")).Add(NA("b")).Add(NA("u")).Add(NA("t")).Add(NA("day")).Add(NA("name"))
        shape = self.pick(["scalar", "group", "group", "limit", "limit", "clusters", "clusters"])
        spec = ss.AggregationSpecification()
        inputs = ["a", "b", "k1", "u", "t", "day", "k2"]
        for i in range(int(self.rng.integers(0, 3))):
            e.AddAs("x%d" % i, self.integer(int(self.rng.integers(0, 3))))
            inputs.append("x%d" % i)
        any_distinct = False
        for i in range(int(self.rng.integers(1, 7))):
            name = self.pick(inputs)
            agg = self.pick([ss.SUM, ss.MIN, ss.MAX, ss.COUNT, ss.FIRST, ss.LAST, ss.FIRST, ss.LAST])
            if name in ("t", "day", "name") and agg == ss.SUM:
                agg = ss.COUNT
            distinct = shape != "limit" and agg in (ss.SUM, ss.COUNT, ss.MIN, ss.MAX) and self.rng.random() < 0.45
            any_distinct = any_distinct or distinct
            (spec.AddDistinctAggregation if distinct else spec.AddAggregation)(agg, name, "r%d" % i)
        if shape in ("scalar", "group", "clusters") and not any_distinct:
            spec.AddDistinctAggregation(self.pick([ss.SUM, ss.COUNT]), self.pick(["a", "b", "k1", "u"]), "rd")
        child = ss.ScanView(view)
        if self.rng.random() < 0.5:
            child = ss.Filter(self.boolean(int(self.rng.integers(1, 3))), ss.ProjectAllAttributes(), child)
        child = ss.Compute(e, child)
        if shape == "scalar":
            return ss.ScalarAggregate(spec, child), True
        keys = self.pick([["k2"], ["k2", "s"], ["s"], ["k1"], ["k1", "k2"], ["a", "s"], ["day", "s"], ["name", "k2"]])
        if shape == "clusters":
            # (runs of equal keys are short in random rows; ["s"] and ["k2"] give runs of 2 .. 7 rows here and there)
            return ss.AggregateClusters(ss.ProjectNamedAttributes(self.pick([["s"], ["k2"], ["k2", "s"], ["t"]])), spec, child), True
        if shape == "limit":
            return ss.GroupAggregate(ss.ProjectNamedAttributes(keys), spec,
                                     ss.GroupAggregateOptions().set_max_unique_keys_in_result_(int(self.rng.integers(0, 9))), child), True
        return ss.GroupAggregate(ss.ProjectNamedAttributes(keys), spec, None, child), False

    def distinct_limit_plan(self, view):
        """DISTINCT aggregates under GroupAggregateOptions::max_unique_keys_in_result (round 5): the seen-value sets belong to the RESULT
        rows, so the rows folded into the last one share a set.  Keys of every width, FIRST / LAST and plain aggregates next to
        them, limits around and beyond the number of groups.  Returns (operation, True): first-seen order is defined."""
        e = ss.CompoundExpression().Add(NA("k2")).Add(NA("s")).Add(NA("k1")).Add(NA("a")).Add(NA("b")).Add(NA("u")).Add(NA("t")).Add(NA("day")).Add(NA("name")).Add(NA("d0"))
        spec = ss.AggregationSpecification()
        inputs = ["a", "b", "k1", "u", "t", "day", "k2", "s"]
        for i in range(int(self.rng.integers(0, 3))):
            e.AddAs("x%d" % i, self.integer(int(self.rng.integers(0, 3))))
            inputs.append("x%d" % i)
        n_distinct = 0
        concat = self.rng.random() < 0.3      # CONCAT under the limit as well, alone or next to DISTINCT aggregates
        for i in range(int(self.rng.integers(1, 6))):
            name = self.pick(inputs)
            agg = self.pick([ss.SUM, ss.MIN, ss.MAX, ss.COUNT, ss.COUNT, ss.FIRST, ss.LAST])
            if name in ("t", "day") and agg == ss.SUM:
                agg = ss.COUNT
            distinct = agg in (ss.SUM, ss.COUNT, ss.MIN, ss.MAX) and self.rng.random() < (0.3 if concat else 0.6)
            n_distinct += distinct
            (spec.AddDistinctAggregation if distinct else spec.AddAggregation)(agg, name, "r%d" % i)
        if concat or self.rng.random() < 0.15:     # a row-after-row SUM (floating input, integer result) rides the (result row, row id) order as well
            spec.AddAggregationWithDefinedOutputType(ss.SUM, "d0", "rq", self.pick([ss.INT64, ss.INT32]))
        if concat:
            for j in range(int(self.rng.integers(1, 3))):
                (spec.AddDistinctAggregation if self.rng.random() < 0.4 else spec.AddAggregation)(ss.CONCAT, self.pick(["a", "b", "k1", "u", "s", "k2", "name", "day"]), "rc%d" % j)
        elif n_distinct == 0:
            spec.AddDistinctAggregation(self.pick([ss.SUM, ss.COUNT]), self.pick(["a", "b", "k1", "u", "s"]), "rd")
        child = ss.ScanView(view)
        if self.rng.random() < 0.5:
            child = ss.Filter(self.boolean(int(self.rng.integers(1, 3))), ss.ProjectAllAttributes(), child)
        child = ss.Compute(e, child)
        keys = self.pick([["k2"], ["k2", "s"], ["s"], ["k1"], ["k1", "k2"], ["a", "s"], ["day", "s"], ["name", "k2"], ["t"]])
        limit = int(self.pick([0, 1, 2, 3, 5, 8, 40, 300, 5000, 100000]))
        return ss.GroupAggregate(ss.ProjectNamedAttributes(keys), spec, ss.GroupAggregateOptions().set_max_unique_keys_in_result_(limit), child), True

    def sequential_sum_plan(self, view):
        """SUM of FLOAT / DOUBLE inputs into integer results (folded row after row in input order) next to ordinary aggregates,
        as ScalarAggregate / GroupAggregate / AggregateClusters.  Returns (operation, result order is defined)."""
        e = ss.CompoundExpression().Add(NA("k2")).Add(NA("s")).Add(NA("k1")).Add(NA("a")).Add(NA("b")).Add(NA("d0")).Add(NA("d1")).Add(NA("f")).Add(NA("t"))
        inputs = ["d0", "d1", "f"]
        if self.rng.random() < 0.5:
            e.AddAs("y", ss.Plus(self.floating(int(self.rng.integers(0, 3))), ss.ConstDouble(0.0)))
            inputs.append("y")
        spec = ss.AggregationSpecification()
        for i in range(int(self.rng.integers(1, 4))):
            spec.AddAggregationWithDefinedOutputType(ss.SUM, self.pick(inputs), "q%d" % i, self.pick([ss.INT64, ss.INT32, ss.INT64]))
        for i in range(int(self.rng.integers(0, 4))):
            spec.AddAggregation(self.pick([ss.SUM, ss.MIN, ss.MAX, ss.COUNT, ss.FIRST, ss.LAST]), self.pick(["a", "b", "k1", "t"]), "r%d" % i)
        for i in range(int(self.pick([0, 0, 1, 2]))):     # DISTINCT aggregates next to them: the rows are sorted back into input order
            spec.AddDistinctAggregation(self.pick([ss.SUM, ss.COUNT]), self.pick(["a", "b", "k1"]), "rd%d" % i)
        child = ss.ScanView(view)
        if self.rng.random() < 0.5:
            child = ss.Filter(self.boolean(int(self.rng.integers(1, 3))), ss.ProjectAllAttributes(), child)
        child = ss.Compute(e, child)
        shape = self.pick(["scalar", "group", "group", "clusters"])
        if shape == "scalar":
            return ss.ScalarAggregate(spec, child), True
        if shape == "clusters":
            return ss.AggregateClusters(ss.ProjectNamedAttributes(self.pick([["s"], ["k2"], ["k2", "s"], ["t"]])), spec, child), True
        return ss.GroupAggregate(ss.ProjectNamedAttributes(self.pick([["k2"], ["k2", "s"], ["k1"], ["k1", "k2"], ["a", "s"]])), spec, None, child), False

    def sort_plan(self, view):
        e = ss.CompoundExpression().Add(NA("b")).Add(NA("k1")).Add(NA("d0")).Add(NA("t")).Add(NA("u")).Add(NA("name")).Add(NA("day"))
        for i in range(int(self.rng.integers(0, 3))):
            e.AddAs("e%d" % i, self.any_expr(int(self.rng.integers(1, 4))))
        child = ss.ScanView(view)
        if self.rng.random() < 0.5:
            child = ss.Filter(self.boolean(int(self.rng.integers(1, 3))), ss.ProjectAllAttributes(), child)
        names = ["b", "k1", "d0", "t", "u", "name", "day"]
        order = ss.SortOrder()
        for k in self.rng.permutation(len(names))[: int(self.rng.integers(1, 4))]:
            order.add(names[int(k)], self.pick([ss.ASCENDING, ss.DESCENDING]))
        return ss.Sort(order, None, 0, ss.Compute(e, child))

    def join_plan(self, view):
        # star join against a 7-row dimension keyed by k2 (one key missing for LEFT_OUTER / INNER to differ)
        dim_schema = ss.TupleSchema([ss.Attribute("id", ss.INT32), ss.Attribute("weight", ss.INT64, ss.NULLABLE), ss.Attribute("rate", ss.DOUBLE)])
        unique = self.rng.random() < 0.5
        ids = np.array([0, 1, 2, 3, 5, 6] if unique else [0, 1, 5, 1, 3, 5, 5, 2], dtype=np.int32)   # NOT_UNIQUE: keys repeat, rows multiply
        seq = np.arange(len(ids), dtype=np.int64)
        dim = ss.View(dim_schema, [ids, ss.Column(seq * 100 - 250, seq == 3), seq * 0.25])
        joined = ss.HashJoin(self.pick([ss.INNER, ss.LEFT_OUTER]), ss.ProjectNamedAttribute("k2"), ss.ProjectNamedAttribute("id"),
                             ss.CompoundMultiSourceProjector().add(0, ss.ProjectAllAttributes()).add(1, ss.ProjectNamedAttributes(["weight", "rate"])),
                             ss.UNIQUE if unique else ss.NOT_UNIQUE, ss.ScanView(view), ss.ScanView(dim))
        e = ss.CompoundExpression().AddAs("j0", ss.Plus(NA("weight"), self.integer(int(self.rng.integers(0, 3))))) \
            .AddAs("j1", ss.Multiply(NA("rate"), self.floating(int(self.rng.integers(0, 3))))).AddAs("j2", self.any_expr(2))
        if self.rng.random() < 0.5:
            joined = ss.Filter(self.boolean(int(self.rng.integers(1, 3))), ss.ProjectAllAttributes(), joined)
        return ss.Compute(e, joined)

    def plan(self, view):
        r = self.rng.random()
        if r < 0.4:
            return self.compute_plan(view), True
        if r < 0.55:
            return self.aggregate_plan(view, False), True
        if r < 0.75:
            return self.aggregate_plan(view, True), False      # group order is unspecified
        if r < 0.9:
            return self.sort_plan(view), True
        return self.join_plan(view), True


# ---- random PLAIN GroupAggregates (tests/test_dense_gpu.py, tests/fuzz_worker.py "plain_group"): keys and aggregate inputs are
# ---- input columns, Filters are `column CMP constant` -- the stages that take the dense-slot shapes
def random_plain_group(seed, view):
    rng = np.random.default_rng(90000 + seed)

    def pick(xs):
        return xs[int(rng.integers(0, len(xs)))]
    keys = pick([["k2"], ["k2", "s"], ["s"], ["k1"], ["k1", "k2"], ["t", "k2"], ["day"], ["day", "s"], ["name"], ["name", "k2"], ["s", "day"], ["t"], ["k1", "s", "t"]])
    spec = ss.AggregationSpecification()
    inputs = [c for c in ("a", "b", "k1", "k2", "u", "w", "d1", "f", "t", "s", "day", "name", "d0") if c not in keys]
    for i in range(int(rng.integers(1, 7))):
        col = pick(inputs)
        aggs = [ss.MIN, ss.MAX, ss.COUNT]
        if col in ("a", "b", "k1", "k2", "u", "w", "d1", "d0"):
            aggs.append(ss.SUM)            # (d0 / d1 hold multiples of 0.25: every partial sum is exact)
        # (FIRST / LAST order by a row id the stage computes: not a plain stage -- the hashed tests cover them)
        spec.AddAggregation(pick(aggs), col, "r%d" % i)
    if rng.random() < 0.5:
        spec.AddAggregation(ss.COUNT, "", "rows")
    child = ss.ScanView(view)
    for _ in range(int(rng.integers(0, 3))):
        col, const = pick([("b", ss.ConstInt64(int(rng.integers(-60, 60)))), ("k2", ss.ConstInt32(int(rng.integers(0, 7)))), ("a", ss.ConstInt64(int(rng.integers(-1000, 1000)))),
                           ("d1", ss.ConstDouble(float(rng.integers(-64, 64)))), ("u", ss.ConstUint32(int(rng.integers(0, 1 << 32))))])
        cmp = pick([ss.Less, ss.LessOrEqual, ss.Greater, ss.GreaterOrEqual, ss.Equal, ss.NotEqual])
        child = ss.Filter(cmp(ss.NamedAttribute(col), const) if rng.random() < 0.7 else cmp(const, ss.NamedAttribute(col)), ss.ProjectAllAttributes(), child)
    return ss.GroupAggregate(ss.ProjectNamedAttributes(keys), spec, None, child)
