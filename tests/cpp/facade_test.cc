// Exercises include/supersonic_amd/supersonic.h the way the reference's own guide/tests use
// supersonic/supersonic.h (test/guide/*.cc, cursor/core/*_test.cc): build an Operation tree
// with the public factories, CreateCursor(), pull with Next().
//
//   facade_test bind   -- binding, naming and bind-error checks only (no GPU needed)
//   facade_test run    -- also executes the pipelines on the GPU and checks the values
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <memory>
#include <vector>

#include "supersonic_amd/supersonic.h"

using namespace supersonic;  // NOLINT

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)
#define CHECK_EQ(a, b) do { auto va = (a); auto vb = (b); if (!(va == vb)) { printf("FAIL %s:%d: %s == %s\n", __FILE__, __LINE__, #a, #b); ++g_fail; } } while (0)

struct Input {
  static const int N = 5000;
  std::vector<int64_t> a, b;
  std::vector<int32_t> k;
  std::vector<double> d;
  std::vector<char> d_null;
  TupleSchema schema;
  std::unique_ptr<View> view;
  Input() : a(N), b(N), k(N), d(N), d_null(N) {
    uint64_t s = 12345;
    for (int i = 0; i < N; ++i) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      a[i] = static_cast<int64_t>((s >> 33) % 1000) - 300;
      b[i] = static_cast<int64_t>((s >> 20) % 77);
      k[i] = static_cast<int32_t>((s >> 45) % 7);
      d[i] = static_cast<double>((s >> 11) % 64) * 0.25;
      d_null[i] = ((s >> 5) % 9) == 0;
    }
    schema.add_attribute(Attribute("a", INT64, NOT_NULLABLE));
    schema.add_attribute(Attribute("b", INT64, NOT_NULLABLE));
    schema.add_attribute(Attribute("k", INT32, NOT_NULLABLE));
    schema.add_attribute(Attribute("d", DOUBLE, NULLABLE));
    view.reset(new View(schema));
    view->mutable_column(0)->Reset(a.data(), nullptr);
    view->mutable_column(1)->Reset(b.data(), nullptr);
    view->mutable_column(2)->Reset(k.data(), nullptr);
    view->mutable_column(3)->Reset(d.data(), reinterpret_cast<const bool*>(d_null.data()));
    view->set_row_count(N);
  }
};

// Filter -> Compute -> ScalarAggregate, the Q-FPA shape.
static Operation* Fpa(const Input& in) {
  return ScalarAggregate(
      (new AggregationSpecification)->AddAggregation(SUM, "s", "sum_s")->AddAggregation(COUNT, "", "cnt")->AddAggregation(MIN, "d", "min_d"),
      Compute((new CompoundExpression)->AddAs("s", Plus(NamedAttribute("a"), Multiply(NamedAttribute("b"), ConstInt64(3))))->Add(NamedAttribute("d")),
              Filter(Greater(NamedAttribute("a"), ConstInt64(0)), ProjectAllAttributes(), ScanView(*in.view))));
}

static void TestBind(const Input& in) {
  {
    std::unique_ptr<Operation> op(Fpa(in));
    FailureOrOwned<Cursor> c = op->CreateCursor();
    CHECK(c.is_success());
    if (c.is_success()) {
      CHECK_EQ(c->schema().attribute_count(), 3);
      CHECK_EQ(c->schema().attribute(0).name(), std::string("sum_s"));
      CHECK_EQ(c->schema().attribute(0).type(), INT64);
      CHECK_EQ(c->schema().attribute(1).type(), UINT64);
      CHECK(!c->schema().attribute(1).is_nullable());
      CHECK_EQ(c->schema().attribute(2).type(), DOUBLE);
    }
  }
  {  // result naming of expressions (expression tests: "(a + b)", CAST_..)
    std::unique_ptr<Operation> op(Compute((new CompoundExpression)->Add(Plus(NamedAttribute("a"), NamedAttribute("k")))->Add(Less(NamedAttribute("d"), NamedAttribute("a"))), ScanView(*in.view)));
    FailureOrOwned<Cursor> c = op->CreateCursor();
    CHECK(c.is_success());
    if (c.is_success()) {
      CHECK_EQ(c->schema().attribute(0).name(), std::string("(a + CAST_INT32_TO_INT64(k))"));
      CHECK_EQ(c->schema().attribute(1).type(), BOOL);
      CHECK(c->schema().attribute(1).is_nullable());
    }
  }
  {  // missing attribute -> ERROR_ATTRIBUTE_MISSING at CreateCursor
    std::unique_ptr<Operation> op(Filter(Less(NamedAttribute("nope"), ConstInt64(1)), ProjectAllAttributes(), ScanView(*in.view)));
    FailureOrOwned<Cursor> c = op->CreateCursor();
    CHECK(c.is_failure());
    if (c.is_failure()) CHECK_EQ(c.exception().return_code(), ERROR_ATTRIBUTE_MISSING);
  }
  {  // non-BOOL predicate
    std::unique_ptr<Operation> op(Filter(NamedAttribute("a"), ProjectAllAttributes(), ScanView(*in.view)));
    FailureOrOwned<Cursor> c = op->CreateCursor();
    CHECK(c.is_failure());
  }
  {  // duplicate output name
    std::unique_ptr<Operation> op(ScalarAggregate((new AggregationSpecification)->AddAggregation(SUM, "a", "x")->AddAggregation(MIN, "b", "x"), ScanView(*in.view)));
    FailureOrOwned<Cursor> c = op->CreateCursor();
    CHECK(c.is_failure());
    if (c.is_failure()) CHECK_EQ(c.exception().return_code(), ERROR_ATTRIBUTE_EXISTS);
  }
}

static void TestRun(const Input& in) {
  {
    std::unique_ptr<Operation> op(Fpa(in));
    FailureOrOwned<Cursor> c = op->CreateCursor();
    CHECK(c.is_success());
    ResultView r = c->Next(Cursor::kDefaultRowCount);
    if (r.is_failure()) { printf("run failed: %s\n", r.exception().message().c_str()); ++g_fail; return; }
    CHECK(r.has_data());
    int64_t sum = 0; uint64_t cnt = 0; double mn = INFINITY; bool any = false;
    for (int i = 0; i < Input::N; ++i) if (in.a[i] > 0) {
      sum += in.a[i] + in.b[i] * 3; ++cnt;
      if (!in.d_null[i]) { any = true; mn = fmin(mn, in.d[i]); }
    }
    CHECK_EQ(r.view().row_count(), 1u);
    CHECK_EQ(r.view().column(0).typed_data<int64_t>()[0], sum);
    CHECK_EQ(r.view().column(1).typed_data<uint64_t>()[0], cnt);
    CHECK(any && r.view().column(2).typed_data<double>()[0] == mn);
    CHECK(c->Next(Cursor::kDefaultRowCount).is_eos());
  }
  {  // Filter materialisation pulled in 1024-row views, order preserved
    std::unique_ptr<Operation> op(Filter(Less(NamedAttribute("d"), ConstDouble(4.0)), ProjectNamedAttributes({"a", "d"}), ScanView(*in.view)));
    FailureOrOwned<Cursor> c = op->CreateCursor();
    CHECK(c.is_success());
    int i = 0; rowcount_t total = 0;
    for (;;) {
      ResultView r = c->Next(1024);
      if (r.is_failure()) { printf("run failed: %s\n", r.exception().message().c_str()); ++g_fail; break; }
      if (r.is_eos()) break;
      CHECK(r.view().row_count() <= 1024);
      for (rowcount_t j = 0; j < r.view().row_count(); ++j) {
        while (i < Input::N && (in.d_null[i] || !(in.d[i] < 4.0))) ++i;
        CHECK(i < Input::N);
        if (i >= Input::N) break;
        if (r.view().column(0).typed_data<int64_t>()[j] != in.a[i] || r.view().column(1).typed_data<double>()[j] != in.d[i]) { CHECK(false); break; }
        ++i;
      }
      total += r.view().row_count();
    }
    rowcount_t expect = 0;
    for (int j = 0; j < Input::N; ++j) expect += (!in.d_null[j] && in.d[j] < 4.0);
    CHECK_EQ(total, expect);
  }
  {  // GroupAggregate then Sort by key
    std::unique_ptr<Operation> op(Sort((new SortOrder)->add(ProjectNamedAttribute("k"), ASCENDING), nullptr, 1 << 20,
        GroupAggregate(ProjectNamedAttribute("k"), (new AggregationSpecification)->AddAggregation(SUM, "a", "sa")->AddAggregation(COUNT, "d", "cd"), nullptr, ScanView(*in.view))));
    FailureOrOwned<Cursor> c = op->CreateCursor();
    CHECK(c.is_success());
    ResultView r = c->Next(Cursor::kDefaultRowCount);
    if (r.is_failure()) { printf("run failed: %s\n", r.exception().message().c_str()); ++g_fail; return; }
    int64_t sa[7] = {0}; uint64_t cd[7] = {0};
    for (int i = 0; i < Input::N; ++i) { sa[in.k[i]] += in.a[i]; cd[in.k[i]] += !in.d_null[i]; }
    CHECK_EQ(r.view().row_count(), 7u);
    for (int g = 0; g < 7 && static_cast<supersonic::rowcount_t>(g) < r.view().row_count(); ++g) {
      CHECK_EQ(r.view().column(0).typed_data<int32_t>()[g], g);
      CHECK_EQ(r.view().column(1).typed_data<int64_t>()[g], sa[g]);
      CHECK_EQ(r.view().column(2).typed_data<uint64_t>()[g], cd[g]);
    }
  }
  {  // HashJoin(LEFT_OUTER) against a small dimension table, aggregated in the same pipeline
    std::vector<int32_t> id = {0, 1, 2, 3, 5};
    std::vector<int64_t> w = {100, 10, 20, 30, 50};
    TupleSchema ds;
    ds.add_attribute(Attribute("id", INT32, NOT_NULLABLE));
    ds.add_attribute(Attribute("w", INT64, NOT_NULLABLE));
    View dim(ds);
    dim.mutable_column(0)->Reset(id.data(), nullptr);
    dim.mutable_column(1)->Reset(w.data(), nullptr);
    dim.set_row_count(5);
    std::unique_ptr<Operation> op(ScalarAggregate(
        (new AggregationSpecification)->AddAggregation(SUM, "w", "sw")->AddAggregation(COUNT, "w", "cw")->AddAggregation(COUNT, "", "n"),
        HashJoin(LEFT_OUTER, ProjectNamedAttribute("k"), ProjectNamedAttribute("id"),
                 (new CompoundMultiSourceProjector)->add(0, ProjectNamedAttribute("a"))->add(1, ProjectNamedAttribute("w")), UNIQUE,
                 ScanView(*in.view), ScanView(dim))));
    FailureOrOwned<Cursor> c = op->CreateCursor();
    CHECK(c.is_success());
    if (c.is_success()) {
      ResultView r = c->Next(Cursor::kDefaultRowCount);
      if (r.is_failure()) { printf("join failed: %s\n", r.exception().message().c_str()); ++g_fail; }
      else {
        const int64_t wt[7] = {100, 10, 20, 30, 0, 50, 0};
        int64_t sw = 0; uint64_t cw = 0;
        for (int i = 0; i < Input::N; ++i) { sw += wt[in.k[i]]; cw += (in.k[i] != 4 && in.k[i] != 6); }
        CHECK_EQ(r.view().column(0).typed_data<int64_t>()[0], sw);
        CHECK_EQ(r.view().column(1).typed_data<uint64_t>()[0], cw);
        CHECK_EQ(r.view().column(2).typed_data<uint64_t>()[0], static_cast<uint64_t>(Input::N));
      }
    }
  }
  {  // HashJoin(INNER, NOT_UNIQUE): rhs keys repeat, rows multiply (lhs order, matches in rhs order)
    std::vector<int32_t> id = {1, 3, 1, 5, 1};
    std::vector<int64_t> w = {10, 30, 11, 50, 12};
    TupleSchema ds;
    ds.add_attribute(Attribute("id", INT32, NOT_NULLABLE));
    ds.add_attribute(Attribute("w", INT64, NOT_NULLABLE));
    View dim(ds);
    dim.mutable_column(0)->Reset(id.data(), nullptr);
    dim.mutable_column(1)->Reset(w.data(), nullptr);
    dim.set_row_count(5);
    std::unique_ptr<Operation> op(
        HashJoin(INNER, ProjectNamedAttribute("k"), ProjectNamedAttribute("id"),
                 (new CompoundMultiSourceProjector)->add(0, ProjectNamedAttribute("a"))->add(1, ProjectNamedAttribute("w")), NOT_UNIQUE,
                 ScanView(*in.view), ScanView(dim)));
    FailureOrOwned<Cursor> c = op->CreateCursor();
    CHECK(c.is_success());
    if (c.is_success()) {
      std::vector<int64_t> want_a, want_w;
      for (int i = 0; i < Input::N; ++i)
        for (int j = 0; j < 5; ++j) if (id[j] == in.k[i]) { want_a.push_back(in.a[i]); want_w.push_back(w[j]); }
      size_t at = 0; bool ok = true;
      for (;;) {
        ResultView r = c->Next(1024);
        if (r.is_failure()) { printf("multi join failed: %s\n", r.exception().message().c_str()); ++g_fail; break; }
        if (r.is_eos()) break;
        for (rowcount_t j = 0; j < r.view().row_count() && ok; ++j, ++at)
          ok = at < want_a.size() && r.view().column(0).typed_data<int64_t>()[j] == want_a[at] && r.view().column(1).typed_data<int64_t>()[j] == want_w[at];
      }
      CHECK(ok);
      CHECK_EQ(at, want_a.size());
    }
  }
  {  // BestEffortGroupAggregate: the reference's own vector (aggregate_groups_test.cc:601-626): "at 20 bytes quota, the buffer is filled
     // after processing 3 rows" -- views {1: 7, 3: -3} and {2: 4, 3: -5}, each key-unique, no ERROR_MEMORY_EXCEEDED
    std::vector<int32_t> key = {1, 1, 3, 2, 3}, val = {3, 4, -3, 4, -5};
    TupleSchema ts;
    ts.add_attribute(Attribute("col0", INT32, NULLABLE));
    ts.add_attribute(Attribute("col1", INT32, NULLABLE));
    View rows(ts);
    rows.mutable_column(0)->Reset(key.data(), nullptr);
    rows.mutable_column(1)->Reset(val.data(), nullptr);
    rows.set_row_count(5);
    std::unique_ptr<Operation> op(BestEffortGroupAggregate(ProjectNamedAttribute("col0"), (new AggregationSpecification)->AddAggregation(SUM, "col1", "sum"),
                                                           (new GroupAggregateOptions)->set_memory_quota(20)->set_estimated_result_row_count(2), ScanView(rows)));
    FailureOrOwned<Cursor> c = op->CreateCursor();
    CHECK(c.is_success());
    if (c.is_success()) {
      CHECK_EQ(c->GetCursorId(), BEST_EFFORT_GROUP_AGGREGATE);
      const int32_t want[2][2][2] = {{{1, 7}, {3, -3}}, {{2, 4}, {3, -5}}};
      for (int v = 0; v < 2; ++v) {
        ResultView r = c->Next(Cursor::kDefaultRowCount);
        CHECK(r.has_data());
        if (!r.has_data()) break;
        CHECK_EQ(r.view().row_count(), static_cast<rowcount_t>(2));
        for (int j = 0; j < 2 && r.view().row_count() == 2; ++j) {
          CHECK_EQ(r.view().column(0).typed_data<int32_t>()[j], want[v][j][0]);
          CHECK_EQ(r.view().column(1).typed_data<int32_t>()[j], want[v][j][1]);
        }
      }
      CHECK(c->Next(Cursor::kDefaultRowCount).is_eos());
    }
  }
  {  // signaling division by zero surfaces as ERROR_EVALUATION_ERROR from Next()
    std::unique_ptr<Operation> op(Compute(DivideSignaling(NamedAttribute("a"), Minus(NamedAttribute("b"), NamedAttribute("b"))), ScanView(*in.view)));
    FailureOrOwned<Cursor> c = op->CreateCursor();
    CHECK(c.is_success());
    ResultView r = c->Next(Cursor::kDefaultRowCount);
    CHECK(r.is_failure());
    if (r.is_failure()) CHECK_EQ(r.exception().return_code(), ERROR_EVALUATION_ERROR);
  }
}


// The seams around the hot path: BufferAllocator / MemoryLimit (base/memory/memory.h, memory_test.cc's quota cases),
// Expression::Bind (expression.h:158-160) and the STRING dictionary at bind time.  No GPU needed.
static void TestSeamsBind(const Input& in) {
  {
    MemoryLimit limit(1000);
    std::unique_ptr<Buffer> a(limit.Allocate(600));
    CHECK(a != nullptr);
    if (a) { CHECK_EQ(a->size(), static_cast<size_t>(600)); memset(a->data(), 7, 600); }
    CHECK_EQ(limit.Available(), static_cast<size_t>(400));
    std::unique_ptr<Buffer> over(limit.Allocate(500));
    CHECK(over == nullptr);                                   // NULL Buffer: what callers turn into ERROR_MEMORY_EXCEEDED
    std::unique_ptr<Buffer> best(limit.BestEffortAllocate(500, 100));
    CHECK(best != nullptr);
    if (best) CHECK_EQ(best->size(), static_cast<size_t>(400));
    CHECK(!limit.Reallocate(700, a.get()));                   // does not fit; the buffer is left as it was
    CHECK_EQ(a->size(), static_cast<size_t>(600));
    CHECK_EQ(static_cast<const char*>(a->data())[599], 7);
    best.reset();
    // the new size has to fit NEXT TO the old buffer (the reference's mediator is conservative, memory_test.cc:183-198)
    CHECK(!limit.Reallocate(900, a.get()));
    CHECK(limit.Reallocate(400, a.get()));
    CHECK_EQ(a->size(), static_cast<size_t>(400));
    CHECK_EQ(static_cast<const char*>(a->data())[399], 7);
    CHECK_EQ(limit.GetUsage(), static_cast<size_t>(400));
    a.reset();
    CHECK_EQ(limit.GetUsage(), static_cast<size_t>(0));
    std::unique_ptr<Buffer> zero(limit.Allocate(0));
    CHECK(zero != nullptr && zero->data() != nullptr);         // memory.h:112-117
    std::unique_ptr<Buffer> heap(HeapBufferAllocator::Get()->Allocate(1 << 20));
    CHECK(heap != nullptr);
  }
  {
    std::unique_ptr<const Expression> e(Plus(NamedAttribute("a"), NamedAttribute("d")));
    FailureOrOwned<BoundExpressionTree> t = e->Bind(in.schema, HeapBufferAllocator::Get(), 100);
    CHECK(t.is_success());
    e.reset();                                                  // the bound tree does not need the Expression any more
    if (t.is_success()) {
      CHECK_EQ(t->result_schema().attribute_count(), 1);
      CHECK_EQ(t->result_schema().attribute(0).type(), DOUBLE);
      CHECK(t->result_schema().attribute(0).is_nullable());
      CHECK_EQ(t->row_capacity(), static_cast<rowcount_t>(100));
    }
    std::unique_ptr<const Expression> bad(Plus(NamedAttribute("a"), NamedAttribute("zzz")));
    FailureOrOwned<BoundExpressionTree> f = bad->Bind(in.schema, HeapBufferAllocator::Get(), 100);
    CHECK(f.is_failure());
    if (f.is_failure()) CHECK_EQ(f.exception().return_code(), ERROR_ATTRIBUTE_MISSING);
  }
  {  // STRING: constants and columns bind through the dictionary
    TupleSchema s; s.add_attribute(Attribute("name", STRING, NULLABLE)); s.add_attribute(Attribute("v", INT32, NOT_NULLABLE));
    StringPiece cells[3] = {"pear", "apple", "fig"}; int32_t v[3] = {1, 2, 3};
    View view(s); view.mutable_column(0)->Reset(cells, nullptr); view.mutable_column(1)->Reset(v, nullptr); view.set_row_count(3);
    std::unique_ptr<Operation> op(Filter(Less(NamedAttribute("name"), ConstString("grape")), ProjectAllAttributes(), ScanView(view)));
    FailureOrOwned<Cursor> c = op->CreateCursor();
    CHECK(c.is_success());
    if (c.is_success()) CHECK_EQ(c->schema().attribute(0).type(), STRING);
    std::unique_ptr<Operation> mism(Filter(Less(NamedAttribute("name"), ConstInt32(1)), ProjectAllAttributes(), ScanView(view)));
    CHECK(mism->CreateCursor().is_failure());
    CHECK(StringPiece("ab") < StringPiece("abc"));
    CHECK(StringPiece("b") != StringPiece("a"));
  }
}

static void TestSeamsRun(const Input& in) {
  {  // Bind once, Evaluate View after View (what ComputeCursor::Next does with its bound tree)
    std::unique_ptr<const Expression> e((new CompoundExpression)->AddAs("s", Plus(NamedAttribute("a"), Multiply(NamedAttribute("b"), ConstInt64(3))))
                                            ->AddAs("h", DivideNulling(NamedAttribute("d"), CastTo(DOUBLE, NamedAttribute("k")))));
    FailureOrOwned<BoundExpressionTree> t = e->Bind(in.schema, HeapBufferAllocator::Get(), 2048);
    CHECK(t.is_success());
    if (t.is_failure()) return;
    for (rowcount_t first = 0; first < Input::N; first += 2048) {
      const rowcount_t n = std::min<rowcount_t>(2048, Input::N - first);
      View slice(in.schema);
      slice.mutable_column(0)->Reset(in.a.data() + first, nullptr);
      slice.mutable_column(1)->Reset(in.b.data() + first, nullptr);
      slice.mutable_column(2)->Reset(in.k.data() + first, nullptr);
      slice.mutable_column(3)->Reset(in.d.data() + first, reinterpret_cast<const bool*>(in.d_null.data()) + first);
      slice.set_row_count(n);
      EvaluationResult r = t->Evaluate(slice);
      if (r.is_failure()) { printf("evaluate failed: %s\n", r.exception().message().c_str()); ++g_fail; return; }
      CHECK_EQ(r.get().row_count(), n);
      bool ok = true;
      for (rowcount_t i = 0; i < n && ok; ++i) {
        const rowcount_t g = first + i;
        ok = r.get().column(0).typed_data<int64_t>()[i] == in.a[g] + in.b[g] * 3;
        const bool want_null = in.d_null[g] || in.k[g] == 0;
        ok = ok && r.get().column(1).is_null()[i] == want_null;
        if (!want_null) ok = ok && r.get().column(1).typed_data<double>()[i] == in.d[g] / in.k[g];
      }
      CHECK(ok);
    }
    {  // the node-level form: skip vectors in and out (expression.h:46-92)
      const rowcount_t n = 2000;
      View slice(in.schema);
      slice.mutable_column(0)->Reset(in.a.data(), nullptr); slice.mutable_column(1)->Reset(in.b.data(), nullptr);
      slice.mutable_column(2)->Reset(in.k.data(), nullptr); slice.mutable_column(3)->Reset(in.d.data(), reinterpret_cast<const bool*>(in.d_null.data()));
      slice.set_row_count(n);
      std::vector<char> skip0(n), skip1(n);
      for (rowcount_t i = 0; i < n; ++i) { skip0[i] = (i % 3) == 0; skip1[i] = (i % 5) == 0; }
      BoolView skips(2);
      skips.ResetColumn(0, reinterpret_cast<bool*>(skip0.data())); skips.ResetColumn(1, reinterpret_cast<bool*>(skip1.data()));
      skips.set_row_count(n);
      EvaluationResult r = t->DoEvaluate(slice, skips);
      CHECK(r.is_success());
      if (r.is_success()) {
        bool ok = r.get().row_count() == n;
        for (rowcount_t i = 0; i < n && ok; ++i) {
          const bool s0 = (i % 3) == 0, want1 = (i % 5) == 0 || in.d_null[i] || in.k[i] == 0;
          ok = r.get().column(0).is_null()[i] == s0 && (s0 || r.get().column(0).typed_data<int64_t>()[i] == in.a[i] + in.b[i] * 3);
          ok = ok && r.get().column(1).is_null()[i] == want1 && (want1 || r.get().column(1).typed_data<double>()[i] == in.d[i] / in.k[i]);
          ok = ok && (skip0[i] != 0) == s0 && (skip1[i] != 0) == want1;      // the vectors on return: the result's NULLs
        }
        CHECK(ok);
      }
    }
    EvaluationResult many = t->Evaluate(*in.view);             // 5000 rows > capacity 2048
    CHECK(many.is_failure());
    if (many.is_failure()) CHECK_EQ(many.exception().return_code(), ERROR_TOO_MANY_ROWS);
  }
  {  // STRING end to end: filter on a constant, group by a STRING key, MIN/MAX of strings
    TupleSchema s; s.add_attribute(Attribute("name", STRING, NULLABLE)); s.add_attribute(Attribute("v", INT64, NOT_NULLABLE));
    const char* words[6] = {"pear", "apple", "fig", "apple", "pear", "kiwi"};
    std::vector<StringPiece> cells; std::vector<int64_t> v; std::vector<char> z;
    for (int i = 0; i < 600; ++i) { cells.push_back(words[i % 6]); v.push_back(i); z.push_back(i % 50 == 49); }
    View view(s); view.mutable_column(0)->Reset(cells.data(), reinterpret_cast<const bool*>(z.data())); view.mutable_column(1)->Reset(v.data(), nullptr); view.set_row_count(600);
    std::unique_ptr<Operation> op(Sort((new SortOrder)->add(ProjectNamedAttribute("name"), ASCENDING), nullptr, 0,
        GroupAggregate(ProjectNamedAttribute("name"), (new AggregationSpecification)->AddAggregation(SUM, "v", "sv")->AddAggregation(COUNT, "", "n"), nullptr,
            Filter(NotEqual(NamedAttribute("name"), ConstString("fig")), ProjectAllAttributes(), ScanView(view)))));
    FailureOrOwned<Cursor> c = op->CreateCursor();
    CHECK(c.is_success());
    if (c.is_success()) {
      ResultView r = c->Next(Cursor::kDefaultRowCount);
      if (r.is_failure()) { printf("string run failed: %s\n", r.exception().message().c_str()); ++g_fail; return; }
      // NULL names fail the predicate (NULL != 'fig' is NULL); groups in StringPiece order
      const char* want[3] = {"apple", "kiwi", "pear"};
      CHECK_EQ(r.view().row_count(), static_cast<rowcount_t>(3));
      for (int gi = 0; gi < 3 && r.view().row_count() == 3; ++gi) {
        int64_t sum = 0; uint64_t cnt = 0;
        for (int i = 0; i < 600; ++i) if (!z[i] && !strcmp(words[i % 6], want[gi])) { sum += v[i]; ++cnt; }
        CHECK(r.view().column(0).typed_data<StringPiece>()[gi] == StringPiece(want[gi]));
        CHECK_EQ(r.view().column(1).typed_data<int64_t>()[gi], sum);
        CHECK_EQ(r.view().column(2).typed_data<uint64_t>()[gi], cnt);
      }
    }
    {  // CONCAT (column_aggregator_test.cc:365-412): printed values joined with ',' in input order; StringPiece cells of the result's own dictionary
      TupleSchema cs; cs.add_attribute(Attribute("g", INT32, NOT_NULLABLE)); cs.add_attribute(Attribute("w", STRING, NULLABLE)); cs.add_attribute(Attribute("i", INT32, NOT_NULLABLE));
      const char* ws[5] = {"baba", "baba", "dada", "aba", "wada"};
      std::vector<int32_t> g = {0, 1, 0, 1, 0}, iv = {-5, 0, 345, 2, -2};
      std::vector<StringPiece> wc; for (int k = 0; k < 5; ++k) wc.push_back(ws[k]);
      std::vector<char> wz = {0, 0, 0, 1, 0};
      View cv(cs); cv.mutable_column(0)->Reset(g.data(), nullptr); cv.mutable_column(1)->Reset(wc.data(), reinterpret_cast<const bool*>(wz.data()));
      cv.mutable_column(2)->Reset(iv.data(), nullptr); cv.set_row_count(5);
      std::unique_ptr<Operation> cop(GroupAggregate(ProjectNamedAttribute("g"),
          (new AggregationSpecification)->AddAggregation(CONCAT, "w", "cw")->AddAggregation(CONCAT, "i", "ci")->AddAggregation(SUM, "i", "si"), nullptr, ScanView(cv)));
      FailureOrOwned<Cursor> cc = cop->CreateCursor();
      CHECK(cc.is_success());
      if (cc.is_success()) {
        CHECK_EQ(cc->schema().attribute(1).type(), STRING);
        ResultView r = cc->Next(Cursor::kDefaultRowCount);
        if (r.is_failure()) { printf("CONCAT run failed: %s\n", r.exception().message().c_str()); ++g_fail; }
        else {
          CHECK_EQ(r.view().row_count(), static_cast<rowcount_t>(2));
          for (rowcount_t row = 0; row < r.view().row_count() && r.view().row_count() == 2; ++row) {
            const bool g0 = r.view().column(0).typed_data<int32_t>()[row] == 0;
            CHECK(r.view().column(1).typed_data<StringPiece>()[row] == StringPiece(g0 ? "baba,dada,wada" : "baba"));
            CHECK(r.view().column(2).typed_data<StringPiece>()[row] == StringPiece(g0 ? "-5,345,-2" : "0,2"));
            CHECK_EQ(r.view().column(3).typed_data<int32_t>()[row], g0 ? 338 : 2);
          }
        }
      }
    }
    std::unique_ptr<Operation> mm(ScalarAggregate((new AggregationSpecification)->AddAggregation(MIN, "name", "lo")->AddAggregation(MAX, "name", "hi"), ScanView(view)));
    FailureOrOwned<Cursor> c2 = mm->CreateCursor();
    CHECK(c2.is_success());
    if (c2.is_success()) {
      ResultView r = c2->Next(Cursor::kDefaultRowCount);
      CHECK(r.has_data());
      if (r.has_data()) {
        CHECK(r.view().column(0).typed_data<StringPiece>()[0] == StringPiece("apple"));
        CHECK(r.view().column(1).typed_data<StringPiece>()[0] == StringPiece("pear"));
      }
    }
    // and through a bound expression: the dictionary is rebuilt per evaluated View
    std::unique_ptr<const Expression> e(If(Less(NamedAttribute("name"), ConstString("g")), NamedAttribute("name"), ConstString("other")));
    FailureOrOwned<BoundExpressionTree> t = e->Bind(s, HeapBufferAllocator::Get(), 1024);
    CHECK(t.is_success());
    if (t.is_success()) {
      EvaluationResult r = t->Evaluate(view);
      CHECK(r.is_success());
      bool ok = r.is_success() && r.get().row_count() == 600;
      for (int i = 0; ok && i < 600; ++i) {
        if (z[i]) ok = r.get().column(0).typed_data<StringPiece>()[i] == StringPiece("other") || r.get().column(0).is_null() != nullptr;
        else ok = r.get().column(0).typed_data<StringPiece>()[i] == (strcmp(words[i % 6], "g") < 0 ? StringPiece(words[i % 6]) : StringPiece("other"));
      }
      CHECK(ok);
    }
  }
  {  // SetBufferAllocator(MemoryLimit): running past the quota is ERROR_MEMORY_EXCEEDED from Next() (aggregate_groups.cc:372-402)
    MemoryLimit tiny(4096), roomy(static_cast<size_t>(1) << 30);
    for (int pass = 0; pass < 2; ++pass) {
      std::unique_ptr<Operation> op(GroupAggregate(ProjectNamedAttribute("k"), (new AggregationSpecification)->AddAggregation(SUM, "a", "sa"), nullptr, ScanView(*in.view)));
      op->SetBufferAllocator(pass == 0 ? static_cast<BufferAllocator*>(&tiny) : &roomy, true);
      FailureOrOwned<Cursor> c = op->CreateCursor();
      CHECK(c.is_success());
      ResultView r = c->Next(Cursor::kDefaultRowCount);
      if (pass == 0) { CHECK(r.is_failure()); if (r.is_failure()) CHECK_EQ(r.exception().return_code(), ERROR_MEMORY_EXCEEDED); }
      else { CHECK(r.has_data()); if (r.has_data()) CHECK_EQ(r.view().row_count(), static_cast<rowcount_t>(7)); }
    }
  }
}

// The View file format through the mirror: FileOutput's byte image (file_io.cc:15-30,122-147), then (GPU) FileInput ->
// ScanDeviceView -> query -> WriteResultToFile.
static std::string ReadAll(const std::string& path) {
  std::string out; FILE* f = fopen(path.c_str(), "rb"); if (!f) return out;
  char buf[4096]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
  fclose(f); return out;
}
static void TestFileFormat(bool run) {
  char dir_template[] = "/tmp/ssgpu_facade_XXXXXX";
  const std::string dir = mkdtemp(dir_template);
  {  // a STRING column: lengths (0 for NULL and empty), then one run of bytes
    TupleSchema schema;
    schema.add_attribute(Attribute("s", STRING, NULLABLE));
    schema.add_attribute(Attribute("k", INT32, NOT_NULLABLE));
    const StringPiece cells[4] = {StringPiece("ab", 2), StringPiece("ignored", 7), StringPiece("", 0), StringPiece("xyz", 3)};
    const bool nulls[4] = {false, true, false, false};
    const int32_t k[4] = {1, 2, 3, 4};
    View v(schema);
    v.mutable_column(0)->Reset(cells, nulls); v.mutable_column(1)->Reset(k, nullptr); v.set_row_count(4);
    std::unique_ptr<Sink> sink(FileOutput(dir + "/s.ssv"));
    FailureOr<rowcount_t> w = sink->Write(v);
    CHECK(w.is_success()); if (w.is_success()) CHECK_EQ(w.get(), static_cast<rowcount_t>(4));
    CHECK(sink->Finalize().is_success());
    std::string want;
    const uint64_t rc = 4, lens[4] = {2, 0, 0, 3};
    want.append(reinterpret_cast<const char*>(&rc), 8);
    want.append(reinterpret_cast<const char*>(nulls), 4);
    want.append(reinterpret_cast<const char*>(lens), 32);
    want.append("abxyz");
    want.append(reinterpret_cast<const char*>(k), 16);
    CHECK(ReadAll(dir + "/s.ssv") == want);
  }
  const rowcount_t n = 20000;   // three chunks
  TupleSchema schema;
  schema.add_attribute(Attribute("k", INT32, NULLABLE));
  schema.add_attribute(Attribute("v", INT64, NOT_NULLABLE));
  std::vector<int32_t> k(n); std::vector<int64_t> val(n); std::vector<char> kn(n);
  for (rowcount_t i = 0; i < n; ++i) { k[i] = static_cast<int32_t>(i % 5); val[i] = static_cast<int64_t>(i); kn[i] = (i % 11) == 0; }
  View v(schema);
  v.mutable_column(0)->Reset(k.data(), reinterpret_cast<const bool*>(kn.data())); v.mutable_column(1)->Reset(val.data(), nullptr); v.set_row_count(n);
  {
    std::unique_ptr<Sink> sink(FileOutput(dir + "/t.ssv"));
    CHECK(sink->Write(v).is_success());
    CHECK(sink->Finalize().is_success());
    const std::string raw = ReadAll(dir + "/t.ssv");
    CHECK_EQ(raw.size(), static_cast<size_t>(3 * 8 + n * (1 + 4 + 8)));
    uint64_t first = 0; memcpy(&first, raw.data(), 8); CHECK_EQ(first, static_cast<uint64_t>(8192));
  }
  if (!run) return;
  FailureOrOwned<DeviceTable> table = FileInput(schema, dir + "/t.ssv");
  CHECK(table.is_success());
  if (!table.is_success()) return;
  CHECK_EQ(table->row_count(), n);
  std::unique_ptr<Operation> op(GroupAggregate(ProjectNamedAttribute("k"), (new AggregationSpecification)->AddAggregation(SUM, "v", "sv"), nullptr,
                                               ScanDeviceView(table->view())));
  FailureOrOwned<Cursor> c = op->CreateCursor();
  CHECK(c.is_success());
  CHECK(WriteResultToFile(c.get(), dir + "/out.ssv").is_success());
  FailureOrOwned<DeviceTable> back = FileInput(c->schema(), dir + "/out.ssv", true);
  CHECK(back.is_success());
  if (back.is_success()) CHECK_EQ(back->row_count(), static_cast<rowcount_t>(6));   // keys 0..4 and NULL
  int64_t total = 0; rowcount_t rows = 0;
  for (;;) {
    ResultView r = c->Next(Cursor::kDefaultRowCount);
    if (!r.has_data()) { CHECK(r.is_eos()); break; }
    for (rowcount_t i = 0; i < r.view().row_count(); ++i) total += r.view().column(1).typed_data<int64_t>()[i];
    rows += r.view().row_count();
  }
  CHECK_EQ(rows, static_cast<rowcount_t>(6));
  CHECK_EQ(total, static_cast<int64_t>(n) * static_cast<int64_t>(n - 1) / 2);
  CHECK(FileInput(schema, dir + "/missing.ssv").is_failure());
}

// Chunked staging (SetHostStagingChunkRows): a ScalarAggregate over a host View larger than the chunk is fed to the device chunk by
// chunk -- the copy of one overlapping the kernel over the one before -- and gives the row the whole-block path gives
static void TestHostStaging() {
  const rowcount_t n = 100003;
  TupleSchema schema;
  schema.add_attribute(Attribute("a", INT64, NOT_NULLABLE));
  schema.add_attribute(Attribute("x", DOUBLE, NULLABLE));
  std::vector<int64_t> a(n); std::vector<double> x(n); std::vector<char> xn(n);
  for (rowcount_t i = 0; i < n; ++i) { a[i] = static_cast<int64_t>((i * 7919) % 1000); x[i] = 0.25 * static_cast<double>(i % 4001) - 300.0; xn[i] = (i % 13) == 0; }
  View v(schema);
  v.mutable_column(0)->Reset(a.data(), nullptr); v.mutable_column(1)->Reset(x.data(), reinterpret_cast<const bool*>(xn.data())); v.set_row_count(n);
  auto make = [&]() {
    return ScalarAggregate((new AggregationSpecification)->AddAggregation(SUM, "a", "sa")->AddAggregation(COUNT, "x", "cx")->AddAggregation(MIN, "x", "mn")
                               ->AddAggregation(SUM, "x", "sx")->AddAggregation(LAST, "x", "lx")->AddAggregation(FIRST, "a", "fa"),
                           Filter(Greater(NamedAttribute("a"), ConstInt64(499)), ProjectAllAttributes(), ScanView(v)));
  };
  int64_t sa[2] = {0, 0}, fa[2] = {0, 0}; uint64_t cx[2] = {0, 0}; double mn[2] = {0, 0}, sx[2] = {0, 0}, lx[2] = {0, 0};
  for (int pass = 0; pass < 2; ++pass) {
    SetHostStagingChunkRows(pass == 0 ? 0 : 4096);     // whole block, then 25 chunks
    std::unique_ptr<Operation> op(make());
    FailureOrOwned<Cursor> c = op->CreateCursor();
    CHECK(c.is_success());
    if (!c.is_success()) continue;
    ResultView r = c->Next(Cursor::kDefaultRowCount);
    CHECK(r.has_data());
    if (!r.has_data()) continue;
    CHECK_EQ(r.view().row_count(), static_cast<rowcount_t>(1));
    sa[pass] = r.view().column(0).typed_data<int64_t>()[0]; cx[pass] = r.view().column(1).typed_data<uint64_t>()[0];
    mn[pass] = r.view().column(2).typed_data<double>()[0]; sx[pass] = r.view().column(3).typed_data<double>()[0];
    lx[pass] = r.view().column(4).typed_data<double>()[0]; fa[pass] = r.view().column(5).typed_data<int64_t>()[0];
  }
  SetHostStagingChunkRows(0);
  CHECK_EQ(sa[0], sa[1]); CHECK_EQ(cx[0], cx[1]); CHECK(mn[0] == mn[1]); CHECK(sx[0] == sx[1]); CHECK(lx[0] == lx[1]); CHECK_EQ(fa[0], fa[1]);
  int64_t want_sa = 0; for (rowcount_t i = 0; i < n; ++i) if (a[i] > 499) want_sa += a[i];
  CHECK_EQ(sa[1], want_sa);
}

// The other chunked forms through the facade (ssgpu.h "CHUNKED STAGING" 2 and 3): a materialising Filter appends every chunk's rows, a
// GroupAggregate leaves a partial table per chunk and is merged once -- both must give what the whole-block path gives
static void TestHostStagingOtherForms() {
  const rowcount_t n = 120007;
  TupleSchema schema;
  schema.add_attribute(Attribute("k", INT32, NOT_NULLABLE));
  schema.add_attribute(Attribute("a", INT64, NOT_NULLABLE));
  schema.add_attribute(Attribute("x", DOUBLE, NULLABLE));
  std::vector<int32_t> k(n); std::vector<int64_t> a(n); std::vector<double> x(n); std::vector<char> xn(n);
  for (rowcount_t i = 0; i < n; ++i) { k[i] = static_cast<int32_t>((i * 31) % 977); a[i] = static_cast<int64_t>((i * 7919) % 1000); x[i] = 0.25 * static_cast<double>(i % 4001) - 300.0; xn[i] = (i % 11) == 0; }
  View v(schema);
  v.mutable_column(0)->Reset(k.data(), nullptr); v.mutable_column(1)->Reset(a.data(), nullptr);
  v.mutable_column(2)->Reset(x.data(), reinterpret_cast<const bool*>(xn.data())); v.set_row_count(n);
  // form 3: per key SUM(a), COUNT(x), SUM(x), MAX(x) -- summed over the keys (the result's row order is unspecified)
  int64_t sum_a[2] = {0, 0}; uint64_t cnt_x[2] = {0, 0}; double sum_x[2] = {0, 0}, max_x[2] = {0, 0}; rowcount_t groups[2] = {0, 0};
  // form 2: the rows of Filter(a > 499), in input order -- a running checksum that depends on the order
  uint64_t checksum[2] = {0, 0}; rowcount_t kept[2] = {0, 0};
  for (int pass = 0; pass < 2; ++pass) {
    SetHostStagingChunkRows(pass == 0 ? 0 : 5000);
    {
      std::unique_ptr<Operation> op(GroupAggregate(ProjectNamedAttribute("k"),
                                                   (new AggregationSpecification)->AddAggregation(SUM, "a", "sa")->AddAggregation(COUNT, "x", "cx")
                                                       ->AddAggregation(SUM, "x", "sx")->AddAggregation(MAX, "x", "mx"),
                                                   nullptr, Filter(Greater(NamedAttribute("a"), ConstInt64(99)), ProjectAllAttributes(), ScanView(v))));
      FailureOrOwned<Cursor> c = op->CreateCursor();
      CHECK(c.is_success());
      if (c.is_success())
        for (;;) {
          ResultView r = c->Next(Cursor::kDefaultRowCount);
          if (!r.has_data()) { CHECK(r.is_eos()); break; }
          for (rowcount_t i = 0; i < r.view().row_count(); ++i) {
            sum_a[pass] += r.view().column(1).typed_data<int64_t>()[i]; cnt_x[pass] += r.view().column(2).typed_data<uint64_t>()[i];
            if (!r.view().column(3).is_null() || !r.view().column(3).is_null()[i]) sum_x[pass] += r.view().column(3).typed_data<double>()[i];
            if (!r.view().column(4).is_null() || !r.view().column(4).is_null()[i]) max_x[pass] = std::max(max_x[pass], r.view().column(4).typed_data<double>()[i]);
          }
          groups[pass] += r.view().row_count();
        }
    }
    {
      std::unique_ptr<Operation> op(Filter(Greater(NamedAttribute("a"), ConstInt64(499)), ProjectAllAttributes(), ScanView(v)));
      FailureOrOwned<Cursor> c = op->CreateCursor();
      CHECK(c.is_success());
      if (c.is_success())
        for (;;) {
          ResultView r = c->Next(Cursor::kDefaultRowCount);
          if (!r.has_data()) { CHECK(r.is_eos()); break; }
          for (rowcount_t i = 0; i < r.view().row_count(); ++i)
            checksum[pass] = checksum[pass] * 1000003ull + static_cast<uint64_t>(r.view().column(0).typed_data<int32_t>()[i]) * 31ull + static_cast<uint64_t>(r.view().column(1).typed_data<int64_t>()[i]);
          kept[pass] += r.view().row_count();
        }
    }
  }
  SetHostStagingChunkRows(0);
  CHECK_EQ(groups[0], static_cast<rowcount_t>(977)); CHECK_EQ(groups[1], groups[0]);
  CHECK_EQ(sum_a[0], sum_a[1]); CHECK_EQ(cnt_x[0], cnt_x[1]); CHECK(sum_x[0] == sum_x[1]); CHECK(max_x[0] == max_x[1]);
  CHECK_EQ(kept[0], kept[1]); CHECK_EQ(checksum[0], checksum[1]);
  int64_t want = 0; rowcount_t want_kept = 0;
  for (rowcount_t i = 0; i < n; ++i) { if (a[i] > 99) want += a[i]; if (a[i] > 499) ++want_kept; }
  CHECK_EQ(sum_a[1], want); CHECK_EQ(kept[1], want_kept);
}

int main(int argc, char** argv) {
  const bool run = argc > 1 && !strcmp(argv[1], "run");
  Input in;
  TestBind(in);
  TestSeamsBind(in);
  TestFileFormat(run);
  if (run) { TestRun(in); TestSeamsRun(in); TestHostStaging(); TestHostStagingOtherForms(); }
  printf(g_fail ? "FAILED (%d)\n" : "PASSED\n", g_fail);
  return g_fail ? 1 : 0;
}
