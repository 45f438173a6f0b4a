// User code in the style of the reference's guide (test/guide/primer.cc, group_sort.cc, join.cc), written against the
// reference's OWN include path and idioms -- `#include "supersonic/supersonic.h"`, `using supersonic::...`, unqualified
// int32 / int64, typed_data<INT32>(), SucceedOrDie(op->CreateCursor()), cursor->Next(-1), View copies,
// column(i).attribute().name(), TableRowWriter -- and built with nothing but -I<repo>/include -lssgpu.  What it shows: a
// user of the reference can point the same source at the MI355X library.  The scenarios are this repository's own (they
// mirror the SHAPE of the guide's examples, not their text): a bound a + b, a grouped SUM, a two-key sort, a Table fed
// row by row, and the Cursor / Operation interface methods.
//   guide_test bind   (no GPU: everything up to CreateCursor / Bind)       guide_test run   (GPU: results checked)
#include <stdio.h>
#include <string.h>

#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "supersonic/supersonic.h"

using supersonic::Attribute;
using supersonic::AggregationSpecification;
using supersonic::BoundExpressionTree;
using supersonic::CompoundSingleSourceProjector;
using supersonic::Cursor;
using supersonic::EvaluationResult;
using supersonic::Expression;
using supersonic::FailureOrOwned;
using supersonic::HeapBufferAllocator;
using supersonic::MemoryLimit;
using supersonic::Operation;
using supersonic::ResultView;
using supersonic::SingleSourceProjector;
using supersonic::SortOrder;
using supersonic::StringPiece;
using supersonic::SucceedOrDie;
using supersonic::Table;
using supersonic::TableRowWriter;
using supersonic::TupleSchema;
using supersonic::View;
using supersonic::rowcount_t;

using supersonic::INT32;
using supersonic::INT64;
using supersonic::DOUBLE;
using supersonic::STRING;
using supersonic::UINT64;
using supersonic::NOT_NULLABLE;
using supersonic::NULLABLE;
using supersonic::SUM;
using supersonic::COUNT;
using supersonic::MAX;
using supersonic::ASCENDING;
using supersonic::DESCENDING;

static int g_fail = 0;
static bool g_run = false;
#define EXPECT_TRUE(c) do { if (!(c)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)
#define EXPECT_EQ(a, b) do { if (!((a) == (b))) { printf("FAIL %s:%d: %s == %s\n", __FILE__, __LINE__, #a, #b); ++g_fail; } } while (0)

// ---- scenario 1: an expression bound once, evaluated over a View -------------------------------------------------------
static BoundExpressionTree* BindSum() {
  std::unique_ptr<const Expression> addition(supersonic::Plus(supersonic::AttributeAt(0), supersonic::AttributeAt(1)));
  TupleSchema schema;
  schema.add_attribute(Attribute("a", INT32, NOT_NULLABLE));
  schema.add_attribute(Attribute("b", INT32, NOT_NULLABLE));
  FailureOrOwned<BoundExpressionTree> bound = addition->Bind(schema, HeapBufferAllocator::Get(), 2048);
  EXPECT_TRUE(bound.is_success());
  if (bound.is_failure()) { printf("  %s\n", bound.exception().message().c_str()); return NULL; }
  EXPECT_EQ(bound->result_schema().attribute(0).name(), std::string("(a + b)"));
  EXPECT_EQ(bound->result_schema().attribute(0).type(), INT32);
  return bound.release();
}

static void ScenarioBoundExpression() {
  int32 a[8] = {10, -1, 2, 7, 40, 5, 0, 2147483647};
  int32 b[8] = {5, 1, 2, -7, 2, 5, 0, 1};
  std::unique_ptr<BoundExpressionTree> tree(BindSum());
  if (!tree.get() || !g_run) return;
  TupleSchema schema;
  schema.add_attribute(Attribute("a", INT32, NOT_NULLABLE));
  schema.add_attribute(Attribute("b", INT32, NOT_NULLABLE));
  View input(schema);
  input.set_row_count(8);
  input.mutable_column(0)->Reset(a, NULL);
  input.mutable_column(1)->Reset(b, NULL);
  EvaluationResult result = tree->Evaluate(input);
  EXPECT_TRUE(result.is_success());
  if (result.is_failure()) { printf("  %s\n", result.exception().message().c_str()); return; }
  EXPECT_EQ(1, result.get().column_count());
  EXPECT_EQ(8u, result.get().row_count());
  const int32* sum = result.get().column(0).typed_data<INT32>();
  for (int i = 0; i < 8; ++i) EXPECT_EQ(sum[i], static_cast<int32>(static_cast<uint32>(a[i]) + static_cast<uint32>(b[i])));   // (wraps like the reference's int32 +)
  // more rows than the bound capacity: ERROR_TOO_MANY_ROWS, as expression.cc:57-66
  std::vector<int32> big(4096, 1);
  View too_many(schema);
  too_many.set_row_count(4096);
  too_many.mutable_column(0)->Reset(big.data(), NULL);
  too_many.mutable_column(1)->Reset(big.data(), NULL);
  EvaluationResult refused = tree->Evaluate(too_many);
  EXPECT_TRUE(refused.is_failure());
  if (refused.is_failure()) EXPECT_EQ(refused.exception().return_code(), supersonic::ERROR_TOO_MANY_ROWS);
}

// ---- scenario 2: a grouped SUM drained with Next(-1) ---------------------------------------------------------------------
static Cursor* GroupedTotals(int32* keys, double* values, size_t rows, View* input_view /* must outlive the cursor */) {
  input_view->set_row_count(rows);
  input_view->mutable_column(0)->Reset(keys, NULL);
  input_view->mutable_column(1)->Reset(values, NULL);
  std::unique_ptr<AggregationSpecification> specification(new AggregationSpecification());
  specification->AddAggregation(SUM, "amount", "total");
  specification->AddAggregation(COUNT, "", "rows");
  std::unique_ptr<const SingleSourceProjector> key_projector(supersonic::ProjectNamedAttribute("shop"));
  // the operation must outlive its cursors (operation.h:59): a function-local static keeps it for this test program
  static std::vector<std::unique_ptr<Operation> > keep;
  keep.emplace_back(supersonic::GroupAggregate(key_projector.release(), specification.release(), NULL, supersonic::ScanView(*input_view)));
  Operation* aggregation = keep.back().get();
  EXPECT_EQ(aggregation->DebugDescription(), std::string("GroupAggregate(ScanView())"));
  return SucceedOrDie(aggregation->CreateCursor());
}

static void ScenarioGroupedSum() {
  const unsigned size = 10;
  int32 shop[size] = {7, 3, 7, 9, 3, 3, 7, 9, 9, 9};
  double amount[size] = {1.5, 2.25, 4.0, 8.5, 0.25, 16.0, 32.5, 0.5, 0.125, 64.0};
  std::map<int32, std::pair<double, uint64> > expected;
  for (unsigned i = 0; i < size; ++i) { expected[shop[i]].first += amount[i]; expected[shop[i]].second += 1; }

  TupleSchema schema;
  schema.add_attribute(Attribute("shop", INT32, NOT_NULLABLE));
  schema.add_attribute(Attribute("amount", DOUBLE, NOT_NULLABLE));
  View input_view(schema);
  std::unique_ptr<Cursor> cursor(GroupedTotals(shop, amount, size, &input_view));
  EXPECT_EQ(3, cursor->column_count());
  EXPECT_EQ(cursor->schema().attribute(1).name(), std::string("total"));
  EXPECT_EQ(cursor->GetCursorId(), supersonic::GROUP_AGGREGATE);
  if (!g_run) return;

  ResultView result(cursor->Next(-1));        // "as many rows as you have": rowcount_t is unsigned
  EXPECT_TRUE(result.has_data());
  EXPECT_TRUE(!result.is_eos());
  if (!result.has_data()) { if (result.is_failure()) printf("  %s\n", result.exception().message().c_str()); return; }
  View result_view(result.view());            // a View is a copyable reference to the rows
  EXPECT_EQ(3, result_view.column_count());
  EXPECT_EQ(expected.size(), result_view.row_count());
  EXPECT_EQ(std::string("shop"), result_view.column(0).attribute().name());
  EXPECT_EQ(std::string("total"), result_view.column(1).attribute().name());
  const int32* keys = result_view.column(0).typed_data<INT32>();
  const double* totals = result_view.column(1).typed_data<DOUBLE>();
  const uint64* counts = result_view.column(2).typed_data<UINT64>();
  for (rowcount_t i = 0; i < result_view.row_count(); ++i) {
    EXPECT_EQ(expected[keys[i]].first, totals[i]);
    EXPECT_EQ(expected[keys[i]].second, counts[i]);
  }
  EXPECT_TRUE(cursor->Next(-1).is_eos());
}

// ---- scenario 3: ORDER BY two keys, pulled in small views -----------------------------------------------------------------
static void ScenarioTwoKeySort() {
  const int n = 5000;
  std::vector<int32> region(n);
  std::vector<int64> stamp(n);
  std::vector<double> payload(n);
  uint64 seed = 12345;
  for (int i = 0; i < n; ++i) {
    seed = seed * 6364136223846793005ull + 1442695040888963407ull;
    region[i] = static_cast<int32>((seed >> 33) % 17);
    stamp[i] = static_cast<int64>((seed >> 11) % 100000) - 50000;
    payload[i] = i * 0.5;
  }
  TupleSchema schema;
  schema.add_attribute(Attribute("region", INT32, NOT_NULLABLE));
  schema.add_attribute(Attribute("stamp", INT64, NOT_NULLABLE));
  schema.add_attribute(Attribute("payload", DOUBLE, NOT_NULLABLE));
  View input(schema);
  input.set_row_count(n);
  input.mutable_column(0)->Reset(region.data(), NULL);
  input.mutable_column(1)->Reset(stamp.data(), NULL);
  input.mutable_column(2)->Reset(payload.data(), NULL);

  std::unique_ptr<SortOrder> order(new SortOrder());
  order->add(supersonic::ProjectNamedAttribute("region"), ASCENDING);
  order->add(supersonic::ProjectNamedAttribute("stamp"), DESCENDING);
  std::unique_ptr<Operation> sort(supersonic::Sort(order.release(), supersonic::ProjectAllAttributes(), /*memory_limit=*/1 << 30, supersonic::ScanView(input)));
  std::unique_ptr<Cursor> cursor(SucceedOrDie(sort->CreateCursor()));
  EXPECT_EQ(cursor->GetCursorId(), supersonic::SORT);
  if (!g_run) return;

  rowcount_t seen = 0;
  int32 last_region = -1;
  int64 last_stamp = 0;
  double payload_sum = 0;
  for (;;) {
    ResultView rv(cursor->Next(Cursor::kDefaultRowCount));
    if (rv.is_done()) { EXPECT_TRUE(rv.is_eos()); break; }
    const View& v = rv.view();
    EXPECT_TRUE(v.row_count() >= 1 && v.row_count() <= Cursor::kDefaultRowCount);
    for (rowcount_t i = 0; i < v.row_count(); ++i) {
      const int32 r = v.column(0).typed_data<INT32>()[i];
      const int64 s = v.column(1).typed_data<INT64>()[i];
      EXPECT_TRUE(r > last_region || (r == last_region && s <= last_stamp));
      last_region = r; last_stamp = s;
      payload_sum += v.column(2).typed_data<DOUBLE>()[i];
    }
    seen += v.row_count();
  }
  EXPECT_EQ(seen, static_cast<rowcount_t>(n));
  EXPECT_EQ(payload_sum, 0.5 * (static_cast<double>(n) * (n - 1) / 2));
}

// ---- scenario 4: a Table filled row by row is an Operation like any other ------------------------------------------------
static void ScenarioTableRowWriter() {
  TupleSchema schema;
  schema.add_attribute(Attribute("author", STRING, NOT_NULLABLE));
  schema.add_attribute(Attribute("year", INT32, NULLABLE));
  schema.add_attribute(Attribute("copies", INT64, NOT_NULLABLE));
  std::unique_ptr<Table> books(new Table(schema, HeapBufferAllocator::Get()));
  TableRowWriter writer(books.get());
  writer.AddRow().String("Lem").Int32(1961).Int64(120)
        .AddRow().String("Lem").Int32(1964).Int64(80)
        .AddRow().String("Capek").Null().Int64(40)
        .AddRow().String("Lem").Null().Int64(1)
        .AddRow().String("Capek").Int32(1936).Int64(60)
        .CheckSuccess();
  for (int i = 0; i < 100; ++i) writer.AddRow().String("Filler").Int32(2000 + i).Int64(1);   // grows past the first capacity
  writer.CheckSuccess();
  EXPECT_EQ(105u, books->row_count());
  EXPECT_EQ(std::string("Capek"), books->view().column(0).typed_data<STRING>()[2].ToString());
  EXPECT_TRUE(books->view().column(1).is_null()[2]);
  EXPECT_TRUE(!books->view().column(1).is_null()[4]);

  std::unique_ptr<AggregationSpecification> spec(new AggregationSpecification());
  spec->AddAggregation(SUM, "copies", "copies");
  spec->AddAggregation(MAX, "year", "latest");
  Table* table = books.get();
  std::unique_ptr<Operation> per_author(supersonic::GroupAggregate(supersonic::ProjectNamedAttribute("author"), spec.release(), NULL, books.release()));
  (void)table;
  // the allocator seam of the interface: a quota for the operation's buffers, set where none is set yet
  MemoryLimit limit(1 << 30);
  per_author->SetBufferAllocatorWhereUnset(&limit, /*cascade_to_children=*/true);
  EXPECT_EQ(per_author->DebugDescription(), std::string("GroupAggregate(Table())"));
  std::unique_ptr<Cursor> cursor(SucceedOrDie(per_author->CreateCursor()));
  if (!g_run) return;
  ResultView rv(cursor->Next(-1));
  EXPECT_TRUE(rv.has_data());
  if (!rv.has_data()) { if (rv.is_failure()) printf("  %s\n", rv.exception().message().c_str()); return; }
  const View& v = SucceedOrDie(rv);
  EXPECT_EQ(3u, v.row_count());
  for (rowcount_t i = 0; i < v.row_count(); ++i) {
    const std::string who = v.column(0).typed_data<STRING>()[i].ToString();
    const int64 copies = v.column(1).typed_data<INT64>()[i];
    const int32 latest = v.column(2).typed_data<INT32>()[i];
    if (who == "Lem") { EXPECT_EQ(201, copies); EXPECT_EQ(1964, latest); }
    else if (who == "Capek") { EXPECT_EQ(100, copies); EXPECT_EQ(1936, latest); }
    else { EXPECT_EQ(std::string("Filler"), who); EXPECT_EQ(100, copies); EXPECT_EQ(2099, latest); }
  }
}

// ---- scenario 5: failures are values -----------------------------------------------------------------------------------------
static void ScenarioFailures() {
  TupleSchema schema;
  schema.add_attribute(Attribute("a", INT32, NOT_NULLABLE));
  View input(schema);
  std::unique_ptr<Operation> bad(supersonic::Compute(supersonic::NamedAttribute("no_such_column"), supersonic::ScanView(input)));
  FailureOrOwned<Cursor> result = bad->CreateCursor();
  EXPECT_TRUE(result.is_failure());
  if (result.is_failure()) {
    EXPECT_EQ(result.exception().return_code(), supersonic::ERROR_ATTRIBUTE_MISSING);
    EXPECT_TRUE(!result.exception().PrintStackTrace().empty());
  }
}

int main(int argc, char** argv) {
  g_run = argc > 1 && !strcmp(argv[1], "run");
  ScenarioBoundExpression();
  ScenarioGroupedSum();
  ScenarioTwoKeySort();
  ScenarioTableRowWriter();
  ScenarioFailures();
  printf(g_fail ? "FAILED (%d)\n" : "PASSED\n", g_fail);
  return g_fail ? 1 : 0;
}
