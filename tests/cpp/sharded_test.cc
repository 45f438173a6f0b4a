// A C++ host driving the sharded GroupAggregate (BASELINE config #4) through include/supersonic_amd/sharded.h: RCCL called
// directly (ncclAllGather of the packed partial tables, or the key-range all-to-all of routed ones), no Python, no torch.  Runs with a 1-rank communicator on the
// single-GPU test box -- the protocol (pack -> all-gather -> unpack -> merge) is the N-rank one -- and compares the merged
// result with the plain single-process GroupAggregate.
#include <stdio.h>
#include <string.h>

#include <map>
#include <memory>
#include <vector>

#include "supersonic_amd/sharded.h"

using namespace supersonic;  // NOLINT

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

static AggregationSpecification* Spec() {
  return (new AggregationSpecification)->AddAggregation(SUM, "v", "sv")->AddAggregation(MIN, "d", "mn")->AddAggregation(MAX, "d", "mx")
      ->AddAggregation(COUNT, "v", "cv")->AddAggregation(COUNT, "", "n")->AddAggregation(FIRST, "d", "fd");
}

struct Row { int64_t sv; bool sv_null; double mn, mx; uint64_t cv, n; double fd; };

static bool Drain(Cursor* c, std::map<int32_t, Row>* out) {
  for (;;) {
    ResultView r = c->Next(Cursor::kDefaultRowCount);
    if (r.is_failure()) { printf("cursor failed: %s\n", r.exception().message().c_str()); return false; }
    if (r.is_eos()) return true;
    const View& v = r.view();
    for (rowcount_t i = 0; i < v.row_count(); ++i) {
      Row row;
      row.sv_null = v.column(1).is_null() && v.column(1).is_null()[i];
      row.sv = v.column(1).typed_data<int64_t>()[i];
      row.mn = v.column(2).typed_data<double>()[i]; row.mx = v.column(3).typed_data<double>()[i];
      row.cv = v.column(4).typed_data<uint64_t>()[i]; row.n = v.column(5).typed_data<uint64_t>()[i];
      row.fd = v.column(6).typed_data<double>()[i];
      (*out)[v.column(0).typed_data<int32_t>()[i]] = row;
    }
  }
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "build-only")) { printf("PASSED\n"); return 0; }
  const int N = 60000;
  std::vector<int32_t> k(N); std::vector<int64_t> a(N), v(N); std::vector<char> v_null(N); std::vector<double> d(N);
  uint64_t s = 99;
  for (int i = 0; i < N; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    k[i] = static_cast<int32_t>((s >> 40) % 257); a[i] = static_cast<int64_t>((s >> 20) % 1000);
    v[i] = static_cast<int64_t>((s >> 33) % 2001) - 1000; v_null[i] = ((s >> 7) % 5) == 0;
    d[i] = static_cast<double>((s >> 12) % 4096) * 0.25 - 300.0;
  }
  TupleSchema schema;
  schema.add_attribute(Attribute("k", INT32, NOT_NULLABLE)); schema.add_attribute(Attribute("a", INT64, NOT_NULLABLE));
  schema.add_attribute(Attribute("v", INT64, NULLABLE)); schema.add_attribute(Attribute("d", DOUBLE, NOT_NULLABLE));
  View view(schema);
  view.mutable_column(0)->Reset(k.data(), nullptr); view.mutable_column(1)->Reset(a.data(), nullptr);
  view.mutable_column(2)->Reset(v.data(), reinterpret_cast<const bool*>(v_null.data())); view.mutable_column(3)->Reset(d.data(), nullptr);
  view.set_row_count(N);

  int dev = 0;
  ncclComm_t comm;
  if (ncclCommInitAll(&comm, 1, &dev) != ncclSuccess) { printf("FAIL: ncclCommInitAll\n"); return 1; }

  auto shard = [&]() { return Filter(Greater(NamedAttribute("a"), ConstInt64(299)), ProjectAllAttributes(), ScanView(view)); };
  std::map<int32_t, Row> got, want;
  {
    ShardedGroupAggregate job(comm, 1, {"k"}, Spec(), shard(), /*capacity_rows=*/1024);
    FailureOrOwned<Cursor> c = job.Run();
    CHECK(c.is_success());
    if (c.is_failure()) { printf("sharded run failed: %s\n", c.exception().message().c_str()); return 1; }
    CHECK(c->schema().attribute_count() == 7);
    CHECK(c->schema().attribute(4).name() == "cv" && !c->schema().attribute(4).is_nullable());   // COUNT stays NOT NULL
    CHECK(Drain(c.get(), &got));
    CHECK(job.largest_table() == 257);
    // a second step re-runs the SAME two plans (bound once) and reuses the buffers: same result
    FailureOrOwned<Cursor> again = job.Run();
    CHECK(again.is_success());
    std::map<int32_t, Row> second;
    if (again.is_success()) CHECK(Drain(again.get(), &second));
    CHECK(second.size() == got.size());
    for (auto& kv : got) { auto it = second.find(kv.first); CHECK(it != second.end() && it->second.sv == kv.second.sv && it->second.mx == kv.second.mx && it->second.n == kv.second.n); }
  }
  {
    std::unique_ptr<Operation> plain(GroupAggregate(ProjectNamedAttribute("k"), Spec(), nullptr, shard()));
    FailureOrOwned<Cursor> c = plain->CreateCursor();
    CHECK(c.is_success());
    CHECK(Drain(c.get(), &want));
  }
  CHECK(got.size() == want.size() && got.size() == 257);
  for (auto& kv : want) {
    auto it = got.find(kv.first);
    CHECK(it != got.end());
    if (it == got.end()) continue;
    const Row& g = it->second; const Row& w = kv.second;
    CHECK(g.sv_null == w.sv_null && (g.sv_null || g.sv == w.sv));
    CHECK(g.mn == w.mn && g.mx == w.mx && g.cv == w.cv && g.n == w.n && g.fd == w.fd);
  }
  {  // the key-range exchange: rows routed by key into one image per destination, one all-to-all (here: to itself), merge of the
     // groups this rank owns -- with one rank, all of them
    std::map<int32_t, Row> ranged;
    ShardedGroupAggregate job(comm, 1, {"k"}, Spec(), shard(), /*capacity_rows=*/1024, ShardedGroupAggregate::KEY_RANGE);
    FailureOrOwned<Cursor> c = job.Run();
    CHECK(c.is_success());
    if (c.is_failure()) { printf("key-range run failed: %s\n", c.exception().message().c_str()); return 1; }
    CHECK(Drain(c.get(), &ranged));
    CHECK(ranged.size() == want.size());
    for (auto& kv : want) {
      auto it = ranged.find(kv.first);
      CHECK(it != ranged.end());
      if (it == ranged.end()) continue;
      const Row& g = it->second; const Row& w = kv.second;
      CHECK(g.sv_null == w.sv_null && (g.sv_null || g.sv == w.sv));
      CHECK(g.mn == w.mn && g.mx == w.mx && g.cv == w.cv && g.n == w.n && g.fd == w.fd);
    }
    // images too small for the partial table: nothing is truncated -- every rank reads the same trailer, regrows its images to the
    // largest table seen (with head room) and repeats the step, as distributed.py's check() does
    ShardedGroupAggregate small(comm, 1, {"k"}, Spec(), shard(), /*capacity_rows=*/64, ShardedGroupAggregate::KEY_RANGE);
    FailureOrOwned<Cursor> f = small.Run();
    CHECK(f.is_success());
    std::map<int32_t, Row> regrown;
    if (f.is_success()) { CHECK(Drain(f.get(), &regrown)); CHECK(regrown.size() == want.size()); }
  }
  {  // the DENSE exchange: key ranges agreed at set-up, then shard scan -> ONE all-to-all of slot slices (here: to itself)
     // -> element-wise fold + extraction on the same plan; no merge plan.  FIRST / LAST are not in this form (their values live in the
     // shard that saw the row): the job is refused before any collective, identically on every rank
    auto spec_dense = []() { return (new AggregationSpecification)->AddAggregation(SUM, "v", "sv")->AddAggregation(MIN, "d", "mn")->AddAggregation(MAX, "d", "mx")
                                 ->AddAggregation(COUNT, "v", "cv")->AddAggregation(COUNT, "", "n")->AddAggregation(SUM, "d", "fd"); };
    ShardedGroupAggregate job(comm, 1, {"k"}, spec_dense(), shard(), /*capacity_rows=*/0, ShardedGroupAggregate::DENSE);
    std::map<int32_t, Row> dense, dense2, plain_rows;
    FailureOrOwned<Cursor> c = job.Run();
    CHECK(c.is_success());
    if (c.is_failure()) { printf("dense run failed: %s\n", c.exception().message().c_str()); return 1; }
    CHECK(Drain(c.get(), &dense));
    CHECK(job.dense_slots() == 257);
    FailureOrOwned<Cursor> again = job.Run();            // a second step: the same plan, the same buffers
    CHECK(again.is_success());
    if (again.is_success()) CHECK(Drain(again.get(), &dense2));
    std::unique_ptr<Operation> plain(GroupAggregate(ProjectNamedAttribute("k"), spec_dense(), nullptr, shard()));
    FailureOrOwned<Cursor> pc = plain->CreateCursor();
    CHECK(pc.is_success());
    CHECK(Drain(pc.get(), &plain_rows));
    CHECK(dense.size() == plain_rows.size() && dense.size() == 257 && dense2.size() == 257);
    for (auto& kv : plain_rows) {
      for (auto* got_rows : {&dense, &dense2}) {
        auto it = got_rows->find(kv.first);
        CHECK(it != got_rows->end());
        if (it == got_rows->end()) continue;
        const Row& g = it->second; const Row& w = kv.second;
        CHECK(g.sv_null == w.sv_null && (g.sv_null || g.sv == w.sv));
        CHECK(g.mn == w.mn && g.mx == w.mx && g.cv == w.cv && g.n == w.n && g.fd == w.fd);
      }
    }
    ShardedGroupAggregate refused(comm, 1, {"k"}, Spec(), shard(), /*capacity_rows=*/0, ShardedGroupAggregate::DENSE);   // Spec() has a FIRST
    FailureOrOwned<Cursor> r = refused.Run();
    CHECK(r.is_failure());
    if (r.is_failure()) CHECK(r.exception().return_code() == ERROR_NOT_IMPLEMENTED);
  }
  {  // the same with the all-gather exchange
    ShardedGroupAggregate small(comm, 1, {"k"}, Spec(), shard(), /*capacity_rows=*/64);
    FailureOrOwned<Cursor> c = small.Run();
    CHECK(c.is_success());
    std::map<int32_t, Row> regrown;
    if (c.is_success()) {
      CHECK(Drain(c.get(), &regrown));
      CHECK(regrown.size() == want.size());
      for (auto& kv : want) { auto it = regrown.find(kv.first); CHECK(it != regrown.end()); if (it != regrown.end()) CHECK(it->second.n == kv.second.n && it->second.mn == kv.second.mn); }
    }
    CHECK(small.largest_table() == 257);
  }
  {  // DOUBLE sums across shards stay exact where the exact sum is representable (<= 1 ULP in general): two shards of ONE device
     // meet in this process like two ranks.  Group g: shard 0 holds 1e16 and 101 ones (its exact sum 1e16 + 101 is not a
     // double: the ROUNDED partial is 1e16 + 100 or + 102), shard 1 holds -1e16 and 100 ones.  Exact total: 201.  Adding the
     // shards' rounded sums gives 200 or 202; the (SUM, SUM_RESIDUAL) pairs give 201.
    const int G = 300, PER0 = 102, PER1 = 101;
    std::vector<int32_t> k0, k1; std::vector<double> x0, x1;
    for (int g = 0; g < G; ++g) {
      for (int i = 0; i < PER0; ++i) { k0.push_back(g); x0.push_back(i == 0 ? 1e16 : 1.0); }
      for (int i = 0; i < PER1; ++i) { k1.push_back(g); x1.push_back(i == 0 ? -1e16 : 1.0); }
    }
    TupleSchema s2;
    s2.add_attribute(Attribute("k", INT32, NOT_NULLABLE)); s2.add_attribute(Attribute("x", DOUBLE, NOT_NULLABLE));
    View v0(s2), v1(s2);
    v0.mutable_column(0)->Reset(k0.data(), nullptr); v0.mutable_column(1)->Reset(x0.data(), nullptr); v0.set_row_count(k0.size());
    v1.mutable_column(0)->Reset(k1.data(), nullptr); v1.mutable_column(1)->Reset(x1.data(), nullptr); v1.set_row_count(k1.size());
    for (int form = 0; form < 2; ++form) {
      std::vector<Operation*> shards;
      shards.push_back(ScanView(v0)); shards.push_back(ScanView(v1));
      ShardedGroupAggregate job({"k"}, (new AggregationSpecification)->AddAggregation(SUM, "x", "sx")->AddAggregation(COUNT, "", "n"), shards, 512,
                                form ? ShardedGroupAggregate::KEY_RANGE : ShardedGroupAggregate::ALL_GATHER);
      FailureOrOwned<Cursor> c = job.Run();
      CHECK(c.is_success());
      if (c.is_failure()) { printf("two-shard run failed: %s\n", c.exception().message().c_str()); return 1; }
      CHECK(c->schema().attribute_count() == 3);                       // the residual column is projected away
      int groups = 0, exact = 0;
      for (;;) {
        ResultView r = c->Next(-1);
        if (!r.has_data()) { CHECK(r.is_eos()); break; }
        for (rowcount_t i = 0; i < r.view().row_count(); ++i, ++groups) {
          if (r.view().column(1).typed_data<DOUBLE>()[i] == 201.0) ++exact;
          CHECK(r.view().column(2).typed_data<UINT64>()[i] == static_cast<uint64>(PER0 + PER1));
        }
      }
      CHECK(groups == G);
      CHECK(exact == G);                                                // every cross-shard sum is the exact one
      if (exact != G) printf("cross-shard DOUBLE sums: %d of %d exact (form %d)\n", exact, G, form);
    }
  }
  {  // the headline's shape at N > 1: ShardedScalarAggregate (partial run -> ONE all-gather of the state -> fold + emit) on the
     // one-rank communicator, against the plain single-process cursor; stepped twice on the same bound plan
    auto scalar_child = [&]() {
      return Filter(Greater(NamedAttribute("a"), ConstInt64(299)), ProjectAllAttributes(),
                    Compute((new CompoundExpression)->Add(NamedAttribute("a"))->AddAs("s", Plus(NamedAttribute("a"), NamedAttribute("v")))->Add(NamedAttribute("d")), ScanView(view)));
    };
    auto scalar_spec = []() {
      return (new AggregationSpecification)->AddAggregation(SUM, "s", "sum_s")->AddAggregation(COUNT, "", "cnt")->AddAggregation(MIN, "d", "min_d")
          ->AddAggregation(SUM, "d", "sum_d")->AddAggregation(LAST, "d", "last_d");
    };
    std::unique_ptr<Operation> plain(ScalarAggregate(scalar_spec(), scalar_child()));
    std::unique_ptr<Cursor> pc(SucceedOrDie(plain->CreateCursor()));
    ResultView want_row = pc->Next(-1);
    CHECK(want_row.has_data() && want_row.view().row_count() == 1u);
    ShardedScalarAggregate job(comm, 1, scalar_spec(), scalar_child());
    for (int step = 0; step < 2 && want_row.has_data(); ++step) {
      FailureOrOwned<Cursor> c = job.Run(/*global_row_offset=*/0);
      CHECK(c.is_success());
      if (c.is_failure()) { printf("sharded scalar run failed: %s\n", c.exception().message().c_str()); break; }
      ResultView got_row = c->Next(-1);
      CHECK(got_row.has_data() && got_row.view().row_count() == 1u && got_row.view().column_count() == 5);
      if (!got_row.has_data()) break;
      CHECK(got_row.view().column(0).typed_data<INT64>()[0] == want_row.view().column(0).typed_data<INT64>()[0]);
      CHECK(got_row.view().column(1).typed_data<UINT64>()[0] == want_row.view().column(1).typed_data<UINT64>()[0]);
      CHECK(got_row.view().column(2).typed_data<DOUBLE>()[0] == want_row.view().column(2).typed_data<DOUBLE>()[0]);
      CHECK(got_row.view().column(3).typed_data<DOUBLE>()[0] == want_row.view().column(3).typed_data<DOUBLE>()[0]);
      CHECK(got_row.view().column(4).typed_data<DOUBLE>()[0] == want_row.view().column(4).typed_data<DOUBLE>()[0]);
      CHECK(c->Next(-1).is_eos());
    }
  }
  {
    // what the driver refuses, loudly: STRING keys / STRING results (every plan has its own dictionary: the codes of different shards
    // must not be merged) and the row-after-row SUM of floating values into an integer (a shard's result is not a partial sum)
    StringPiece names[4] = {"pear", "apple", "fig", "apple"}; int64_t vals[4] = {1, 2, 3, 4}; double ds[4] = {0.6, 0.6, 0.6, 2.5};
    TupleSchema ss;
    ss.add_attribute(Attribute("name", STRING, NOT_NULLABLE)); ss.add_attribute(Attribute("v", INT64, NOT_NULLABLE)); ss.add_attribute(Attribute("x", DOUBLE, NOT_NULLABLE));
    View sv(ss);
    sv.mutable_column(0)->Reset(names, nullptr); sv.mutable_column(1)->Reset(vals, nullptr); sv.mutable_column(2)->Reset(ds, nullptr);
    sv.set_row_count(4);
    {
      ShardedGroupAggregate job(comm, 1, {"name"}, (new AggregationSpecification)->AddAggregation(SUM, "v", "s"), ScanView(sv), 1024);
      FailureOrOwned<Cursor> c = job.Run();
      CHECK(c.is_failure());
      if (c.is_failure()) CHECK(c.exception().return_code() == ERROR_NOT_IMPLEMENTED);
    }
    {
      ShardedGroupAggregate job(comm, 1, {"v"}, (new AggregationSpecification)->AddAggregation(MIN, "name", "m"), ScanView(sv), 1024);
      FailureOrOwned<Cursor> c = job.Run();
      CHECK(c.is_failure());
      if (c.is_failure()) CHECK(c.exception().return_code() == ERROR_NOT_IMPLEMENTED);
    }
    {
      ShardedGroupAggregate job(comm, 1, {"v"}, (new AggregationSpecification)->AddAggregationWithDefinedOutputType(SUM, "x", "s", INT64), ScanView(sv), 1024);
      FailureOrOwned<Cursor> c = job.Run();
      CHECK(c.is_failure());
      if (c.is_failure()) CHECK(c.exception().return_code() == ERROR_NOT_IMPLEMENTED);
    }
    {
      ShardedScalarAggregate job(comm, 1, (new AggregationSpecification)->AddAggregation(MAX, "name", "m"), ScanView(sv));
      FailureOrOwned<Cursor> c = job.Run(0);
      CHECK(c.is_failure());
      if (c.is_failure()) CHECK(c.exception().return_code() == ERROR_NOT_IMPLEMENTED);
    }
    {   // (COUNT of a STRING column is a number: merged)
      ShardedGroupAggregate job(comm, 1, {"v"}, (new AggregationSpecification)->AddAggregation(COUNT, "name", "c"), ScanView(sv), 1024);
      FailureOrOwned<Cursor> c = job.Run();
      CHECK(c.is_success());
      if (c.is_success()) { ResultView r = c->Next(-1); CHECK(r.has_data() && r.view().row_count() == 4u); }
    }
  }
  ncclCommDestroy(comm);
  printf(g_fail ? "FAILED (%d)\n" : "PASSED\n", g_fail);
  return g_fail ? 1 : 0;
}
