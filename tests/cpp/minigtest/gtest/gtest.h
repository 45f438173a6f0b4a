// A small stand-alone test harness with googletest's surface -- TEST / TEST_F, testing::Test fixtures, EXPECT_* / ASSERT_*
// with `<< message` streaming, RUN_ALL_TESTS -- so that test programs written for googletest (this repository's
// tests/cpp/*.cc and, read in place from a reference checkout, the reference's own test/guide/*.cc) build and RUN against
// libssgpu without googletest installed.  It is this repository's own code (googletest is not in the image and is not
// copied here); only the macro names and their documented behaviour are shared.  A program gets a main() unless it
// defines MINIGTEST_NO_MAIN.
#ifndef TESTS_CPP_MINIGTEST_GTEST_GTEST_H_
#define TESTS_CPP_MINIGTEST_GTEST_GTEST_H_

#include <stdio.h>
#include <string.h>

#include <cmath>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

namespace testing {

class Test {
 public:
  virtual ~Test() {}
  virtual void SetUp() {}
  virtual void TearDown() {}
  virtual void TestBody() = 0;
};

namespace internal {

struct Registry {
  struct Entry { const char* suite; const char* name; Test* (*make)(); };
  std::vector<Entry> tests;
  bool current_failed = false;     // any failure in the running test
  bool current_fatal = false;      // an ASSERT_* failed: the test body returned
  static Registry& Get() { static Registry r; return r; }
  int Add(const char* suite, const char* name, Test* (*make)()) { tests.push_back(Entry{suite, name, make}); return 0; }
};

// Collects the text streamed behind a failed check and prints the whole report when it goes out of scope.
class Report {
 public:
  Report(const char* file, int line, const std::string& what, bool fatal) {
    Registry::Get().current_failed = true;
    if (fatal) Registry::Get().current_fatal = true;
    text_ << file << ":" << line << ": Failure\n" << what << "\n";
  }
  ~Report() { fprintf(stdout, "%s\n", text_.str().c_str()); fflush(stdout); }
  template <typename T> Report& operator<<(const T& v) { text_ << v; return *this; }
  Report& operator<<(std::ostream& (*manip)(std::ostream&)) { text_ << manip; return *this; }
 private:
  std::ostringstream text_;
};
// `return Voidify() = Report(...) << a << b;` -- the ASSERT_* form: assignment binds looser than <<
struct Voidify { void operator=(const Report&) const {} };

template <typename T, typename = void> struct Printable : std::false_type {};
template <typename T> struct Printable<T, decltype(void(std::declval<std::ostream&>() << std::declval<const T&>()))> : std::true_type {};
template <typename T> typename std::enable_if<Printable<T>::value, std::string>::type Show(const T& v) { std::ostringstream s; s << v; return s.str(); }
template <typename T> typename std::enable_if<!Printable<T>::value, std::string>::type Show(const T&) { return "<value of " + std::to_string(sizeof(T)) + " bytes>"; }
inline std::string Show(bool v) { return v ? "true" : "false"; }
inline std::string Show(std::nullptr_t) { return "nullptr"; }

template <typename A, typename B> std::string Cmp(const char* ea, const char* eb, const char* op, const A& a, const B& b) {
  return std::string("Expected: (") + ea + ") " + op + " (" + eb + "), actual: " + Show(a) + " vs " + Show(b);
}
// comparisons between signed and unsigned integers are what user code writes (EXPECT_EQ(4, view.column_count())): compare
// by value without the compiler's sign-compare diagnostics
struct Eq { template <typename A, typename B> static bool Do(const A& a, const B& b) { return Impl(a, b, std::integral_constant<bool, std::is_integral<A>::value && std::is_integral<B>::value>()); }
  template <typename A, typename B> static bool Impl(const A& a, const B& b, std::true_type) {
    typedef typename std::common_type<A, B>::type C;
    if (std::is_signed<A>::value != std::is_signed<B>::value) {
      if (std::is_signed<A>::value && a < static_cast<A>(0)) return false;
      if (std::is_signed<B>::value && b < static_cast<B>(0)) return false;
    }
    return static_cast<C>(a) == static_cast<C>(b);
  }
  template <typename A, typename B> static bool Impl(const A& a, const B& b, std::false_type) { return a == b; }
};
inline bool StrEq(const char* a, const char* b) { return (a == nullptr || b == nullptr) ? a == b : strcmp(a, b) == 0; }
inline bool AlmostEqual(double a, double b) {          // 4 ULP, as googletest's EXPECT_DOUBLE_EQ
  if (std::isnan(a) || std::isnan(b)) return false;
  if (a == b) return true;
  long long ia, ib; memcpy(&ia, &a, 8); memcpy(&ib, &b, 8);
  if ((ia < 0) != (ib < 0)) return false;
  const long long d = ia > ib ? ia - ib : ib - ia;
  return d <= 4;
}

}  // namespace internal

inline void InitGoogleTest(int*, char**) {}

}  // namespace testing

inline int RUN_ALL_TESTS() {
  testing::internal::Registry& r = testing::internal::Registry::Get();
  int failed = 0;
  printf("[==========] Running %zu tests.\n", r.tests.size());
  for (size_t i = 0; i < r.tests.size(); ++i) {
    printf("[ RUN      ] %s.%s\n", r.tests[i].suite, r.tests[i].name); fflush(stdout);
    r.current_failed = r.current_fatal = false;
    testing::Test* t = r.tests[i].make();
    t->SetUp();
    if (!r.current_fatal) t->TestBody();
    t->TearDown();
    delete t;
    if (r.current_failed) { ++failed; printf("[  FAILED  ] %s.%s\n", r.tests[i].suite, r.tests[i].name); }
    else printf("[       OK ] %s.%s\n", r.tests[i].suite, r.tests[i].name);
  }
  printf("[==========] %zu tests ran.\n", r.tests.size());
  if (failed) printf("[  FAILED  ] %d tests.\n", failed); else printf("[  PASSED  ] %zu tests.\n", r.tests.size());
  fflush(stdout);
  return failed ? 1 : 0;
}

#define MINIGTEST_CLASS_(suite, name) suite##_##name##_Test
#define MINIGTEST_DEFINE_(suite, name, base)                                                                          \
  class MINIGTEST_CLASS_(suite, name) : public base {                                                                 \
   public:                                                                                                            \
    void TestBody() override;                                                                                         \
    static ::testing::Test* Make() { return new MINIGTEST_CLASS_(suite, name)(); }                                    \
  };                                                                                                                  \
  static int minigtest_reg_##suite##_##name __attribute__((unused)) =                                                 \
      ::testing::internal::Registry::Get().Add(#suite, #name, &MINIGTEST_CLASS_(suite, name)::Make);                  \
  void MINIGTEST_CLASS_(suite, name)::TestBody()
#define TEST(suite, name) MINIGTEST_DEFINE_(suite, name, ::testing::Test)
#define TEST_F(fixture, name) MINIGTEST_DEFINE_(fixture, name, fixture)

// (the dangling-else form keeps `EXPECT_X(...) << "text";` one statement)
#define MINIGTEST_CHECK_(ok, what, fatal_return)                                                                      \
  switch (0) case 0: default:                                                                                         \
    if (ok) ; else fatal_return ::testing::internal::Report(__FILE__, __LINE__, what, sizeof(#fatal_return) > 1)
#define MINIGTEST_NONFATAL_
#define MINIGTEST_FATAL_ return ::testing::internal::Voidify() =

#define MINIGTEST_BOOL_(c, want, F) MINIGTEST_CHECK_(static_cast<bool>(c) == want, std::string("Value of: " #c "\nExpected: ") + (want ? "true" : "false"), F)
#define MINIGTEST_CMP_(a, b, op, expr, F) MINIGTEST_CHECK_(expr, ::testing::internal::Cmp(#a, #b, op, (a), (b)), F)

#define EXPECT_TRUE(c) MINIGTEST_BOOL_(c, true, MINIGTEST_NONFATAL_)
#define EXPECT_FALSE(c) MINIGTEST_BOOL_(c, false, MINIGTEST_NONFATAL_)
#define ASSERT_TRUE(c) MINIGTEST_BOOL_(c, true, MINIGTEST_FATAL_)
#define ASSERT_FALSE(c) MINIGTEST_BOOL_(c, false, MINIGTEST_FATAL_)
#define EXPECT_EQ(a, b) MINIGTEST_CMP_(a, b, "==", ::testing::internal::Eq::Do((a), (b)), MINIGTEST_NONFATAL_)
#define ASSERT_EQ(a, b) MINIGTEST_CMP_(a, b, "==", ::testing::internal::Eq::Do((a), (b)), MINIGTEST_FATAL_)
#define EXPECT_NE(a, b) MINIGTEST_CMP_(a, b, "!=", !::testing::internal::Eq::Do((a), (b)), MINIGTEST_NONFATAL_)
#define ASSERT_NE(a, b) MINIGTEST_CMP_(a, b, "!=", !::testing::internal::Eq::Do((a), (b)), MINIGTEST_FATAL_)
#define EXPECT_LT(a, b) MINIGTEST_CMP_(a, b, "<", (a) < (b), MINIGTEST_NONFATAL_)
#define EXPECT_LE(a, b) MINIGTEST_CMP_(a, b, "<=", (a) <= (b), MINIGTEST_NONFATAL_)
#define EXPECT_GT(a, b) MINIGTEST_CMP_(a, b, ">", (a) > (b), MINIGTEST_NONFATAL_)
#define EXPECT_GE(a, b) MINIGTEST_CMP_(a, b, ">=", (a) >= (b), MINIGTEST_NONFATAL_)
#define ASSERT_LT(a, b) MINIGTEST_CMP_(a, b, "<", (a) < (b), MINIGTEST_FATAL_)
#define ASSERT_LE(a, b) MINIGTEST_CMP_(a, b, "<=", (a) <= (b), MINIGTEST_FATAL_)
#define ASSERT_GT(a, b) MINIGTEST_CMP_(a, b, ">", (a) > (b), MINIGTEST_FATAL_)
#define ASSERT_GE(a, b) MINIGTEST_CMP_(a, b, ">=", (a) >= (b), MINIGTEST_FATAL_)
#define EXPECT_STREQ(a, b) MINIGTEST_CMP_(a, b, "streq", ::testing::internal::StrEq((a), (b)), MINIGTEST_NONFATAL_)
#define ASSERT_STREQ(a, b) MINIGTEST_CMP_(a, b, "streq", ::testing::internal::StrEq((a), (b)), MINIGTEST_FATAL_)
#define EXPECT_DOUBLE_EQ(a, b) MINIGTEST_CMP_(a, b, "~=", ::testing::internal::AlmostEqual((a), (b)), MINIGTEST_NONFATAL_)
#define ASSERT_DOUBLE_EQ(a, b) MINIGTEST_CMP_(a, b, "~=", ::testing::internal::AlmostEqual((a), (b)), MINIGTEST_FATAL_)
#define ADD_FAILURE() ::testing::internal::Report(__FILE__, __LINE__, "Failed", false)
#define FAIL() return ::testing::internal::Voidify() = ::testing::internal::Report(__FILE__, __LINE__, "Failed", true)
#define SUCCEED() static_cast<void>(0)

#ifndef MINIGTEST_NO_MAIN
int main(int argc, char** argv) {
  ::testing::InitGoogleTest(&argc, argv);
  return RUN_ALL_TESTS();
}
#endif

#endif  // TESTS_CPP_MINIGTEST_GTEST_GTEST_H_
