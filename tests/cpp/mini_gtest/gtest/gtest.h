// Minimal test-harness header with the handful of GoogleTest names the reference's test sources use
// (TEST, TEST_F, ::testing::Test with SetUp(), EXPECT_TRUE / EXPECT_NEAR / EXPECT_DOUBLE_EQ, InitGoogleTest,
// RUN_ALL_TESTS). GoogleTest itself is not installed in this image. It exists so that the reference's OWN test files
// (/root/reference/tests/src/*.cc, compiled where they lie, never copied) can be built against this repository's
// drop-in header and run on the GPU — see tests/cpp/Makefile target `reference_tests`.
// Set LTP_TEST_FILTER to a substring to run only matching tests.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <functional>
#include <string>
#include <vector>

namespace testing {

class Test {
 public:
  virtual ~Test() {}
  virtual void SetUp() {}
  virtual void TearDown() {}
  virtual void TestBody() = 0;
};

struct Registry {
  struct Entry { std::string name; std::function<Test*()> make; };
  static std::vector<Entry>& tests() { static std::vector<Entry> t; return t; }
  static long& checks() { static long c = 0; return c; }
  static long& failures() { static long f = 0; return f; }
  static bool add(const char* suite, const char* name, std::function<Test*()> make) {
    tests().push_back({std::string(suite) + "." + name, make});
    return true;
  }
};

inline void InitGoogleTest(int*, char**) {}

inline void report(bool ok, const char* file, int line, const std::string& what) {
  ++Registry::checks();
  if (!ok) {
    ++Registry::failures();
    if (Registry::failures() <= 50) std::printf("FAIL %s:%d  %s\n", file, line, what.c_str());
  }
}

}  // namespace testing

inline int RUN_ALL_TESTS() {
  const char* filter = std::getenv("LTP_TEST_FILTER");
  int ran = 0;
  for (auto& e : testing::Registry::tests()) {
    if (filter && *filter && e.name.find(filter) == std::string::npos) continue;
    const long before = testing::Registry::failures(), c0 = testing::Registry::checks();
    std::printf("[ RUN  ] %s\n", e.name.c_str());
    std::fflush(stdout);
    testing::Test* t = e.make();
    t->SetUp();
    t->TestBody();
    t->TearDown();
    delete t;
    std::printf("[ %s ] %s (%ld checks)\n", testing::Registry::failures() == before ? " OK " : "FAIL", e.name.c_str(),
                testing::Registry::checks() - c0);
    ++ran;
  }
  std::printf("%d tests, %ld checks, %ld failures\n", ran, testing::Registry::checks(), testing::Registry::failures());
  return testing::Registry::failures() ? 1 : 0;
}

#define LTP_TEST_CLASS(suite, name) suite##_##name##_Test
#define LTP_DEFINE_TEST(suite, name, base)                                                                          \
  class LTP_TEST_CLASS(suite, name) : public base {                                                                 \
   public:                                                                                                          \
    void TestBody() override;                                                                                       \
  };                                                                                                                \
  static bool suite##_##name##_registered =                                                                         \
      ::testing::Registry::add(#suite, #name, []() -> ::testing::Test* { return new LTP_TEST_CLASS(suite, name)(); }); \
  void LTP_TEST_CLASS(suite, name)::TestBody()
#define TEST(suite, name) LTP_DEFINE_TEST(suite, name, ::testing::Test)
#define TEST_F(fixture, name) LTP_DEFINE_TEST(fixture, name, fixture)

#define EXPECT_TRUE(c) ::testing::report(static_cast<bool>(c), __FILE__, __LINE__, std::string("EXPECT_TRUE(") + #c + ")")
#define EXPECT_FALSE(c) ::testing::report(!static_cast<bool>(c), __FILE__, __LINE__, std::string("EXPECT_FALSE(") + #c + ")")
#define EXPECT_NEAR(a, b, tol)                                                                                      \
  do {                                                                                                              \
    const double a_ = (a), b_ = (b), t_ = (tol);                                                                    \
    ::testing::report(std::fabs(a_ - b_) <= t_, __FILE__, __LINE__,                                                 \
                      std::string("EXPECT_NEAR(") + #a + ", " + #b + ") " + std::to_string(a_) + " vs " + std::to_string(b_)); \
  } while (0)
#define EXPECT_DOUBLE_EQ(a, b)                                                                                      \
  do {                                                                                                              \
    const double a_ = (a), b_ = (b);                                                                                \
    ::testing::report(a_ == b_ || std::fabs(a_ - b_) <= 4 * 2.220446049250313e-16 * std::fmax(std::fabs(a_), std::fabs(b_)), \
                      __FILE__, __LINE__, std::string("EXPECT_DOUBLE_EQ(") + #a + ", " + #b + ")");                 \
  } while (0)
