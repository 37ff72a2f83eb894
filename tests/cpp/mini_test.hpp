// A few macros instead of gtest (the reference vendors gtest-1.7.0; it is not on the GPU box).
#pragma once
#include <cmath>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

struct TestCase { std::string name; std::function<void()> fn; };
inline std::vector<TestCase>& registry() { static std::vector<TestCase> r; return r; }
inline int& failures() { static int f = 0; return f; }
struct Registrar { Registrar(const char* n, std::function<void()> f) { registry().push_back({n, f}); } };
#define TEST(suite, name) static void suite##_##name(); static Registrar reg_##suite##_##name(#suite "." #name, suite##_##name); static void suite##_##name()
#define FAIL_MSG(...) do { std::printf("  FAILED %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); ++failures(); } while (0)
#define EXPECT_TRUE(c) do { if (!(c)) FAIL_MSG("%s", #c); } while (0)
#define EXPECT_EQ(a, b) do { if (!((a) == (b))) FAIL_MSG("%s == %s", #a, #b); } while (0)
#define EXPECT_NEAR(a, b, tol) do { const double a_ = (a), b_ = (b); if (!(std::fabs(a_ - b_) <= (tol))) FAIL_MSG("%s = %.10g, expected %.10g +- %g", #b, b_, a_, (double)(tol)); } while (0)
// gtest's EXPECT_FLOAT_EQ is 4 ULP
#define EXPECT_FLOAT_EQ(a, b) do { const float a_ = (a), b_ = (b); const float m_ = std::fmax(std::fabs(a_), std::fabs(b_)); \
    const float ulp_ = std::nextafter(m_, INFINITY) - m_; if (!(std::fabs(a_ - b_) <= 4 * ulp_)) FAIL_MSG("%s = %.9g, expected %.9g (4 ulp)", #b, b_, a_); } while (0)
inline int run_all_tests()
{
    for (auto& t : registry()) {
        const int before = failures();
        try { t.fn(); } catch (const std::exception& e) { FAIL_MSG("exception: %s", e.what()); }
        std::printf("[%s] %s\n", failures() == before ? "  OK  " : "FAILED", t.name.c_str());
    }
    std::printf("%d failure(s) in %zu test(s)\n", failures(), registry().size());
    return failures() ? 1 : 0;
}
