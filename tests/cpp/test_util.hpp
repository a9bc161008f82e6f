// test_util.hpp — tiny assertion helpers + the synchronous work-order driver the
// reference's operator unit tests use (HashJoinOperator_unittest.cpp:341-359,
// AggregationOperator_unittest.cpp:416-457): no Foreman, no Worker, the test
// thread calls getAllWorkOrders() and execute()s what comes out.
#ifndef QSX_TESTS_CPP_TEST_UTIL_HPP_
#define QSX_TESTS_CPP_TEST_UTIL_HPP_

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "quickstep_gpu.hpp"

static int g_failures = 0;

#define EXPECT_TRUE(cond)                                                         \
  do {                                                                            \
    if (!(cond)) {                                                                \
      std::fprintf(stderr, "%s:%d: EXPECT_TRUE(%s) failed\n", __FILE__, __LINE__, #cond); \
      ++g_failures;                                                               \
    }                                                                             \
  } while (0)
#define EXPECT_EQ(a, b)                                                           \
  do {                                                                            \
    const auto va__ = (a);                                                        \
    const auto vb__ = (b);                                                        \
    if (!(va__ == vb__)) {                                                        \
      std::fprintf(stderr, "%s:%d: EXPECT_EQ(%s, %s) failed: %lld vs %lld\n", __FILE__, __LINE__, #a, #b, \
                   static_cast<long long>(va__), static_cast<long long>(vb__));   \
      ++g_failures;                                                               \
    }                                                                             \
  } while (0)
#define EXPECT_NEAR(a, b, tol)                                                    \
  do {                                                                            \
    const double va__ = (a), vb__ = (b), vt__ = std::fabs(tol);                   \
    if (!(std::fabs(va__ - vb__) <= vt__)) {                                      \
      std::fprintf(stderr, "%s:%d: EXPECT_NEAR(%s, %s) failed: %.17g vs %.17g\n", __FILE__, __LINE__, #a, #b, va__, vb__); \
      ++g_failures;                                                               \
    }                                                                             \
  } while (0)

inline void fetchAndExecuteWorkOrders(quickstep::RelationalOperator *op, quickstep::QueryContext *query_context,
                                      quickstep::StorageManager *storage_manager) {
  quickstep::WorkOrdersContainer container(1);
  tmb::MessageBus bus;
  op->setOperatorIndex(0);
  op->getAllWorkOrders(&container, query_context, storage_manager, 0, &bus);
  while (container.hasNormalWorkOrder(0)) {
    std::unique_ptr<quickstep::WorkOrder> wo(container.getNormalWorkOrder(0));
    wo->execute();
  }
}

inline int finish(const char *name) {
  if (g_failures == 0) {
    std::printf("[  PASSED  ] %s\n", name);
    return 0;
  }
  std::printf("[  FAILED  ] %s (%d failures)\n", name, g_failures);
  return 1;
}

#endif  // QSX_TESTS_CPP_TEST_UTIL_HPP_
