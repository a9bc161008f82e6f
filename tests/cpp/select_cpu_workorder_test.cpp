// BASELINE config 1: "SelectOperator: 10M-row INTEGER column, predicate col < K, CPU reference
// WorkOrder (plumbing, no GPU)".  Runs SelectOperator -> SelectWorkOrder::execute() through the
// ForemanSingleNode/Worker stand-in with blocks in HOST memory and the CPU work order; checks the
// selected multiset and prints rows/s.  Needs no GPU.
//   usage: select_cpu_workorder_test [rows=10000000] [workers=hardware_concurrency]
#include <algorithm>
#include <chrono>
#include <random>

#include "test_util.hpp"

using namespace quickstep;

int main(int argc, char **argv) {
  const std::int64_t n = argc > 1 ? std::atoll(argv[1]) : 10000000;
  const std::size_t workers = argc > 2 ? std::atoi(argv[2]) : std::max(1u, std::thread::hardware_concurrency());
  UseHostMemoryForBlocks(true);
  // BasicColumnStore 2 MB blocks of one INT column hold 524 286 tuples
  const std::int64_t kBlockRows = 524286;
  std::mt19937_64 rng(1);
  std::vector<std::int32_t> col(static_cast<std::size_t>(n));
  for (auto &v : col) v = static_cast<std::int32_t>(rng() >> 33);  // uniform in [0, 2^31)

  for (const std::int32_t k : {21474836, 214748364, 1073741824}) {  // ~1 %, 10 %, 50 %
    CatalogRelation input(1, "input"), output(2, "output");
    input.addAttribute("col", Type::Int());
    output.addAttribute("col", Type::Int());
    StorageManager storage;
    for (std::int64_t b = 0; b < n; b += kBlockRows) {
      const std::int64_t rows = std::min(kBlockRows, n - b);
      storage.loadBlock(&input, {col.data() + b}, rows);
    }
    QueryContext ctx;
    Predicate pred;
    pred.conjuncts.push_back({0, ComparisonID::kLess, TypedLiteral::Int(k)});
    const auto pred_id = ctx.addPredicate(pred);
    const auto dest_id = ctx.addInsertDestination(&output, &storage);
    QueryPlan plan;
    plan.addRelationalOperator(new SelectOperator(0, input, false, output, dest_id, pred_id, std::vector<attribute_id>{0},
                                                  true, /*on_gpu=*/false));
    ForemanSingleNode foreman(&plan, &ctx, &storage, workers);
    const auto t0 = std::chrono::steady_clock::now();
    foreman.run();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    std::vector<std::int32_t> got;
    for (block_id b : ctx.getInsertDestination(dest_id)->getTouchedBlocks()) {
      BlockReference blk = storage.getBlock(b);
      const std::size_t at = got.size();
      got.resize(at + static_cast<std::size_t>(blk->numTuples()));
      blk->copyAttributeToHost(0, got.data() + at);
    }
    std::vector<std::int32_t> want;
    for (std::int32_t v : col) if (v < k) want.push_back(v);
    std::sort(got.begin(), got.end());
    std::sort(want.begin(), want.end());
    EXPECT_EQ(got.size(), want.size());
    EXPECT_TRUE(got == want);
    EXPECT_EQ(foreman.getWorkOrderProfilingResults().size(), static_cast<std::size_t>((n + kBlockRows - 1) / kBlockRows));
    std::printf("select col < %d: %lld rows, %zu selected, %zu workers, %.3f ms, %.1f M rows/s\n", k,
                static_cast<long long>(n), got.size(), workers, secs * 1e3, n / secs / 1e6);
  }
  return finish("select_cpu_workorder_test");
}
