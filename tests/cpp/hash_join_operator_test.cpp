// Mirrors relational_operators/tests/HashJoinOperator_unittest.cpp: dim (200 rows) / fact (300 rows)
// tables in 10-tuple blocks (:196-270), LongKeyHashJoinTest (:379-514), IntDuplicateKeyHashJoinTest
// (:516-690), plus semi/anti variants and a Foreman/Worker run of the same plan.  GPU work orders.
#include <algorithm>
#include <map>

#include "test_util.hpp"

using namespace quickstep;

namespace {
constexpr tuple_id kNumDimTuples = 200;
constexpr tuple_id kNumFactTuples = 300;
constexpr tuple_id kBlockSize = 10;

struct Fixture {
  CatalogRelation dim{1, "dim_table"}, fact{2, "fact_table"};
  StorageManager storage;
  Fixture() {
    for (CatalogRelation *r : {&dim, &fact}) {
      r->addAttribute("long", Type::Long());
      r->addAttribute("int", Type::Int());
    }
    // dim: long = tid, int = tid % kBlockSize ; fact: long = tid, int = tid   (:196-270)
    for (tuple_id i = 0; i < kNumDimTuples; i += kBlockSize) {
      std::int64_t l[kBlockSize];
      std::int32_t v[kBlockSize];
      for (tuple_id t = 0; t < kBlockSize; ++t) { l[t] = i + t; v[t] = (i + t) % kBlockSize; }
      storage.loadBlock(&dim, {l, v}, kBlockSize);
    }
    for (tuple_id i = 0; i < kNumFactTuples; i += kBlockSize) {
      std::int64_t l[kBlockSize];
      std::int32_t v[kBlockSize];
      for (tuple_id t = 0; t < kBlockSize; ++t) { l[t] = i + t; v[t] = i + t; }
      storage.loadBlock(&fact, {l, v}, kBlockSize);
    }
  }
};

struct Result {
  std::vector<std::int64_t> dim_long, fact_long;
};

Result collect(QueryContext &ctx, QueryContext::insert_destination_id dest_id, StorageManager &storage, bool two_columns) {
  Result r;
  for (block_id b : ctx.getInsertDestination(dest_id)->getTouchedBlocks()) {
    BlockReference blk = storage.getBlock(b);
    const std::size_t at = r.dim_long.size(), k = static_cast<std::size_t>(blk->numTuples());
    r.dim_long.resize(at + k);
    blk->copyAttributeToHost(0, r.dim_long.data() + at);
    if (two_columns) {
      r.fact_long.resize(at + k);
      blk->copyAttributeToHost(1, r.fact_long.data() + at);
    }
  }
  return r;
}

// exact_stats: the optimizer knows the exact min/max of the build key (dim.long = 0..199, dim.int = 0..9):
// the join table is the directly addressed flavour.
bool g_exact_stats = false;

void runJoin(attribute_id key_attr, TypeID key_type, bool use_foreman, HashJoinOperator::JoinType join_type, Result *out) {
  Fixture f;
  CatalogRelation result(3, "result");
  QueryContext ctx;
  const QueryContext::ExactKeyRange range{0, key_type == kLong ? kNumDimTuples - 1 : kBlockSize - 1};
  const auto table = ctx.addJoinHashTable(key_type, kNumDimTuples, 1, g_exact_stats ? &range : nullptr);
  const auto dest = ctx.addInsertDestination(&result, &f.storage);
  std::vector<bool> on_build;
  QueryContext::scalar_group_id selection;
  const bool inner = join_type == HashJoinOperator::JoinType::kInnerJoin;
  if (inner) {
    result.addAttribute("dim_long", Type::Long());
    result.addAttribute("fact_long", Type::Long());
    selection = ctx.addScalarGroup({0, 0});
    on_build = {true, false};
  } else {
    result.addAttribute("fact_long", Type::Long());
    selection = ctx.addScalarGroup({0});
    on_build = {false};
  }
  auto *builder = new BuildHashOperator(0, f.dim, true, {key_attr}, false, 1, table);
  auto *prober = new HashJoinOperator(0, f.dim, f.fact, true, {key_attr}, false, 1, false, result, dest, table,
                                      QueryContext::kInvalidPredicateId, selection, &on_build, join_type);
  auto *cleaner = new DestroyHashOperator(0, 1, table);
  if (use_foreman) {
    QueryPlan plan;
    const auto b = plan.addRelationalOperator(builder);
    const auto p = plan.addRelationalOperator(prober);
    const auto c = plan.addRelationalOperator(cleaner);
    plan.addDirectDependency(p, b, true);   // BuildHash -> HashJoin is a pipeline breaker
    plan.addDirectDependency(c, p, true);
    ForemanSingleNode foreman(&plan, &ctx, &f.storage, 4);
    foreman.run();
    EXPECT_EQ(foreman.getWorkOrderProfilingResults().size(),
              static_cast<std::size_t>(kNumDimTuples / kBlockSize + kNumFactTuples / kBlockSize + 1));
  } else {
    std::unique_ptr<RelationalOperator> b(builder), p(prober), c(cleaner);
    fetchAndExecuteWorkOrders(b.get(), &ctx, &f.storage);
    fetchAndExecuteWorkOrders(p.get(), &ctx, &f.storage);
    *out = collect(ctx, dest, f.storage, inner);
    fetchAndExecuteWorkOrders(c.get(), &ctx, &f.storage);
    return;
  }
  *out = collect(ctx, dest, f.storage, inner);
}
}  // namespace

int main() {
  if (qsx_device_count() < 1) {
    std::fprintf(stderr, "hash_join_operator_test needs an MI355X: %s\n", qsx_status_string(QSX_ERR_NO_DEVICE));
    return 2;
  }
  for (const int variant : {0, 1, 2, 3}) {
    const bool use_foreman = (variant & 1) != 0;
    g_exact_stats = (variant & 2) != 0;
    {  // LongKeyHashJoinTest: 200 results, every dim.long exactly once (:510-514)
      Result r;
      runJoin(0, kLong, use_foreman, HashJoinOperator::JoinType::kInnerJoin, &r);
      EXPECT_EQ(r.dim_long.size(), static_cast<std::size_t>(kNumDimTuples));
      std::vector<int> counts(kNumDimTuples, 0);
      for (std::size_t i = 0; i < r.dim_long.size(); ++i) {
        EXPECT_TRUE(r.dim_long[i] >= 0 && r.dim_long[i] < kNumDimTuples);
        ++counts[r.dim_long[i]];
        EXPECT_EQ(r.dim_long[i], r.fact_long[i]);
      }
      for (int c : counts) EXPECT_EQ(c, 1);
    }
    {  // IntDuplicateKeyHashJoinTest: 200 results, each dim row once, fact rows 0..9 twenty times (:673-690)
      Result r;
      runJoin(1, kInt, use_foreman, HashJoinOperator::JoinType::kInnerJoin, &r);
      EXPECT_EQ(r.dim_long.size(), static_cast<std::size_t>(kNumDimTuples));
      std::vector<int> dim_counts(kNumDimTuples, 0), fact_counts(kNumFactTuples, 0);
      for (std::size_t i = 0; i < r.dim_long.size(); ++i) {
        ++dim_counts[r.dim_long[i]];
        ++fact_counts[r.fact_long[i]];
      }
      for (int c : dim_counts) EXPECT_EQ(c, 1);
      for (tuple_id i = 0; i < kNumFactTuples; ++i) EXPECT_EQ(fact_counts[i], i < kBlockSize ? kNumDimTuples / kBlockSize : 0);
    }
    {  // semi / anti on the long key: fact rows < 200 have a match, the other 100 do not
      Result semi, anti;
      runJoin(0, kLong, use_foreman, HashJoinOperator::JoinType::kLeftSemiJoin, &semi);
      runJoin(0, kLong, use_foreman, HashJoinOperator::JoinType::kLeftAntiJoin, &anti);
      std::sort(semi.dim_long.begin(), semi.dim_long.end());
      std::sort(anti.dim_long.begin(), anti.dim_long.end());
      EXPECT_EQ(semi.dim_long.size(), static_cast<std::size_t>(kNumDimTuples));
      EXPECT_EQ(anti.dim_long.size(), static_cast<std::size_t>(kNumFactTuples - kNumDimTuples));
      for (std::size_t i = 0; i < semi.dim_long.size(); ++i) EXPECT_EQ(semi.dim_long[i], static_cast<std::int64_t>(i));
      for (std::size_t i = 0; i < anti.dim_long.size(); ++i) EXPECT_EQ(anti.dim_long[i], static_cast<std::int64_t>(kNumDimTuples + i));
    }
  }
  return finish("hash_join_operator_test");
}
