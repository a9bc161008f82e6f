// Mirrors relational_operators/tests/HashJoinOperator_unittest.cpp: dim (200 rows) / fact (300 rows)
// tables in 10-tuple blocks (:196-270), LongKeyHashJoinTest (:379-514), IntDuplicateKeyHashJoinTest
// (:516-690), CompositeKeyHashJoinTest (:999-1177), CompositeKeyHashJoinWithResidualPredicateTest (:1187-1375),
// plus semi/anti/outer variants and a Foreman/Worker run of the same plans.  GPU work orders.
#include <algorithm>
#include <cstdio>
#include <map>

#include "test_util.hpp"

using namespace quickstep;

namespace {
constexpr tuple_id kNumDimTuples = 200;
constexpr tuple_id kNumFactTuples = 300;
constexpr tuple_id kBlockSize = 10;

constexpr std::size_t kMultiplePartitions = 4;   // HashJoinOperator_unittest.cpp:101

// partitioned = insertTuplesWithSingleAttributePartitions (:272-339): HashPartitionSchemeHeader(4, {long}) on both
// tables, tuple tid lives in partition tid % 4 (= long & 3, PartitionSchemeHeader.hpp:207-214), one block per partition.
// dim_partitioned = false with partitioned = true is the broadcast build (BuildHashOperator.hpp:99, 146-152).
struct Fixture {
  CatalogRelation dim{1, "dim_table"}, fact{2, "fact_table"};
  StorageManager storage;
  explicit Fixture(bool partitioned = false, bool dim_partitioned = true) {
    if (partitioned) {
      loadPartitioned(&fact, kNumFactTuples, false, true);
      loadPartitioned(&dim, kNumDimTuples, true, dim_partitioned);
      return;
    }
    loadUnpartitioned();
  }
  void addAttributes(CatalogRelation *r) {
    r->addAttribute("long", Type::Long());
    r->addAttribute("int", Type::Int());
    r->addAttribute("varchar_as_long", Type::Long());
    r->addAttribute("varchar_as_int", Type::Int());
    r->addAttribute("tid_int", Type::Int());
    r->addAttribute("char", Type::Char(4));              // "100" in every tuple of both tables (:196-270)
    r->addAttribute("varchar_as_char", Type::Char(8));   // the digits of tid / 2 * 2 (dim) resp. tid (fact), as the reference's VARCHAR column holds them
  }
  static void chars(std::int64_t varchar_value, std::vector<char> *c4, std::vector<char> *c8) {
    char a[4] = {'1', '0', '0', 0}, b[24] = {0};
    std::snprintf(b, sizeof(b), "%lld", static_cast<long long>(varchar_value));
    c4->insert(c4->end(), a, a + 4);
    c8->insert(c8->end(), b, b + 8);
  }
  void loadPartitioned(CatalogRelation *r, tuple_id num_tuples, bool is_dim, bool with_scheme) {
    addAttributes(r);
    const std::size_t parts = with_scheme ? kMultiplePartitions : 1;
    if (with_scheme) r->setPartitionScheme(kMultiplePartitions, 0);
    for (std::size_t part = 0; part < parts; ++part) {
      std::vector<std::int64_t> l, vl;
      std::vector<std::int32_t> v, vi, ti;
      std::vector<char> c4, c8;
      for (tuple_id tid = 0; tid < num_tuples; ++tid) {
        if (with_scheme && static_cast<std::size_t>(tid) % kMultiplePartitions != part) continue;
        l.push_back(tid); v.push_back(is_dim ? tid % kBlockSize : tid);
        vl.push_back(is_dim ? tid / 2 * 2 : tid); vi.push_back(is_dim ? tid / 2 * 2 : tid); ti.push_back(tid);
        chars(is_dim ? tid / 2 * 2 : tid, &c4, &c8);
      }
      storage.loadBlock(r, {l.data(), v.data(), vl.data(), vi.data(), ti.data(), c4.data(), c8.data()}, static_cast<std::int64_t>(l.size()), part);
    }
  }
  void loadUnpartitioned() {
    // the VARCHAR column of the reference (digits of tid/2*2 resp. tid) is carried as integers,
    // once LONG (composite key wider than 8 bytes: hashed fold) and once INT next to an INT copy
    // of tid (8 bytes: exact packing)
    for (CatalogRelation *r : {&dim, &fact}) addAttributes(r);
    // dim: long = tid, int = tid % kBlockSize, varchar = tid / 2 * 2 ; fact: long = int = varchar = tid   (:196-270)
    for (tuple_id i = 0; i < kNumDimTuples; i += kBlockSize) {
      std::int64_t l[kBlockSize], vl[kBlockSize];
      std::int32_t v[kBlockSize], vi[kBlockSize], ti[kBlockSize];
      std::vector<char> c4, c8;
      for (tuple_id t = 0; t < kBlockSize; ++t) {
        l[t] = i + t; v[t] = (i + t) % kBlockSize; vl[t] = (i + t) / 2 * 2; vi[t] = (i + t) / 2 * 2; ti[t] = i + t;
        chars((i + t) / 2 * 2, &c4, &c8);
      }
      storage.loadBlock(&dim, {l, v, vl, vi, ti, c4.data(), c8.data()}, kBlockSize);
    }
    for (tuple_id i = 0; i < kNumFactTuples; i += kBlockSize) {
      std::int64_t l[kBlockSize], vl[kBlockSize];
      std::int32_t v[kBlockSize], vi[kBlockSize], ti[kBlockSize];
      std::vector<char> c4, c8;
      for (tuple_id t = 0; t < kBlockSize; ++t) {
        l[t] = i + t; v[t] = i + t; vl[t] = i + t; vi[t] = i + t; ti[t] = i + t;
        chars(i + t, &c4, &c8);
      }
      storage.loadBlock(&fact, {l, v, vl, vi, ti, c4.data(), c8.data()}, kBlockSize);
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
// 0: unpartitioned; 1: both tables hash-partitioned 4 ways on `long` (the ...WithSingleAttributePartitions tests,
// :1379-1875); 2: only the probe side partitioned — broadcast build into all 4 tables.
int g_partitioning = 0;
// > 1: BuildHash and HashJoin work orders cover runs of this many blocks (setBlocksPerWorkOrder)
std::size_t g_blocks_per_work_order = 1;

void runJoin(attribute_id key_attr, TypeID key_type, bool use_foreman, HashJoinOperator::JoinType join_type, Result *out) {
  Fixture f(g_partitioning != 0, g_partitioning == 1);
  const std::size_t parts = g_partitioning != 0 ? kMultiplePartitions : 1;
  CatalogRelation result(3, "result");
  QueryContext ctx;
  const QueryContext::ExactKeyRange range{0, key_type == kLong ? kNumDimTuples - 1 : kBlockSize - 1};
  const auto table = ctx.addJoinHashTable(key_type, kNumDimTuples, parts, g_exact_stats ? &range : nullptr);
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
  auto *builder = new BuildHashOperator(0, f.dim, true, {key_attr}, false, parts, table);
  auto *prober = new HashJoinOperator(0, f.dim, f.fact, true, {key_attr}, false, parts, false, result, dest, table,
                                      QueryContext::kInvalidPredicateId, selection, &on_build, join_type);
  auto *cleaner = new DestroyHashOperator(0, parts, table);
  builder->setBlocksPerWorkOrder(g_blocks_per_work_order);
  prober->setBlocksPerWorkOrder(g_blocks_per_work_order);
  if (use_foreman) {
    QueryPlan plan;
    const auto b = plan.addRelationalOperator(builder);
    const auto p = plan.addRelationalOperator(prober);
    const auto c = plan.addRelationalOperator(cleaner);
    plan.addDirectDependency(p, b, true);   // BuildHash -> HashJoin is a pipeline breaker
    plan.addDirectDependency(c, p, true);
    ForemanSingleNode foreman(&plan, &ctx, &f.storage, 4);
    foreman.run();
    const std::size_t build_orders = g_partitioning == 0 ? kNumDimTuples / kBlockSize
                                     : g_partitioning == 1 ? kMultiplePartitions : kMultiplePartitions /* 1 block x 4 tables */;
    const std::size_t probe_orders = g_partitioning == 0 ? kNumFactTuples / kBlockSize : kMultiplePartitions;
    if (g_blocks_per_work_order == 1) EXPECT_EQ(foreman.getWorkOrderProfilingResults().size(), build_orders + probe_orders + parts);
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

// Composite keys / residual predicate / outer join.  Output: (dim.long [build side], fact.long [probe side]) for
// inner and outer joins, fact.long for semi and anti joins.
struct JoinedRows {
  std::vector<std::int64_t> dim_long, fact_long;
  std::vector<bool> dim_is_null;
};

JoinedRows runGeneralJoin(const std::vector<attribute_id> &keys, TypeID table_key_type, const Predicate *residual,
                          HashJoinOperator::JoinType join_type, bool use_foreman) {
  // g_partitioning == 1: SingleAttributePartitionedCompositeKeyHashJoin[WithResidualPredicate]Test (:1524-1875) —
  // every composite key here contains the tuple id, so equal keys share a partition
  Fixture f(g_partitioning != 0, g_partitioning == 1);
  const std::size_t parts = g_partitioning != 0 ? kMultiplePartitions : 1;
  CatalogRelation result(4, "result");
  QueryContext ctx;
  const auto table = ctx.addJoinHashTable(table_key_type, kNumDimTuples, parts);
  const auto dest = ctx.addInsertDestination(&result, &f.storage);
  const bool pairs = join_type == HashJoinOperator::JoinType::kInnerJoin || join_type == HashJoinOperator::JoinType::kLeftOuterJoin;
  std::vector<bool> on_build;
  QueryContext::scalar_group_id selection;
  if (pairs) {
    result.addAttribute("dim_long", Type::Long().getNullableVersion());
    result.addAttribute("fact_long", Type::Long());
    selection = ctx.addScalarGroup({0, 0});
    on_build = {true, false};
  } else {
    result.addAttribute("fact_long", Type::Long());
    selection = ctx.addScalarGroup({0});
    on_build = {false};
  }
  const auto residual_id = residual != nullptr ? ctx.addPredicate(*residual) : QueryContext::kInvalidPredicateId;
  auto *builder = new BuildHashOperator(0, f.dim, true, keys, false, parts, table);
  auto *prober = new HashJoinOperator(0, f.dim, f.fact, true, keys, false, parts, false, result, dest, table, residual_id, selection,
                                      &on_build, join_type);
  auto *cleaner = new DestroyHashOperator(0, parts, table);
  builder->setBlocksPerWorkOrder(g_blocks_per_work_order);
  prober->setBlocksPerWorkOrder(g_blocks_per_work_order);
  std::unique_ptr<RelationalOperator> b, p, c;
  if (use_foreman) {
    QueryPlan plan;
    const auto bi = plan.addRelationalOperator(builder);
    const auto pi = plan.addRelationalOperator(prober);
    const auto ci = plan.addRelationalOperator(cleaner);
    plan.addDirectDependency(pi, bi, true);
    plan.addDirectDependency(ci, pi, true);
    ForemanSingleNode foreman(&plan, &ctx, &f.storage, 4);
    foreman.run();
  } else {
    b.reset(builder); p.reset(prober); c.reset(cleaner);
    fetchAndExecuteWorkOrders(b.get(), &ctx, &f.storage);
    fetchAndExecuteWorkOrders(p.get(), &ctx, &f.storage);
  }
  JoinedRows rows;
  for (block_id blk_id : ctx.getInsertDestination(dest)->getTouchedBlocks()) {
    BlockReference blk = f.storage.getBlock(blk_id);
    const std::size_t at = rows.fact_long.size(), k = static_cast<std::size_t>(blk->numTuples());
    rows.fact_long.resize(at + k);
    if (pairs) {
      rows.dim_long.resize(at + k);
      blk->copyAttributeToHost(0, rows.dim_long.data() + at);
      blk->copyAttributeToHost(1, rows.fact_long.data() + at);
      std::vector<std::uint64_t> nulls((k + 63) / 64 + 1, 0);
      blk->copyNullBitmapToHost(0, nulls.data());
      for (std::size_t i = 0; i < k; ++i) rows.dim_is_null.push_back((nulls[i >> 6] >> (63 - (i & 63))) & 1u);
    } else {
      blk->copyAttributeToHost(0, rows.fact_long.data() + at);
    }
  }
  if (!use_foreman) fetchAndExecuteWorkOrders(c.get(), &ctx, &f.storage);
  return rows;
}
}  // namespace

int main() {
  if (qsx_device_count() < 1) {
    std::fprintf(stderr, "hash_join_operator_test needs an MI355X: %s\n", qsx_status_string(QSX_ERR_NO_DEVICE));
    return 2;
  }
  for (const int variant : {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13}) {
    const bool use_foreman = (variant & 1) != 0;
    g_exact_stats = (variant & 2) != 0;
    g_partitioning = variant >= 10 || variant < 4 ? 0 : (variant < 8 ? 1 : 2);   // 8, 9: broadcast build, hashed tables
    g_blocks_per_work_order = variant >= 10 ? 7 : 1;                              // 10..13: runs of 7 of the 10-tuple blocks
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
    if (g_partitioning != 1)   // dim.int is not the partitioning attribute: equal keys sit in different partitions
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
  // ---- CHAR join keys (carried as the LONG qsx_join_key_pack_char makes of them) ---------------------------------------
  for (const int variant : {0, 1, 2, 3}) {
    const bool use_foreman = (variant & 1) != 0;
    g_exact_stats = false;
    g_partitioning = 0;
    g_blocks_per_work_order = variant >= 2 ? 7 : 1;
    {  // CharKeyCartesianProductHashJoinTest (:692-826): every tuple holds "100": 200 x 300 results, every dim row 300 times
      Result r;
      runJoin(5, kChar, use_foreman, HashJoinOperator::JoinType::kInnerJoin, &r);
      EXPECT_EQ(r.dim_long.size(), static_cast<std::size_t>(kNumDimTuples) * kNumFactTuples);
      std::vector<int> dim_counts(kNumDimTuples, 0), fact_counts(kNumFactTuples, 0);
      for (std::size_t i = 0; i < r.dim_long.size(); ++i) {
        ++dim_counts[r.dim_long[i]];
        ++fact_counts[r.fact_long[i]];
      }
      for (int c : dim_counts) EXPECT_EQ(c, kNumFactTuples);
      for (int c : fact_counts) EXPECT_EQ(c, kNumDimTuples);
    }
    {  // VarCharDuplicateKeyHashJoinTest (:828-997), the strings as CHAR(8): dim holds the digits of tid / 2 * 2, fact of tid —
       // 200 results, every dim row once, the even fact rows below 200 twice
      Result r;
      runJoin(6, kChar, use_foreman, HashJoinOperator::JoinType::kInnerJoin, &r);
      EXPECT_EQ(r.dim_long.size(), static_cast<std::size_t>(kNumDimTuples));
      std::vector<int> dim_counts(kNumDimTuples, 0), fact_counts(kNumFactTuples, 0);
      for (std::size_t i = 0; i < r.dim_long.size(); ++i) {
        ++dim_counts[r.dim_long[i]];
        ++fact_counts[r.fact_long[i]];
        EXPECT_EQ(r.dim_long[i] / 2 * 2, r.fact_long[i]);
      }
      for (int c : dim_counts) EXPECT_EQ(c, 1);
      for (tuple_id i = 0; i < kNumFactTuples; ++i) EXPECT_EQ(fact_counts[i], (i < kNumDimTuples && (i & 1) == 0) ? 2 : 0);
    }
  }
  g_blocks_per_work_order = 1;
  // ---- composite keys, residual predicates, outer join -------------------------------------------------------------
  using JT = HashJoinOperator::JoinType;
  Predicate dim_long_lt_15;   // residual of CompositeKeyHashJoinWithResidualPredicateTest (:1268-1272): dim.long < 15
  dim_long_lt_15.conjuncts.push_back(ComparisonPredicate(0, ComparisonID::kLess, TypedLiteral::Long(15), /*build_side=*/true));
  const std::vector<std::vector<attribute_id>> composite_keys = {{0, 2} /* (LONG, LONG): hashed fold + component check */,
                                                                 {4, 3} /* (INT, INT): exact 8-byte packing */};
  for (const int variant : {0, 1, 2, 3, 4, 5, 6, 7}) {
    const bool use_foreman = (variant & 1) != 0;
    g_partitioning = variant >= 6 ? 0 : variant / 2;
    g_blocks_per_work_order = variant >= 6 ? 7 : 1;   // 6, 7: run work orders (these joins fall back to block by block inside them)
    for (const auto &keys : composite_keys) {
      {  // CompositeKeyHashJoinTest: 100 results, the even tids below 200, each once on both sides (:1159-1177)
        JoinedRows r = runGeneralJoin(keys, kLong, nullptr, JT::kInnerJoin, use_foreman);
        EXPECT_EQ(r.dim_long.size(), static_cast<std::size_t>(100));
        std::vector<int> seen(kNumDimTuples, 0);
        for (std::size_t i = 0; i < r.dim_long.size(); ++i) {
          EXPECT_EQ(r.dim_long[i], r.fact_long[i]);
          EXPECT_TRUE(r.dim_long[i] >= 0 && r.dim_long[i] < kNumDimTuples && (r.dim_long[i] & 1) == 0);
          if (r.dim_long[i] >= 0 && r.dim_long[i] < kNumDimTuples) ++seen[r.dim_long[i]];
          EXPECT_TRUE(!r.dim_is_null[i]);
        }
        for (tuple_id i = 0; i < kNumDimTuples; ++i) EXPECT_EQ(seen[i], (i & 1) ? 0 : 1);
      }
      {  // ... WithResidualPredicateTest: 8 results, tids 0, 2, ..., 14 (:1350-1368)
        JoinedRows r = runGeneralJoin(keys, kLong, &dim_long_lt_15, JT::kInnerJoin, use_foreman);
        EXPECT_EQ(r.dim_long.size(), static_cast<std::size_t>(8));
        std::sort(r.dim_long.begin(), r.dim_long.end());
        std::sort(r.fact_long.begin(), r.fact_long.end());
        for (std::size_t i = 0; i < r.dim_long.size(); ++i) {
          EXPECT_EQ(r.dim_long[i], static_cast<std::int64_t>(2 * i));
          EXPECT_EQ(r.fact_long[i], static_cast<std::int64_t>(2 * i));
        }
      }
      {  // semi / anti with the residual (HashJoinOperator.cpp:680-793, :880-1000): 8 resp. 292 fact rows
        JoinedRows semi = runGeneralJoin(keys, kLong, &dim_long_lt_15, JT::kLeftSemiJoin, use_foreman);
        JoinedRows anti = runGeneralJoin(keys, kLong, &dim_long_lt_15, JT::kLeftAntiJoin, use_foreman);
        EXPECT_EQ(semi.fact_long.size(), static_cast<std::size_t>(8));
        EXPECT_EQ(anti.fact_long.size(), static_cast<std::size_t>(kNumFactTuples - 8));
        std::sort(semi.fact_long.begin(), semi.fact_long.end());
        for (std::size_t i = 0; i < semi.fact_long.size(); ++i) EXPECT_EQ(semi.fact_long[i], static_cast<std::int64_t>(2 * i));
        std::vector<int> seen(kNumFactTuples, 0);
        for (std::int64_t v : anti.fact_long) ++seen[v];
        for (tuple_id i = 0; i < kNumFactTuples; ++i) EXPECT_EQ(seen[i], (i < 15 && (i & 1) == 0) ? 0 : 1);
      }
      {  // left outer join (HashOuterJoinWorkOrder, :960-1099): 100 matched rows + 200 fact rows with NULL dim.long
        JoinedRows r = runGeneralJoin(keys, kLong, nullptr, JT::kLeftOuterJoin, use_foreman);
        EXPECT_EQ(r.fact_long.size(), static_cast<std::size_t>(kNumFactTuples));
        std::vector<int> seen(kNumFactTuples, 0);
        for (std::size_t i = 0; i < r.fact_long.size(); ++i) {
          const std::int64_t fl = r.fact_long[i];
          ++seen[fl];
          const bool matched = fl < kNumDimTuples && (fl & 1) == 0;
          EXPECT_EQ(static_cast<int>(r.dim_is_null[i]), matched ? 0 : 1);
          if (matched) EXPECT_EQ(r.dim_long[i], fl);
        }
        for (tuple_id i = 0; i < kNumFactTuples; ++i) EXPECT_EQ(seen[i], 1);
      }
    }
    {  // attribute-vs-attribute residual on the LONG key join: fact.int == dim.int holds for tids 0..9 only
      Predicate same_int;
      same_int.conjuncts.push_back(ComparisonPredicate::Attributes(1, /*lhs_on_build=*/false, ComparisonID::kEqual, 1, /*rhs_on_build=*/true));
      JoinedRows r = runGeneralJoin({0}, kLong, &same_int, JT::kInnerJoin, use_foreman);
      EXPECT_EQ(r.dim_long.size(), static_cast<std::size_t>(kBlockSize));
      std::sort(r.dim_long.begin(), r.dim_long.end());
      for (std::size_t i = 0; i < r.dim_long.size(); ++i) EXPECT_EQ(r.dim_long[i], static_cast<std::int64_t>(i));
    }
  }
  return finish("hash_join_operator_test");
}
