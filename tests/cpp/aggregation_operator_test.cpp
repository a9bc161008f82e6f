// Mirrors relational_operators/tests/AggregationOperator_unittest.cpp: 300 rows in 10-tuple blocks,
// columns GroupBy-0/1, IntType, LongType, FloatType, DoubleType (:192-207); scalar SUM/AVG/COUNT with
// and without predicate (:593-602, :877-886), zero-row behaviour (:1160-1345) and the GROUP BY checks
// (:1349-1470), driven synchronously like the reference test and through Foreman/Worker with a
// Select -> Aggregation streaming edge.  GPU work orders.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <tuple>

#include "test_util.hpp"

using namespace quickstep;

namespace {
constexpr tuple_id kNumTuples = 300;
constexpr int kGroupByWidth = 20;
constexpr int kGroupByRepeats = kNumTuples / kGroupByWidth;
constexpr int kGroupBy1Size = 4;
constexpr tuple_id kNumTuplesPerBlock = 10;

std::int64_t Summation(int n) { return static_cast<std::int64_t>(n) * (n + 1) / 2; }
std::int64_t ArithmeticSum(int a, int d, int n) { return static_cast<std::int64_t>(n) * (2 * a + (n - 1) * d) / 2; }

struct Fixture {
  CatalogRelation table{100, "aggregate_test"};
  StorageManager storage;
  Fixture() {
    table.addAttribute("GroupBy-0", Type::Int());
    table.addAttribute("GroupBy-1", Type::Int());
    table.addAttribute("IntType-0", Type::Int());
    table.addAttribute("LongType-0", Type::Long());
    table.addAttribute("FloatType-0", Type::Float());
    table.addAttribute("DoubleType-0", Type::Double());
    for (tuple_id i = 0; i < kNumTuples; i += kNumTuplesPerBlock) {
      std::int32_t g0[kNumTuplesPerBlock], g1[kNumTuplesPerBlock], iv[kNumTuplesPerBlock];
      std::int64_t lv[kNumTuplesPerBlock];
      float fv[kNumTuplesPerBlock];
      double dv[kNumTuplesPerBlock];
      for (tuple_id t = 0; t < kNumTuplesPerBlock; ++t) {
        const int val = i + t, gid = val % kGroupByWidth;
        g0[t] = gid % kGroupBy1Size; g1[t] = gid / kGroupBy1Size; iv[t] = val; lv[t] = val;
        fv[t] = static_cast<float>(0.1 * val); dv[t] = 0.1 * val;
      }
      storage.loadBlock(&table, {g0, g1, iv, lv, fv, dv}, kNumTuplesPerBlock);
    }
  }
};

struct Rows {
  std::vector<std::vector<std::int64_t>> ints;   // per output attribute (raw 8-byte words)
  std::size_t n = 0;
};

// Output attributes are 4-byte keys or 8-byte values; read everything as raw words.
std::vector<std::vector<unsigned char>> readAll(QueryContext &ctx, QueryContext::insert_destination_id dest,
                                                StorageManager &storage, const CatalogRelation &rel, std::size_t *rows) {
  std::vector<std::vector<unsigned char>> cols(rel.size());
  *rows = 0;
  for (block_id b : ctx.getInsertDestination(dest)->getTouchedBlocks()) {
    BlockReference blk = storage.getBlock(b);
    for (std::size_t a = 0; a < rel.size(); ++a) {
      const std::size_t w = rel.getAttributeType(static_cast<attribute_id>(a)).width;
      const std::size_t at = cols[a].size();
      cols[a].resize(at + w * blk->numTuples());
      blk->copyAttributeToHost(static_cast<attribute_id>(a), cols[a].data() + at);
    }
    *rows += static_cast<std::size_t>(blk->numTuples());
  }
  return cols;
}
template <typename T>
T at(const std::vector<unsigned char> &col, std::size_t i) {
  T v;
  std::memcpy(&v, col.data() + i * sizeof(T), sizeof(T));
  return v;
}

// ---- a group-by with many groups through the operators ------------------------------------------------------------------------
// 24 M rows in 1 M-row blocks, 500 000 random groups, COUNT / SUM / MIN / MAX — AggregationWorkOrders over runs of 12 blocks on four
// Workers: each run is laid end to end and takes the two partition passes and the per-piece LDS tables (csrc/agg_pieces.hpp);
// with a predicate on a plain attribute the state filters inside its kernels after one partition pass.  Against the same
// aggregation computed on the host.
extern "C" long long qsx_debug_agg_run_concats(void);
extern "C" long long qsx_debug_agg_two_level_updates(void);
void testManyGroupsThroughRuns() {
  constexpr std::int64_t kRows = 24000000, kBlock = 1000000;
  constexpr std::int32_t kGroups = 500000;
  std::vector<std::int32_t> key(kRows);
  std::vector<double> val(kRows);
  std::vector<std::int64_t> qty(kRows);
  std::uint64_t x = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (std::int64_t i = 0; i < kRows; ++i) {
    key[i] = static_cast<std::int32_t>(rnd() % kGroups) * 3 - 7;
    val[i] = static_cast<double>(static_cast<std::int64_t>(rnd() % 200001) - 100000) / 8.0;   // multiples of 1/8: sums exact in any order
    qty[i] = static_cast<std::int64_t>(rnd() % 1000003) - 500000;
  }
  for (const int with_predicate : {0, 1}) {
    struct G { std::int64_t count = 0; double sum = 0; std::int64_t min_q = INT64_MAX; double max_v = -1e300; };
    std::vector<G> want(kGroups);
    for (std::int64_t i = 0; i < kRows; ++i) {
      if (with_predicate && !(qty[i] < 250000)) continue;
      G &g = want[(key[i] + 7) / 3];
      ++g.count; g.sum += val[i]; g.min_q = std::min(g.min_q, qty[i]); g.max_v = std::max(g.max_v, val[i]);
    }
    CatalogRelation rel(200, "many_groups"), result(201, "result");
    StorageManager storage;
    rel.addAttribute("k", Type::Int());
    rel.addAttribute("v", Type::Double());
    rel.addAttribute("q", Type::Long());
    for (std::int64_t at = 0; at < kRows; at += kBlock) storage.loadBlock(&rel, {key.data() + at, val.data() + at, qty.data() + at}, kBlock);
    result.addAttribute("k", Type::Int());
    result.addAttribute("count", Type::Long());
    result.addAttribute("sum", Type::Double().getNullableVersion());
    result.addAttribute("min_q", Type::Long().getNullableVersion());
    result.addAttribute("max_v", Type::Double().getNullableVersion());
    QueryContext ctx;
    Predicate pred;
    pred.conjuncts.push_back({2, ComparisonID::kLess, TypedLiteral::Long(250000)});
    const auto pred_id = ctx.addPredicate(pred);
    AggregationStateSpec spec;
    spec.input_relation = &rel;
    spec.group_by = {0};
    spec.aggregates = {{AggregationID::kCount, kInvalidAttributeID}, {AggregationID::kSum, 1}, {AggregationID::kMin, 2}, {AggregationID::kMax, 1}};
    spec.predicate = with_predicate ? ctx.getPredicate(pred_id) : nullptr;
    spec.strategy = QSX_AGG_GENERIC;
    spec.estimated_num_groups = kGroups;
    const auto state = ctx.addAggregationState(spec);
    const auto dest = ctx.addInsertDestination(&result, &storage);
    auto *aggregate = new AggregationOperator(0, rel, true, state);
    auto *finalize = new FinalizeAggregationOperator(0, state, 1, false, 1, result, dest);
    aggregate->setBlocksPerWorkOrder(12);
    const long long concats = qsx_debug_agg_run_concats(), two_level = qsx_debug_agg_two_level_updates();
    QueryPlan plan;
    const auto a = plan.addRelationalOperator(aggregate);
    const auto z = plan.addRelationalOperator(finalize);
    plan.addDirectDependency(z, a, true);
    ForemanSingleNode foreman(&plan, &ctx, &storage, 4);
    foreman.run();
    if (!with_predicate) {
      EXPECT_EQ(qsx_debug_agg_run_concats() - concats, 2ll);          // two work orders, each a run laid end to end
      EXPECT_EQ(qsx_debug_agg_two_level_updates() - two_level, 2ll);
    }
    std::size_t rows = 0;
    auto cols = readAll(ctx, dest, storage, result, &rows);
    std::size_t present = 0;
    for (const G &g : want) present += g.count != 0 ? 1 : 0;
    EXPECT_EQ(rows, present);
    std::size_t wrong = 0;
    for (std::size_t i = 0; i < rows; ++i) {
      const std::int32_t k = at<std::int32_t>(cols[0], i);
      const G &g = want[static_cast<std::size_t>((k + 7) / 3)];
      if ((k + 7) % 3 != 0 || at<std::int64_t>(cols[1], i) != g.count || at<double>(cols[2], i) != g.sum || at<std::int64_t>(cols[3], i) != g.min_q ||
          at<double>(cols[4], i) != g.max_v) {
        ++wrong;
      }
    }
    EXPECT_EQ(wrong, static_cast<std::size_t>(0));
  }
}
}  // namespace

#include <cstring>

int main() {
  if (qsx_device_count() < 1) {
    std::fprintf(stderr, "aggregation_operator_test needs an MI355X: %s\n", qsx_status_string(QSX_ERR_NO_DEVICE));
    return 2;
  }
  // ---- scalar aggregates, synchronous driver ------------------------------------------------------
  for (const int predicate_value : {-2, 30, -1}) {  // -2: no predicate; 30: IntType-0 < 30; -1: zero rows
    Fixture f;
    CatalogRelation result(101, "result");
    result.addAttribute("sum_int", Type::Long());
    result.addAttribute("sum_double", Type::Double());
    result.addAttribute("avg_long", Type::Double());
    result.addAttribute("count", Type::Long());
    QueryContext ctx;
    Predicate pred;
    pred.conjuncts.push_back({2, ComparisonID::kLess, TypedLiteral::Int(predicate_value)});
    const auto pred_id = ctx.addPredicate(pred);
    AggregationStateSpec spec;
    spec.input_relation = &f.table;
    spec.aggregates = {{AggregationID::kSum, 2}, {AggregationID::kSum, 5}, {AggregationID::kAvg, 3}, {AggregationID::kCount, kInvalidAttributeID}};
    spec.predicate = predicate_value == -2 ? nullptr : ctx.getPredicate(pred_id);
    const auto state = ctx.addAggregationState(spec);
    const auto dest = ctx.addInsertDestination(&result, &f.storage);
    AggregationOperator op(0, f.table, true, state);
    FinalizeAggregationOperator fin(0, state, 1, false, 1, result, dest);
    DestroyAggregationStateOperator destroy(0, state);
    fetchAndExecuteWorkOrders(&op, &ctx, &f.storage);
    fetchAndExecuteWorkOrders(&fin, &ctx, &f.storage);
    std::size_t rows;
    auto cols = readAll(ctx, dest, f.storage, result, &rows);
    EXPECT_EQ(rows, static_cast<std::size_t>(1));  // exactly one row, even for zero input rows (:1160-1345)
    const int last = predicate_value == -2 ? 299 : predicate_value - 1;
    const std::int64_t count = last + 1 > 0 ? last + 1 : 0;  // "< -1" selects nothing
    EXPECT_EQ(at<std::int64_t>(cols[3], 0), count);
    if (count > 0) {
      EXPECT_EQ(at<std::int64_t>(cols[0], 0), Summation(last));            // SUM(int) is LONG, exact
      EXPECT_NEAR(at<double>(cols[1], 0), 0.1 * Summation(last), 1e-5 * 0.1 * Summation(last));
      EXPECT_NEAR(at<double>(cols[2], 0), Summation(last) / static_cast<double>(count), 1e-9);
    }
    fetchAndExecuteWorkOrders(&destroy, &ctx, &f.storage);
  }

  // ---- GROUP BY, both drivers, both hash strategies -------------------------------------------------
  for (const bool use_foreman : {false, true}) {
    for (const qsx_agg_strategy_t strategy : {QSX_AGG_COMPACT_KEY, QSX_AGG_GENERIC}) {
      for (const bool with_predicate : {false, true}) {
        Fixture f;
        CatalogRelation selected(102, "selected"), result(103, "result");
        result.addAttribute("GroupBy-0", Type::Int());
        result.addAttribute("GroupBy-1", Type::Int());
        result.addAttribute("sum_int", Type::Long());
        result.addAttribute("sum_double", Type::Double());
        result.addAttribute("avg_int", Type::Double());
        result.addAttribute("count", Type::Long());
        QueryContext ctx;
        Predicate pred;
        pred.conjuncts.push_back({2, ComparisonID::kLess, TypedLiteral::Int(kGroupByWidth * (kGroupByRepeats >> 1))});  // :539
        const auto pred_id = ctx.addPredicate(pred);
        AggregationStateSpec spec;
        spec.input_relation = &f.table;
        spec.group_by = {0, 1};
        spec.aggregates = {{AggregationID::kSum, 2}, {AggregationID::kSum, 5}, {AggregationID::kAvg, 2}, {AggregationID::kCount, kInvalidAttributeID}};
        spec.predicate = with_predicate ? ctx.getPredicate(pred_id) : nullptr;
        spec.strategy = strategy;
        spec.estimated_num_groups = kGroupByWidth;
        const auto dest = ctx.addInsertDestination(&result, &f.storage);
        if (use_foreman) {
          // Select(all columns, no predicate) --streaming--> Aggregation --breaker--> Finalize --breaker--> Destroy
          for (const char *name : {"GroupBy-0", "GroupBy-1", "IntType-0", "LongType-0", "FloatType-0", "DoubleType-0"}) {
            selected.addAttribute(name, f.table.getAttributeType(f.table.getAttributeByName(name)));
          }
          spec.input_relation = &selected;
          const auto state = ctx.addAggregationState(spec);
          const auto sel_dest = ctx.addInsertDestination(&selected, &f.storage);
          QueryPlan plan;
          const auto s = plan.addRelationalOperator(new SelectOperator(0, f.table, false, selected, sel_dest,
                                                                       QueryContext::kInvalidPredicateId,
                                                                       std::vector<attribute_id>{0, 1, 2, 3, 4, 5}, true));
          const auto a = plan.addRelationalOperator(new AggregationOperator(0, selected, false, state));
          const auto fz = plan.addRelationalOperator(new FinalizeAggregationOperator(0, state, 1, false, 1, result, dest));
          const auto d = plan.addRelationalOperator(new DestroyAggregationStateOperator(0, state));
          plan.addDirectDependency(a, s, false);
          plan.addDirectDependency(fz, a, true);
          plan.addDirectDependency(d, fz, true);
          ForemanSingleNode foreman(&plan, &ctx, &f.storage, 4);
          foreman.run();
          EXPECT_EQ(foreman.getWorkOrderProfilingResults().size(), static_cast<std::size_t>(30 + 30 + 1 + 1));
        } else {
          const auto state = ctx.addAggregationState(spec);
          AggregationOperator op(0, f.table, true, state);
          FinalizeAggregationOperator fin(0, state, 1, false, 1, result, dest);
          fetchAndExecuteWorkOrders(&op, &ctx, &f.storage);
          fetchAndExecuteWorkOrders(&fin, &ctx, &f.storage);
        }
        std::size_t rows;
        auto cols = readAll(ctx, dest, f.storage, result, &rows);
        EXPECT_EQ(rows, static_cast<std::size_t>(kGroupByWidth));
        const int repeats = with_predicate ? kGroupByRepeats >> 1 : kGroupByRepeats;
        std::vector<bool> seen(kGroupByWidth, false);
        for (std::size_t i = 0; i < rows; ++i) {
          const int gid = at<std::int32_t>(cols[0], i) + at<std::int32_t>(cols[1], i) * kGroupBy1Size;  // :517
          EXPECT_TRUE(gid >= 0 && gid < kGroupByWidth && !seen[gid]);
          seen[gid] = true;
          const std::int64_t sum = ArithmeticSum(gid, kGroupByWidth, repeats);
          EXPECT_EQ(at<std::int64_t>(cols[2], i), sum);
          EXPECT_NEAR(at<double>(cols[3], i), 0.1 * sum, 1e-5 * 0.1 * sum + 1e-12);
          EXPECT_NEAR(at<double>(cols[4], i), sum / static_cast<double>(repeats), 1e-5 * sum / repeats + 1e-12);
          EXPECT_EQ(at<std::int64_t>(cols[5], i), static_cast<std::int64_t>(repeats));
        }
      }
    }
  }
  // ---- MIN / MAX (GroupBy_{Max,Min}_*: :1550-1680; ScalarAttribute_*_{Max,Min}_*: :675-712) ---------------------------
  for (const bool with_predicate : {false, true}) {
    Fixture f;
    CatalogRelation result(104, "result");
    result.addAttribute("GroupBy-0", Type::Int());
    result.addAttribute("GroupBy-1", Type::Int());
    result.addAttribute("max_int", Type::Int());       // MIN / MAX keep the argument's type
    result.addAttribute("min_long", Type::Long());
    result.addAttribute("max_float", Type::Float());
    result.addAttribute("min_double", Type::Double());
    QueryContext ctx;
    Predicate pred;
    pred.conjuncts.push_back({2, ComparisonID::kLess, TypedLiteral::Int(kGroupByWidth * (kGroupByRepeats >> 1))});
    const auto pred_id = ctx.addPredicate(pred);
    AggregationStateSpec spec;
    spec.input_relation = &f.table;
    spec.group_by = {0, 1};
    spec.aggregates = {{AggregationID::kMax, 2}, {AggregationID::kMin, 3}, {AggregationID::kMax, 4}, {AggregationID::kMin, 5}};
    spec.predicate = with_predicate ? ctx.getPredicate(pred_id) : nullptr;
    spec.estimated_num_groups = kGroupByWidth;
    const auto state = ctx.addAggregationState(spec);
    const auto dest = ctx.addInsertDestination(&result, &f.storage);
    AggregationOperator op(0, f.table, true, state);
    FinalizeAggregationOperator fin(0, state, 1, false, 1, result, dest);
    fetchAndExecuteWorkOrders(&op, &ctx, &f.storage);
    fetchAndExecuteWorkOrders(&fin, &ctx, &f.storage);
    std::size_t rows;
    auto cols = readAll(ctx, dest, f.storage, result, &rows);
    EXPECT_EQ(rows, static_cast<std::size_t>(kGroupByWidth));
    const int repeats = with_predicate ? kGroupByRepeats >> 1 : kGroupByRepeats;
    for (std::size_t i = 0; i < rows; ++i) {
      const int gid = at<std::int32_t>(cols[0], i) + at<std::int32_t>(cols[1], i) * kGroupBy1Size;
      const int max = kGroupByWidth * (repeats - 1) + gid;                  // :1555
      EXPECT_EQ(at<std::int32_t>(cols[2], i), max);
      EXPECT_EQ(at<std::int64_t>(cols[3], i), static_cast<std::int64_t>(gid));  // :1648
      EXPECT_TRUE(at<float>(cols[4], i) == static_cast<float>(0.1 * max));
      EXPECT_TRUE(at<double>(cols[5], i) == 0.1 * gid);
    }
  }
  // ---- CrossReferenceCoalesceAggregate (ExecutionGenerator.cpp:2054-2210): customer LEFT OUTER JOIN orders GROUP BY
  // c_custkey fused into  InitializeAggregation -> BuildAggregationExistenceMap(customer.c_custkey) -> Aggregation(orders
  // GROUP BY o_custkey, collision-free vector) -> Finalize.  Customers without orders come out with COUNT 0 / SUM 0
  // (CollisionFreeVectorTable.hpp:700-727), the shape of TPC-H Q13.
  for (const bool use_foreman : {false, true}) {
    constexpr int kCustomers = 50, kWithOrders = 40;
    Fixture f;   // "orders": IntType-0 = val; o_custkey := GroupBy-0 replaced below by a dedicated relation
    CatalogRelation customer(110, "customer"), orders(111, "orders"), result(112, "result");
    customer.addAttribute("c_custkey", Type::Int());
    orders.addAttribute("o_custkey", Type::Int());
    orders.addAttribute("o_totalprice", Type::Long());
    for (int b = 0; b < kCustomers; b += kNumTuplesPerBlock) {
      std::int32_t k[kNumTuplesPerBlock];
      for (tuple_id t = 0; t < kNumTuplesPerBlock; ++t) k[t] = kCustomers - 1 - (b + t);   // not sorted
      f.storage.loadBlock(&customer, {k}, kNumTuplesPerBlock);
    }
    std::vector<std::int64_t> want_sum(kCustomers, 0), want_count(kCustomers, 0);
    for (tuple_id i = 0; i < kNumTuples; i += kNumTuplesPerBlock) {
      std::int32_t k[kNumTuplesPerBlock];
      std::int64_t v[kNumTuplesPerBlock];
      for (tuple_id t = 0; t < kNumTuplesPerBlock; ++t) {
        k[t] = ((i + t) * 7) % kWithOrders;
        v[t] = i + t;
        if (v[t] >= 20) { want_sum[k[t]] += v[t]; ++want_count[k[t]]; }
      }
      f.storage.loadBlock(&orders, {k, v}, kNumTuplesPerBlock);
    }
    result.addAttribute("c_custkey", Type::Int());
    result.addAttribute("count", Type::Long());
    result.addAttribute("sum", Type::Long());
    QueryContext ctx;
    Predicate pred;   // the right child's filter predicate is fused into the state (:2096-2100)
    pred.conjuncts.push_back({1, ComparisonID::kGreaterOrEqual, TypedLiteral::Long(20)});
    const auto pred_id = ctx.addPredicate(pred);
    AggregationStateSpec spec;
    spec.input_relation = &orders;
    spec.group_by = {0};
    spec.aggregates = {{AggregationID::kCount, kInvalidAttributeID}, {AggregationID::kSum, 1}};
    spec.predicate = ctx.getPredicate(pred_id);
    spec.strategy = QSX_AGG_COLLISION_FREE;
    spec.collision_free_num_entries = kCustomers;
    const auto state = ctx.addAggregationState(spec);
    const auto dest = ctx.addInsertDestination(&result, &f.storage);
    if (use_foreman) {
      QueryPlan plan;
      // InitializeAggregation in front, in 3 slices ("aggr_state_num_init_partitions", ExecutionGenerator.cpp:204-208, 2160-2170)
      const auto in = plan.addRelationalOperator(new InitializeAggregationOperator(0, state, 1, 3));
      const auto e = plan.addRelationalOperator(new BuildAggregationExistenceMapOperator(0, customer, 0, true, state));
      plan.addDirectDependency(e, in, true);
      const auto a = plan.addRelationalOperator(new AggregationOperator(0, orders, true, state));
      const auto fz = plan.addRelationalOperator(new FinalizeAggregationOperator(0, state, 1, false, 2, result, dest));
      const auto d = plan.addRelationalOperator(new DestroyAggregationStateOperator(0, state));
      plan.addDirectDependency(a, e, true);    // "Start aggregation after building existence map" (:2176-2179)
      plan.addDirectDependency(fz, a, true);
      plan.addDirectDependency(d, fz, true);
      ForemanSingleNode foreman(&plan, &ctx, &f.storage, 4);
      foreman.run();
      EXPECT_EQ(foreman.getWorkOrderProfilingResults().size(), static_cast<std::size_t>(3 + 5 + 30 + 2 + 1));
    } else {
      InitializeAggregationOperator init(0, state, 1, 2);
      EXPECT_TRUE(init.getOperatorType() == RelationalOperator::kInitializeAggregation);
      EXPECT_TRUE(init.getName() == "InitializeAggregationOperator");
      fetchAndExecuteWorkOrders(&init, &ctx, &f.storage);
      BuildAggregationExistenceMapOperator exist(0, customer, 0, true, state);
      EXPECT_TRUE(exist.getOperatorType() == RelationalOperator::kBuildAggregationExistenceMap);
      AggregationOperator op(0, orders, true, state);
      FinalizeAggregationOperator fin(0, state, 1, false, 1, result, dest);
      fetchAndExecuteWorkOrders(&exist, &ctx, &f.storage);
      fetchAndExecuteWorkOrders(&op, &ctx, &f.storage);
      fetchAndExecuteWorkOrders(&fin, &ctx, &f.storage);
    }
    std::size_t rows;
    auto cols = readAll(ctx, dest, f.storage, result, &rows);
    EXPECT_EQ(rows, static_cast<std::size_t>(kCustomers));    // every customer, with or without orders
    std::vector<bool> seen(kCustomers, false);
    for (std::size_t i = 0; i < rows; ++i) {
      const int key = at<std::int32_t>(cols[0], i);
      EXPECT_TRUE(key >= 0 && key < kCustomers && !seen[key]);
      seen[key] = true;
      EXPECT_EQ(at<std::int64_t>(cols[1], i), want_count[key]);
      EXPECT_EQ(at<std::int64_t>(cols[2], i), want_sum[key]);
    }
  }
  // ---- InitializeAggregation on a state that is not a collision-free vector: the reference's LOG(FATAL)
  // (AggregationOperationState.cpp:418-426) is an ExecutionError here; re-initialising a USED collision-free state empties it
  {
    Fixture f;
    CatalogRelation keys(113, "keys"), result(114, "result");
    keys.addAttribute("k", Type::Int());
    std::int32_t k[kNumTuplesPerBlock];
    for (tuple_id t = 0; t < kNumTuplesPerBlock; ++t) k[t] = t % 7;
    f.storage.loadBlock(&keys, {k}, kNumTuplesPerBlock);
    result.addAttribute("k", Type::Int());
    result.addAttribute("count", Type::Long());
    QueryContext ctx;
    AggregationStateSpec spec;
    spec.input_relation = &keys;
    spec.group_by = {0};
    spec.aggregates = {{AggregationID::kCount, kInvalidAttributeID}};
    const auto hashed = ctx.addAggregationState(spec);
    InitializeAggregationOperator bad(0, hashed, 1, 1);
    bool threw = false;
    try {
      fetchAndExecuteWorkOrders(&bad, &ctx, &f.storage);
    } catch (const ExecutionError &e) {
      threw = e.status() == QSX_ERR_UNSUPPORTED;
    }
    EXPECT_TRUE(threw);
    spec.strategy = QSX_AGG_COLLISION_FREE;
    spec.collision_free_num_entries = 7;
    const auto dense = ctx.addAggregationState(spec);
    const auto dest = ctx.addInsertDestination(&result, &f.storage);
    AggregationOperator once(0, keys, true, dense), again(0, keys, true, dense);
    fetchAndExecuteWorkOrders(&once, &ctx, &f.storage);
    InitializeAggregationOperator init(0, dense, 1, 4);
    fetchAndExecuteWorkOrders(&init, &ctx, &f.storage);      // what the first pass counted is gone
    fetchAndExecuteWorkOrders(&again, &ctx, &f.storage);
    FinalizeAggregationOperator fin(0, dense, 1, false, 1, result, dest);
    fetchAndExecuteWorkOrders(&fin, &ctx, &f.storage);
    std::size_t rows;
    auto cols = readAll(ctx, dest, f.storage, result, &rows);
    EXPECT_EQ(rows, static_cast<std::size_t>(7));
    for (std::size_t i = 0; i < rows; ++i) {
      const int key = at<std::int32_t>(cols[0], i);
      std::int64_t want = 0;
      for (tuple_id t = 0; t < kNumTuplesPerBlock; ++t) want += (t % 7) == key;
      EXPECT_EQ(at<std::int64_t>(cols[1], i), want);
    }
  }
  // ---- DISTINCT aggregates: query_optimizer/tests/execution_generator/Distinct.test -----------------------------------
  // foo(x INT, y DOUBLE, z INT) = (i, (i + 0.5) % 100, i % 3) for i in 0..29999 (:18-24); w = x % y stands in for the
  // scalar argument of the third query.  30 blocks of 1000 tuples, 4 workers.
  {
    constexpr int kRows = 30000, kBlock = 1000;
    StorageManager storage;
    CatalogRelation foo(120, "foo");
    foo.addAttribute("x", Type::Int());
    foo.addAttribute("y", Type::Double());
    foo.addAttribute("z", Type::Int());
    foo.addAttribute("w", Type::Double());
    for (int b = 0; b < kRows; b += kBlock) {
      std::vector<std::int32_t> x(kBlock), z(kBlock);
      std::vector<double> y(kBlock), w(kBlock);
      for (int t = 0; t < kBlock; ++t) {
        const int i = b + t;
        x[t] = i; y[t] = std::fmod(i + 0.5, 100.0); z[t] = i % 3; w[t] = std::fmod(static_cast<double>(i), y[t]);
      }
      storage.loadBlock(&foo, {x.data(), y.data(), z.data(), w.data()}, kBlock);
    }
    auto distinct = [](AggregationID fn, attribute_id arg) { AggregateSpec a{fn, arg}; a.is_distinct = true; return a; };
    {  // SELECT COUNT(*), COUNT(DISTINCT x), COUNT(DISTINCT y), COUNT(DISTINCT z) FROM foo  (:27-37)
      CatalogRelation result(121, "result");
      for (const char *name : {"count", "dx", "dy", "dz"}) result.addAttribute(name, Type::Long());
      QueryContext ctx;
      AggregationStateSpec spec;
      spec.input_relation = &foo;
      spec.aggregates = {{AggregationID::kCount, kInvalidAttributeID}, distinct(AggregationID::kCount, 0),
                         distinct(AggregationID::kCount, 1), distinct(AggregationID::kCount, 2)};
      const auto state = ctx.addAggregationState(spec);
      const auto dest = ctx.addInsertDestination(&result, &storage);
      QueryPlan plan;
      const auto a = plan.addRelationalOperator(new AggregationOperator(0, foo, true, state));
      const auto fz = plan.addRelationalOperator(new FinalizeAggregationOperator(0, state, 1, false, 1, result, dest));
      plan.addDirectDependency(fz, a, true);
      ForemanSingleNode foreman(&plan, &ctx, &storage, 4);
      foreman.run();
      std::size_t rows;
      auto cols = readAll(ctx, dest, storage, result, &rows);
      EXPECT_EQ(rows, static_cast<std::size_t>(1));
      EXPECT_EQ(at<std::int64_t>(cols[0], 0), static_cast<std::int64_t>(30000));
      EXPECT_EQ(at<std::int64_t>(cols[1], 0), static_cast<std::int64_t>(30000));
      EXPECT_EQ(at<std::int64_t>(cols[2], 0), static_cast<std::int64_t>(100));
      EXPECT_EQ(at<std::int64_t>(cols[3], 0), static_cast<std::int64_t>(3));
    }
    for (const qsx_agg_strategy_t strategy : {QSX_AGG_COMPACT_KEY, QSX_AGG_GENERIC, QSX_AGG_COLLISION_FREE}) {
      // SELECT SUM(y), SUM(DISTINCT y), COUNT(DISTINCT y), AVG(DISTINCT y), z FROM foo GROUP BY z ORDER BY z  (:40-56)
      // SELECT MAX(x) * SUM(DISTINCT y), COUNT(DISTINCT x % y) + z, z FROM foo GROUP BY z ORDER BY z        (:58-72)
      CatalogRelation result(122, "result");
      result.addAttribute("z", Type::Int());
      result.addAttribute("sum_y", Type::Double());
      result.addAttribute("sum_distinct_y", Type::Double());
      result.addAttribute("count_distinct_y", Type::Long());
      result.addAttribute("avg_distinct_y", Type::Double());
      result.addAttribute("max_x", Type::Int());
      result.addAttribute("count_distinct_w", Type::Long());
      QueryContext ctx;
      AggregationStateSpec spec;
      spec.input_relation = &foo;
      spec.group_by = {2};
      spec.aggregates = {{AggregationID::kSum, 1}, distinct(AggregationID::kSum, 1), distinct(AggregationID::kCount, 1),
                         distinct(AggregationID::kAvg, 1), {AggregationID::kMax, 0}, distinct(AggregationID::kCount, 3)};
      spec.strategy = strategy;
      spec.estimated_num_groups = 3;
      spec.collision_free_num_entries = 3;
      const auto state = ctx.addAggregationState(spec);
      const auto dest = ctx.addInsertDestination(&result, &storage);
      AggregationOperator op(0, foo, true, state);
      FinalizeAggregationOperator fin(0, state, 1, false, 2, result, dest);   // two finalize work orders, one emits
      fetchAndExecuteWorkOrders(&op, &ctx, &storage);
      fetchAndExecuteWorkOrders(&fin, &ctx, &storage);
      std::size_t rows;
      auto cols = readAll(ctx, dest, storage, result, &rows);
      EXPECT_EQ(rows, static_cast<std::size_t>(3));
      const std::int64_t want_product[3] = {149985000, 149990000, 149995000};
      const std::int64_t want_count_plus_z[3] = {196, 197, 195};
      for (std::size_t i = 0; i < rows && i < 3; ++i) {
        const int z = at<std::int32_t>(cols[0], i);
        EXPECT_EQ(z, static_cast<int>(i));                                     // groups come out in key order
        EXPECT_NEAR(at<double>(cols[1], i), 500000.0, 1e-6 * 500000.0);
        EXPECT_TRUE(at<double>(cols[2], i) == 5000.0);                         // 0.5 + 1.5 + ... + 99.5, exact in double
        EXPECT_EQ(at<std::int64_t>(cols[3], i), static_cast<std::int64_t>(100));
        EXPECT_TRUE(at<double>(cols[4], i) == 50.0);
        EXPECT_EQ(static_cast<std::int64_t>(at<std::int32_t>(cols[5], i) * at<double>(cols[2], i)), want_product[z]);
        EXPECT_EQ(at<std::int64_t>(cols[6], i) + z, want_count_plus_z[z]);
      }
    }
    {  // DISTINCT over a scalar expression argument (the shape of Distinct.test:58-72's COUNT(DISTINCT x % y)): the distinctify
       // key is (z, value); y + y takes the 100 values 1, 3, .. 199 in every group, x - x is 0 everywhere, and a plain
       // SUM(y * 2) runs next to them in the state's own expression program
      CatalogRelation result(124, "result");
      result.addAttribute("z", Type::Int());
      result.addAttribute("count_distinct_2y", Type::Long());
      result.addAttribute("sum_distinct_2y", Type::Double());
      result.addAttribute("count_distinct_zero", Type::Long());
      result.addAttribute("sum_2y", Type::Double());
      QueryContext ctx;
      const ScalarPtr y = Scalar::Attribute(1), x = Scalar::Attribute(0);
      const ScalarPtr two_y = Scalar::Binary(BinaryOperationID::kAdd, y, y);
      auto distinct_expr = [](AggregationID fn, ScalarPtr e) { AggregateSpec a(fn, std::move(e)); a.is_distinct = true; return a; };
      AggregationStateSpec spec;
      spec.input_relation = &foo;
      spec.group_by = {2};
      spec.aggregates = {distinct_expr(AggregationID::kCount, two_y), distinct_expr(AggregationID::kSum, two_y),
                         distinct_expr(AggregationID::kCount, Scalar::Binary(BinaryOperationID::kSubtract, x, x)),
                         AggregateSpec(AggregationID::kSum, Scalar::Binary(BinaryOperationID::kMultiply, y, Scalar::Literal(2.0)))};
      spec.strategy = QSX_AGG_GENERIC;
      spec.estimated_num_groups = 3;
      const auto state = ctx.addAggregationState(spec);
      const auto dest = ctx.addInsertDestination(&result, &storage);
      AggregationOperator op(0, foo, true, state);
      FinalizeAggregationOperator fin(0, state, 1, false, 1, result, dest);
      fetchAndExecuteWorkOrders(&op, &ctx, &storage);
      fetchAndExecuteWorkOrders(&fin, &ctx, &storage);
      std::size_t rows;
      auto cols = readAll(ctx, dest, storage, result, &rows);
      EXPECT_EQ(rows, static_cast<std::size_t>(3));
      for (std::size_t i = 0; i < rows && i < 3; ++i) {
        EXPECT_EQ(at<std::int32_t>(cols[0], i), static_cast<int>(i));
        EXPECT_EQ(at<std::int64_t>(cols[1], i), static_cast<std::int64_t>(100));
        EXPECT_TRUE(at<double>(cols[2], i) == 10000.0);                       // 1 + 3 + ... + 199
        EXPECT_EQ(at<std::int64_t>(cols[3], i), static_cast<std::int64_t>(1));
        EXPECT_NEAR(at<double>(cols[4], i), 1000000.0, 1e-6 * 1000000.0);     // 10 000 rows x mean 50 x 2
      }
    }
    {  // every aggregate DISTINCT, with the state's predicate: x < 150 leaves y in {0.5 .. 99.5} for 150 rows
      CatalogRelation result(123, "result");
      result.addAttribute("z", Type::Int());
      result.addAttribute("count_distinct_y", Type::Long());
      result.addAttribute("min_distinct_x", Type::Int());
      QueryContext ctx;
      Predicate pred;
      pred.conjuncts.push_back({0, ComparisonID::kLess, TypedLiteral::Int(150)});
      const auto pred_id = ctx.addPredicate(pred);
      AggregationStateSpec spec;
      spec.input_relation = &foo;
      spec.group_by = {2};
      spec.aggregates = {distinct(AggregationID::kCount, 1), distinct(AggregationID::kMin, 0)};
      spec.predicate = ctx.getPredicate(pred_id);
      spec.strategy = QSX_AGG_GENERIC;
      spec.estimated_num_groups = 3;
      const auto state = ctx.addAggregationState(spec);
      const auto dest = ctx.addInsertDestination(&result, &storage);
      AggregationOperator op(0, foo, true, state);
      FinalizeAggregationOperator fin(0, state, 1, false, 1, result, dest);
      fetchAndExecuteWorkOrders(&op, &ctx, &storage);
      fetchAndExecuteWorkOrders(&fin, &ctx, &storage);
      std::size_t rows;
      auto cols = readAll(ctx, dest, storage, result, &rows);
      EXPECT_EQ(rows, static_cast<std::size_t>(3));
      for (std::size_t i = 0; i < rows && i < 3; ++i) {
        // rows 0..149 of residue z: i = z, z+3, ...; y = i + 0.5 for i < 100, i - 99.5 beyond -> 50 rows, values of
        // i mod 100 distinct unless i and i-100 share the residue mod 3 (never: 100 % 3 == 1)
        EXPECT_EQ(at<std::int32_t>(cols[0], i), static_cast<int>(i));
        EXPECT_EQ(at<std::int64_t>(cols[1], i), static_cast<std::int64_t>(50));
        EXPECT_EQ(at<std::int32_t>(cols[2], i), static_cast<int>(i));
      }
    }
  }
  // ---- GROUP BY a key wider than 8 bytes: (GroupBy-0 INT, LongType-0 / 20 LONG, GroupBy-1 INT) = 16 bytes ---------------
  // (the reference's PackedPayloadHashTable takes any composite key; the device groups by a hash of the packed key words
  // and proves the grouping, include/qsx.h QSX_GROUPS_HASH_COLLISION).  Same data as above: val = tid, gid = val % 20,
  // the LONG key val / 20 is the "repeat" number, so every (g0, repeat, g1) is one row and (g0, g1) x repeat = 300 groups.
  for (const bool partitioned_finalize : {false, true}) {
    Fixture f;
    CatalogRelation wide(110, "wide_input"), result(111, "result");
    wide.addAttribute("GroupBy-0", Type::Int());
    wide.addAttribute("repeat", Type::Long());
    wide.addAttribute("GroupBy-1", Type::Int());
    wide.addAttribute("DoubleType-0", Type::Double());
    StorageManager storage;
    for (tuple_id i = 0; i < kNumTuples; i += 50) {
      std::int32_t g0[50], g1[50];
      std::int64_t rep[50];
      double dv[50];
      for (tuple_id t = 0; t < 50; ++t) {
        const int val = i + t, gid = val % kGroupByWidth;
        g0[t] = gid % kGroupBy1Size; g1[t] = gid / kGroupBy1Size;
        rep[t] = (static_cast<std::int64_t>(val / kGroupByWidth) % 5) << 40;   // 5 distinct LONG values beyond 32 bits
        dv[t] = 0.1 * val;
      }
      storage.loadBlock(&wide, {g0, rep, g1, dv}, 50);
    }
    result.addAttribute("GroupBy-0", Type::Int());
    result.addAttribute("repeat", Type::Long());
    result.addAttribute("GroupBy-1", Type::Int());
    result.addAttribute("sum_double", Type::Double());
    result.addAttribute("count", Type::Long());
    QueryContext ctx;
    AggregationStateSpec spec;
    spec.input_relation = &wide;
    spec.group_by = {0, 1, 2};
    spec.aggregates = {{AggregationID::kSum, 3}, {AggregationID::kCount, kInvalidAttributeID}};
    spec.strategy = QSX_AGG_GENERIC;
    spec.estimated_num_groups = 16;     // 100 groups: the table grows
    const auto state = ctx.addAggregationState(spec);
    const auto dest = ctx.addInsertDestination(&result, &storage);
    AggregationOperator op(0, wide, true, state);
    const std::size_t parts = partitioned_finalize ? 3 : 1;
    FinalizeAggregationOperator fin(0, state, 1, false, parts, result, dest);
    fetchAndExecuteWorkOrders(&op, &ctx, &storage);
    fetchAndExecuteWorkOrders(&fin, &ctx, &storage);
    std::size_t rows;
    auto cols = readAll(ctx, dest, storage, result, &rows);
    EXPECT_EQ(rows, static_cast<std::size_t>(kGroupByWidth * 5));
    std::map<std::tuple<int, std::int64_t, int>, std::pair<double, std::int64_t>> want;
    for (int val = 0; val < kNumTuples; ++val) {
      const int gid = val % kGroupByWidth;
      auto &w = want[{gid % kGroupBy1Size, (static_cast<std::int64_t>(val / kGroupByWidth) % 5) << 40, gid / kGroupBy1Size}];
      w.first += 0.1 * val;
      w.second += 1;
    }
    std::size_t matched = 0;
    for (std::size_t i = 0; i < rows; ++i) {
      const auto it = want.find({at<std::int32_t>(cols[0], i), at<std::int64_t>(cols[1], i), at<std::int32_t>(cols[2], i)});
      EXPECT_TRUE(it != want.end());
      if (it == want.end()) continue;
      EXPECT_NEAR(at<double>(cols[3], i), it->second.first, 1e-9 * it->second.first + 1e-12);
      EXPECT_EQ(at<std::int64_t>(cols[4], i), it->second.second);
      ++matched;
    }
    EXPECT_EQ(matched, want.size());
  }
  testManyGroupsThroughRuns();
  return finish("aggregation_operator_test");
}
