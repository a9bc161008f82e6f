// BASELINE configs 4 and 5 through the OPERATOR BOUNDARY, one rank of the multi-GPU plans at its per-rank size: the DAGs
// quickstep_amd/plans.py runs through the raw C ABI (bench.py --config c4 | c5), here as RelationalOperators under
// ForemanSingleNode + Workers with the exchange steps issued by PartitionExchangeOperator / ExchangeAggregationStatesOperator
// over a RankGroup of one rank on the real RCCL (every collective degenerates to a copy; the operators, their work orders and
// the C ABI calls are the N-rank ones).
//
//   c4   Select(orders -> hash-partitioned on o_orderkey) --stream--> PartitionExchange --stream--> BuildHash(o_orderkey)
//        Select(lineitem -> hash-partitioned on l_orderkey) --stream--> PartitionExchange --stream--> HashJoin  [after the build]
//        -> output relation (l_orderkey, o_payload, l_payload)
//        (reference shape: relational_operators/HashJoinOperator.cpp:220-231, BuildHashOperator.cpp:82-91,
//        storage/InsertDestination.hpp:490-660; the per-partition work orders, with partition = GPU)
//   c5   TPC-H Q3: Select(customer: c_mktsegment = BUILDING) -> broadcast exchange -> BuildHash(c_custkey) + LIP filter
//        Select(orders: o_orderdate < DATE) + LIP probe -> HashJoin(semi: o_custkey in customer) -> broadcast exchange
//        -> BuildHash(o_orderkey) + LIP filter; Select(lineitem: l_shipdate > DATE) + LIP probe -> HashJoin(l_orderkey)
//        -> Aggregation(GROUP BY l_orderkey: SUM(l_extendedprice * (1 - l_discount)), CollisionFreeVector)
//        -> ExchangeAggregationStates -> FinalizeAggregation (this rank's key range) -> SortRunGeneration -> SortMergeRun (top 10)
//
// usage: partitioned_operators_bench c4 [orders_per_rank [steps warmup workers blocks_per_work_order]]
//        partitioned_operators_bench c5 [sf_per_rank     [steps warmup workers blocks_per_work_order]]
// prints one JSON line; exit code 0 only when every step's results check out.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <string>

#include "test_util.hpp"

using namespace quickstep;

namespace {
constexpr std::int64_t kBlockBytes = 4ll << 20;

struct Xorshift {
  std::uint64_t x;
  explicit Xorshift(std::uint64_t seed) : x(seed * 0x9E3779B97F4A7C15ull + 1) {}
  std::uint64_t next() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; }
};

struct StepTimes {
  double total_ms = 0.0, best_ms = 1e30;
  void add(double ms) {
    total_ms += ms;
    best_ms = std::min(best_ms, ms);
  }
};

void reportWorkOrders(const QueryPlan &plan, const ForemanSingleNode &foreman, std::chrono::steady_clock::time_point t0, double ms) {
  const std::uint64_t t0_us = static_cast<std::uint64_t>(std::chrono::duration_cast<std::chrono::microseconds>(t0.time_since_epoch()).count());
  struct PerOp { double ms = 0; int n = 0; std::uint64_t first = ~0ull, last = 0; };
  std::map<std::size_t, PerOp> per_op;
  for (const WorkOrderTimeEntry &e : foreman.getWorkOrderProfilingResults()) {
    PerOp &p = per_op[e.operator_index];
    p.ms += static_cast<double>(e.end_us - e.start_us) / 1e3;
    p.n += 1;
    p.first = std::min(p.first, e.start_us);
    p.last = std::max(p.last, e.end_us);
  }
  for (const auto &kv : per_op) {
    std::fprintf(stderr, "  %2zu %-36s %3d work orders, %7.3f ms summed, first start %+8.3f ms, last end %+8.3f ms\n", kv.first,
                 plan.getOperator(kv.first)->getName().c_str(), kv.second.n, kv.second.ms,
                 (static_cast<double>(kv.second.first) - static_cast<double>(t0_us)) / 1e3,
                 (static_cast<double>(kv.second.last) - static_cast<double>(t0_us)) / 1e3);
  }
  std::fprintf(stderr, "  step wall %.3f ms\n", ms);
}

template <typename K, typename V>
void loadInBlocks(StorageManager *storage, CatalogRelation *rel, const std::vector<K> &a, const std::vector<V> &b, std::int64_t block_rows) {
  const std::int64_t n = static_cast<std::int64_t>(a.size());
  for (std::int64_t at = 0; at < n; at += block_rows) storage->loadBlock(rel, {a.data() + at, b.data() + at}, std::min(block_rows, n - at));
}

// ---- C4 -------------------------------------------------------------------------------------------------------------------
int runC4(std::int64_t orders_per_rank, int steps, int warmup, std::size_t workers, std::size_t run_blocks) {
  const std::vector<unsigned char> id = RankGroup::MakeUniqueId();
  RankGroup group(1, 0, id.data());
  const std::size_t parts = static_cast<std::size_t>(group.world());
  // this rank's share (plans.generate_c4_inputs): a contiguous key range in random row order, 1-7 lines per order clustered on
  // the key; payloads are functions of the key, so a joined row is checked without the other side
  std::vector<std::int32_t> o_key(static_cast<std::size_t>(orders_per_rank)), l_key;
  std::vector<std::int64_t> o_pay(static_cast<std::size_t>(orders_per_rank)), l_pay;
  {
    Xorshift rng(5);
    std::iota(o_key.begin(), o_key.end(), 1);
    for (std::int64_t i = orders_per_rank - 1; i > 0; --i) std::swap(o_key[static_cast<std::size_t>(i)], o_key[rng.next() % static_cast<std::uint64_t>(i + 1)]);
    for (std::size_t i = 0; i < o_key.size(); ++i) o_pay[i] = 3ll * o_key[i] + 1;
    l_key.reserve(static_cast<std::size_t>(orders_per_rank) * 4 + 16);
    l_pay.reserve(static_cast<std::size_t>(orders_per_rank) * 4 + 16);
    for (std::int64_t k = 1; k <= orders_per_rank; ++k) {
      const int lines = 1 + static_cast<int>(rng.next() % 7);
      for (int l = 0; l < lines; ++l) {
        l_key.push_back(static_cast<std::int32_t>(k));
        l_pay.push_back(5ll * k + l);
      }
    }
  }
  const std::int64_t lines = static_cast<std::int64_t>(l_key.size());
  StorageManager storage;
  CatalogRelation orders(1, "orders"), lineitem(2, "lineitem");
  for (CatalogRelation *r : {&orders, &lineitem}) {
    r->addAttribute("key", Type::Int());
    r->addAttribute("payload", Type::Long());
  }
  const std::int64_t block_rows = kBlockBytes / 12;
  loadInBlocks(&storage, &orders, o_key, o_pay, block_rows);
  loadInBlocks(&storage, &lineitem, l_key, l_pay, block_rows);
  std::vector<std::int32_t>().swap(o_key);
  std::vector<std::int64_t>().swap(o_pay);
  std::vector<std::int32_t>().swap(l_key);
  std::vector<std::int64_t>().swap(l_pay);

  StepTimes times;
  std::size_t work_orders = 0;
  std::uint64_t bytes_exchanged = 0;
  for (int it = 0; it < warmup + steps; ++it) {
    const auto t0 = std::chrono::steady_clock::now();
    // the query's temporary relations (a relation remembers its blocks: new ones per query, like the reference's optimizer makes them)
    CatalogRelation o_scattered(3, "o_scattered"), l_scattered(4, "l_scattered"), o_arrived(5, "o_arrived"), l_arrived(6, "l_arrived"), joined(7, "joined");
    for (CatalogRelation *r : {&o_scattered, &l_scattered, &o_arrived, &l_arrived}) {
      r->addAttribute("key", Type::Int());
      r->addAttribute("payload", Type::Long());
      r->setPartitionScheme(parts, 0);
    }
    joined.addAttribute("key", Type::Int());
    joined.addAttribute("o_payload", Type::Long());
    joined.addAttribute("l_payload", Type::Long());
    QueryContext ctx;
    const QueryContext::ExactKeyRange key_range{1, orders_per_rank * static_cast<std::int64_t>(parts)};   // exact statistics of the primary key
    const auto table = ctx.addJoinHashTable(kInt, orders_per_rank + orders_per_rank / 8, parts, &key_range);
    const auto d_o = ctx.addPartitionAwareInsertDestination(&o_scattered, &storage), d_l = ctx.addPartitionAwareInsertDestination(&l_scattered, &storage);
    const auto d_xo = ctx.addInsertDestination(&o_arrived, &storage), d_xl = ctx.addInsertDestination(&l_arrived, &storage);
    const auto d_out = ctx.addInsertDestination(&joined, &storage);
    const auto selection = ctx.addScalarGroup({0, 1, 1});       // probe key, build payload, probe payload
    const std::vector<bool> on_build{false, true, false};
    QueryPlan plan;
    SelectOperator *op_sel_o = new SelectOperator(0, orders, true, o_scattered, d_o, QueryContext::kInvalidPredicateId, std::vector<attribute_id>{0, 1}, true);
    SelectOperator *op_sel_l = new SelectOperator(0, lineitem, true, l_scattered, d_l, QueryContext::kInvalidPredicateId, std::vector<attribute_id>{0, 1}, true);
    PartitionExchangeOperator *op_x_o = new PartitionExchangeOperator(0, o_scattered, false, o_arrived, d_xo, &group);
    PartitionExchangeOperator *op_x_l = new PartitionExchangeOperator(0, l_scattered, false, l_arrived, d_xl, &group);
    BuildHashOperator *op_build = new BuildHashOperator(0, o_arrived, false, {0}, false, parts, table);
    HashJoinOperator *op_join = new HashJoinOperator(0, o_arrived, l_arrived, false, {0}, false, parts, false, joined, d_out, table,
                                                     QueryContext::kInvalidPredicateId, selection, &on_build, HashJoinOperator::JoinType::kInnerJoin);
    for (SelectOperator *op : {op_sel_o, op_sel_l}) op->setBlocksPerWorkOrder(run_blocks);
    op_build->setBlocksPerWorkOrder(run_blocks);
    op_join->setBlocksPerWorkOrder(run_blocks);
    const auto sel_o = plan.addRelationalOperator(op_sel_o);
    const auto x_o = plan.addRelationalOperator(op_x_o);
    const auto build = plan.addRelationalOperator(op_build);
    const auto sel_l = plan.addRelationalOperator(op_sel_l);
    const auto x_l = plan.addRelationalOperator(op_x_l);
    const auto join = plan.addRelationalOperator(op_join);
    const auto drop = plan.addRelationalOperator(new DestroyHashOperator(0, parts, table));
    plan.addDirectDependency(x_o, sel_o, false);
    plan.addDirectDependency(build, x_o, false);
    plan.addDirectDependency(x_l, sel_l, false);
    plan.addDirectDependency(join, x_l, false);
    plan.addDirectDependency(join, build, true);
    plan.addDirectDependency(drop, join, true);
    ForemanSingleNode foreman(&plan, &ctx, &storage, workers);
    foreman.run();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (it >= warmup) times.add(ms);
    work_orders = foreman.getWorkOrderProfilingResults().size();
    bytes_exchanged = op_x_o->bytesSentToPeers() + op_x_l->bytesSentToPeers();
    if (std::getenv("QSX_BENCH_STEP_TIMES") != nullptr) std::fprintf(stderr, "step %d%s: %.3f ms\n", it, it < warmup ? " (warmup)" : "", ms);
    if (std::getenv("QSX_TEST_PROFILE") != nullptr && it == warmup + steps - 1) reportWorkOrders(plan, foreman, t0, ms);
    // ---- results: one output row per lineitem row, every row satisfying the join condition by its payloads (last step: all rows) ----
    std::int64_t out_rows = 0;
    bool ok = true;
    for (block_id b : ctx.getInsertDestination(d_out)->getTouchedBlocks()) {
      BlockReference blk = storage.getBlock(b);
      const std::size_t k = static_cast<std::size_t>(blk->numTuples());
      out_rows += blk->numTuples();
      if (it == warmup + steps - 1) {
        std::vector<std::int32_t> key(k);
        std::vector<std::int64_t> o(k), l(k);
        blk->copyAttributeToHost(0, key.data());
        blk->copyAttributeToHost(1, o.data());
        blk->copyAttributeToHost(2, l.data());
        for (std::size_t i = 0; i < k; ++i) ok = ok && o[i] == 3ll * key[i] + 1 && l[i] - 5ll * key[i] >= 0 && l[i] - 5ll * key[i] < 7;
      }
      storage.deleteBlockOrBlobFile(b);
    }
    EXPECT_EQ(out_rows, lines);
    EXPECT_TRUE(ok);
    for (QueryContext::insert_destination_id d : {d_o, d_l, d_xo, d_xl}) {
      for (block_id b : ctx.getInsertDestination(d)->getTouchedBlocks()) storage.deleteBlockOrBlobFile(b);
    }
  }
  const double ms_per_step = times.total_ms / steps;
  std::printf("{\"path\": \"operators (Select + PartitionExchange + BuildHash + HashJoin under ForemanSingleNode, one rank over RCCL)\", \"config\": \"c4\", "
              "\"rows_per_s\": %.6g, \"ms_per_step\": %.4f, \"best_ms\": %.4f, \"steps\": %d, \"warmup\": %d, \"workers\": %zu, \"blocks_per_work_order\": %zu, "
              "\"block_bytes\": %lld, \"orders\": %lld, \"lineitems\": %lld, \"partitions\": %zu, \"work_orders_per_step\": %zu, \"bytes_sent_to_peers\": %llu, "
              "\"checked\": %s}\n",
              static_cast<double>(orders_per_rank + lines) / (ms_per_step / 1e3), ms_per_step, times.best_ms, steps, warmup, workers, run_blocks,
              static_cast<long long>(kBlockBytes), static_cast<long long>(orders_per_rank), static_cast<long long>(lines), parts, work_orders,
              static_cast<unsigned long long>(bytes_exchanged), g_failures == 0 ? "true" : "false");
  return g_failures == 0 ? 0 : 1;
}

// ---- C5 -------------------------------------------------------------------------------------------------------------------
constexpr std::int32_t kDateCut = 19950315;   // '1995-03-15' as the 4-byte yyyymmdd stand-in of plans.py
constexpr std::int32_t kSegBuilding = 1;

void loadColumns(StorageManager *storage, CatalogRelation *rel, std::int64_t rows, std::int64_t block_rows, const std::vector<std::pair<const char *, int>> &cols) {
  for (std::int64_t at = 0; at < rows; at += block_rows) {
    std::vector<const void *> ptrs;
    for (const auto &c : cols) ptrs.push_back(c.first + at * c.second);
    storage->loadBlock(rel, ptrs, std::min(block_rows, rows - at));
  }
}

int runC5(double sf, int steps, int warmup, std::size_t workers, std::size_t run_blocks) {
  const std::vector<unsigned char> id = RankGroup::MakeUniqueId();
  RankGroup group(1, 0, id.data());
  const std::size_t world = static_cast<std::size_t>(group.world());
  // plans.generate_q3_inputs for rank 0 of 1: contiguous custkey / orderkey ranges in random row order, 1-4 lines per order in
  // a first run and 0-3 in a second (at one rank both runs are this rank's own orders), both clustered on l_orderkey
  const std::int64_t n_c = static_cast<std::int64_t>(150000 * sf), n_o = static_cast<std::int64_t>(1500000 * sf);
  std::vector<std::int32_t> c_custkey(static_cast<std::size_t>(n_c)), c_mktsegment(static_cast<std::size_t>(n_c));
  std::vector<std::int32_t> o_orderkey(static_cast<std::size_t>(n_o)), o_custkey(static_cast<std::size_t>(n_o)), o_orderdate(static_cast<std::size_t>(n_o));
  std::vector<std::int32_t> l_orderkey, l_shipdate;
  std::vector<double> l_price, l_discount;
  Xorshift rng(7);
  auto date = [&]() { return static_cast<std::int32_t>(19920101 + rng.next() % (19981231 - 19920101)); };
  std::iota(c_custkey.begin(), c_custkey.end(), 1);
  for (std::int64_t i = n_c - 1; i > 0; --i) std::swap(c_custkey[static_cast<std::size_t>(i)], c_custkey[rng.next() % static_cast<std::uint64_t>(i + 1)]);
  for (auto &seg : c_mktsegment) seg = static_cast<std::int32_t>(rng.next() % 5);
  std::iota(o_orderkey.begin(), o_orderkey.end(), 1);
  for (std::int64_t i = n_o - 1; i > 0; --i) std::swap(o_orderkey[static_cast<std::size_t>(i)], o_orderkey[rng.next() % static_cast<std::uint64_t>(i + 1)]);
  for (std::int64_t i = 0; i < n_o; ++i) {
    o_custkey[static_cast<std::size_t>(i)] = static_cast<std::int32_t>(1 + rng.next() % static_cast<std::uint64_t>(n_c));
    o_orderdate[static_cast<std::size_t>(i)] = date();
  }
  l_orderkey.reserve(static_cast<std::size_t>(n_o) * 4 + 16);
  for (int run = 0; run < 2; ++run) {
    for (std::int64_t k = 1; k <= n_o; ++k) {
      const int lines = run == 0 ? 1 + static_cast<int>(rng.next() % 4) : static_cast<int>(rng.next() % 4);
      for (int l = 0; l < lines; ++l) l_orderkey.push_back(static_cast<std::int32_t>(k));
    }
  }
  const std::int64_t n_l = static_cast<std::int64_t>(l_orderkey.size());
  l_shipdate.resize(static_cast<std::size_t>(n_l));
  l_price.resize(static_cast<std::size_t>(n_l));
  l_discount.resize(static_cast<std::size_t>(n_l));
  for (std::int64_t i = 0; i < n_l; ++i) {
    l_shipdate[static_cast<std::size_t>(i)] = date();
    l_price[static_cast<std::size_t>(i)] = 900.0 + static_cast<double>(rng.next() % 10410000) / 100.0;
    l_discount[static_cast<std::size_t>(i)] = static_cast<double>(rng.next() % 11) / 100.0;
  }
  // ---- the same query on the host columns ----
  std::vector<char> building(static_cast<std::size_t>(n_c) + 1, 0), order_ok(static_cast<std::size_t>(n_o) + 1, 0);
  std::int64_t want_customers = 0, want_orders = 0, want_pairs = 0;
  for (std::int64_t i = 0; i < n_c; ++i) {
    if (c_mktsegment[static_cast<std::size_t>(i)] == kSegBuilding) {
      building[static_cast<std::size_t>(c_custkey[static_cast<std::size_t>(i)])] = 1;
      ++want_customers;
    }
  }
  for (std::int64_t i = 0; i < n_o; ++i) {
    if (o_orderdate[static_cast<std::size_t>(i)] < kDateCut && building[static_cast<std::size_t>(o_custkey[static_cast<std::size_t>(i)])]) {
      order_ok[static_cast<std::size_t>(o_orderkey[static_cast<std::size_t>(i)])] = 1;
      ++want_orders;
    }
  }
  std::vector<double> revenue(static_cast<std::size_t>(n_o) + 1, 0.0);
  for (std::int64_t i = 0; i < n_l; ++i) {
    const std::size_t k = static_cast<std::size_t>(l_orderkey[static_cast<std::size_t>(i)]);
    if (l_shipdate[static_cast<std::size_t>(i)] > kDateCut && order_ok[k]) {
      revenue[k] += l_price[static_cast<std::size_t>(i)] * (1.0 - l_discount[static_cast<std::size_t>(i)]);
      ++want_pairs;
    }
  }
  std::int64_t want_groups = 0;
  std::vector<std::pair<double, std::int32_t>> top;
  for (std::size_t k = 1; k < revenue.size(); ++k) {
    if (revenue[k] == 0.0) continue;
    ++want_groups;
    top.emplace_back(revenue[k], static_cast<std::int32_t>(k));
  }
  const std::size_t top_k = std::min<std::size_t>(10, top.size());
  std::partial_sort(top.begin(), top.begin() + static_cast<std::ptrdiff_t>(top_k), top.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
  top.resize(top_k);
  std::vector<double>().swap(revenue);

  StorageManager storage;
  CatalogRelation customer(1, "customer"), orders(2, "orders"), lineitem(3, "lineitem");
  customer.addAttribute("c_custkey", Type::Int());
  customer.addAttribute("c_mktsegment", Type::Int());
  orders.addAttribute("o_orderkey", Type::Int());
  orders.addAttribute("o_custkey", Type::Int());
  orders.addAttribute("o_orderdate", Type::Int());
  lineitem.addAttribute("l_orderkey", Type::Int());
  lineitem.addAttribute("l_extendedprice", Type::Double());
  lineitem.addAttribute("l_discount", Type::Double());
  lineitem.addAttribute("l_shipdate", Type::Int());
  auto bytes = [](const auto &v) { return reinterpret_cast<const char *>(v.data()); };
  loadColumns(&storage, &customer, n_c, kBlockBytes / 8, {{bytes(c_custkey), 4}, {bytes(c_mktsegment), 4}});
  loadColumns(&storage, &orders, n_o, kBlockBytes / 12, {{bytes(o_orderkey), 4}, {bytes(o_custkey), 4}, {bytes(o_orderdate), 4}});
  loadColumns(&storage, &lineitem, n_l, kBlockBytes / 24, {{bytes(l_orderkey), 4}, {bytes(l_price), 8}, {bytes(l_discount), 8}, {bytes(l_shipdate), 4}});
  for (auto *v : {&c_custkey, &c_mktsegment, &o_orderkey, &o_custkey, &o_orderdate, &l_orderkey, &l_shipdate}) std::vector<std::int32_t>().swap(*v);
  std::vector<double>().swap(l_price);
  std::vector<double>().swap(l_discount);

  StepTimes times;
  std::size_t work_orders = 0;
  for (int it = 0; it < warmup + steps; ++it) {
    const auto t0 = std::chrono::steady_clock::now();
    CatalogRelation cust_sel(10, "cust_sel"), cust_all(11, "cust_all"), ord_sel(12, "ord_sel"), ord_ok(13, "ord_ok"), ord_all(14, "ord_all"),
        li_sel(15, "li_sel"), joined(16, "joined"), agg_out(17, "agg_out"), runs(18, "runs"), top_out(19, "top");
    for (CatalogRelation *r : {&cust_sel, &cust_all}) r->addAttribute("c_custkey", Type::Int());
    ord_sel.addAttribute("o_orderkey", Type::Int());
    ord_sel.addAttribute("o_custkey", Type::Int());
    for (CatalogRelation *r : {&ord_ok, &ord_all}) r->addAttribute("o_orderkey", Type::Int());
    for (CatalogRelation *r : {&li_sel, &joined}) {
      r->addAttribute("l_orderkey", Type::Int());
      r->addAttribute("l_extendedprice", Type::Double());
      r->addAttribute("l_discount", Type::Double());
    }
    for (CatalogRelation *r : {&agg_out, &runs, &top_out}) {
      r->addAttribute("l_orderkey", Type::Int());
      r->addAttribute("revenue", Type::Double().getNullableVersion());
    }
    QueryContext ctx;
    Predicate p_cust, p_ord, p_line;
    p_cust.conjuncts.push_back({1, ComparisonID::kEqual, TypedLiteral::Int(kSegBuilding)});
    p_ord.conjuncts.push_back({2, ComparisonID::kLess, TypedLiteral::Int(kDateCut)});
    p_line.conjuncts.push_back({3, ComparisonID::kGreater, TypedLiteral::Int(kDateCut)});
    const auto pid_cust = ctx.addPredicate(p_cust), pid_ord = ctx.addPredicate(p_ord), pid_line = ctx.addPredicate(p_line);
    const auto d_cust = ctx.addInsertDestination(&cust_sel, &storage), d_cust_all = ctx.addInsertDestination(&cust_all, &storage),
               d_ord = ctx.addInsertDestination(&ord_sel, &storage), d_ord_ok = ctx.addInsertDestination(&ord_ok, &storage),
               d_ord_all = ctx.addInsertDestination(&ord_all, &storage), d_li = ctx.addInsertDestination(&li_sel, &storage),
               d_joined = ctx.addInsertDestination(&joined, &storage), d_agg = ctx.addInsertDestination(&agg_out, &storage),
               d_runs = ctx.addInsertDestination(&runs, &storage), d_top = ctx.addInsertDestination(&top_out, &storage);
    const std::int64_t customers_total = n_c * static_cast<std::int64_t>(world), orders_total = n_o * static_cast<std::int64_t>(world);
    const QueryContext::ExactKeyRange cust_range{1, customers_total}, order_range{1, orders_total};   // exact statistics of the primary keys
    const auto t_cust = ctx.addJoinHashTable(kInt, customers_total / 4, 1, &cust_range);
    const auto t_ord = ctx.addJoinHashTable(kInt, orders_total / 8, 1, &order_range);
    // exact LIP bit vectors on custkey / orderkey, filled by the builds, probed by the selects of the next relation
    const auto lip_c = ctx.addLIPFilter(QSX_LIP_BITVECTOR_EXACT, customers_total + 1, 0), lip_o = ctx.addLIPFilter(QSX_LIP_BITVECTOR_EXACT, orders_total + 1, 0);
    QueryContext::LIPFilterDeployment build_c, probe_c, build_o, probe_o;
    build_c.build_entries.push_back({lip_c, 0});     // cust_all.c_custkey
    probe_c.probe_entries.push_back({lip_c, 1});     // orders.o_custkey
    build_o.build_entries.push_back({lip_o, 0});     // ord_all.o_orderkey
    probe_o.probe_entries.push_back({lip_o, 0});     // lineitem.l_orderkey
    const auto dep_build_c = ctx.addLIPDeployment(build_c), dep_probe_c = ctx.addLIPDeployment(probe_c), dep_build_o = ctx.addLIPDeployment(build_o),
               dep_probe_o = ctx.addLIPDeployment(probe_o);
    const auto sel_semi_ord = ctx.addScalarGroup({0});            // o_orderkey of ord_sel (the probe side of the semi join)
    const auto sel_semi_line = ctx.addScalarGroup({0, 1, 2});     // l_orderkey, price, discount of li_sel
    const std::vector<bool> none_on_build1{false}, none_on_build3{false, false, false};
    AggregationStateSpec spec;
    spec.input_relation = &joined;
    spec.group_by = {0};
    spec.aggregates = {AggregateSpec(AggregationID::kSum, Scalar::Binary(BinaryOperationID::kMultiply, Scalar::Attribute(1),
                                                                         Scalar::Binary(BinaryOperationID::kSubtract, Scalar::Literal(1.0), Scalar::Attribute(2))))};
    spec.strategy = QSX_AGG_COLLISION_FREE;
    spec.collision_free_num_entries = orders_total + 1;
    const auto state = ctx.addAggregationState(spec);
    const auto sort_config = ctx.addSortConfig({{1}, {false}});   // revenue DESC

    QueryPlan plan;
    SelectOperator *op_s_cust = new SelectOperator(0, customer, false, cust_sel, d_cust, pid_cust, std::vector<attribute_id>{0}, true);
    PartitionExchangeOperator *op_x_cust = new PartitionExchangeOperator(0, cust_sel, false, cust_all, d_cust_all, &group, /*broadcast=*/true);
    BuildHashOperator *op_b_cust = new BuildHashOperator(0, cust_all, false, {0}, false, 1, t_cust);
    op_b_cust->deployLIPFilters(dep_build_c);
    SelectOperator *op_s_ord = new SelectOperator(0, orders, false, ord_sel, d_ord, pid_ord, std::vector<attribute_id>{0, 1}, true);
    op_s_ord->deployLIPFilters(dep_probe_c);
    HashJoinOperator *op_j_ord = new HashJoinOperator(0, cust_all, ord_sel, false, {1}, false, 1, false, ord_ok, d_ord_ok, t_cust,
                                                      QueryContext::kInvalidPredicateId, sel_semi_ord, &none_on_build1, HashJoinOperator::JoinType::kLeftSemiJoin);
    PartitionExchangeOperator *op_x_ord = new PartitionExchangeOperator(0, ord_ok, false, ord_all, d_ord_all, &group, /*broadcast=*/true);
    BuildHashOperator *op_b_ord = new BuildHashOperator(0, ord_all, false, {0}, false, 1, t_ord);
    op_b_ord->deployLIPFilters(dep_build_o);
    SelectOperator *op_s_line = new SelectOperator(0, lineitem, false, li_sel, d_li, pid_line, std::vector<attribute_id>{0, 1, 2}, true);
    op_s_line->deployLIPFilters(dep_probe_o);
    HashJoinOperator *op_j_line = new HashJoinOperator(0, ord_all, li_sel, false, {0}, false, 1, false, joined, d_joined, t_ord,
                                                       QueryContext::kInvalidPredicateId, sel_semi_line, &none_on_build3, HashJoinOperator::JoinType::kLeftSemiJoin);
    AggregationOperator *op_agg = new AggregationOperator(0, joined, false, state);
    FinalizeAggregationOperator *op_fin = new FinalizeAggregationOperator(0, state, 1, false, world, agg_out, d_agg);
    op_fin->setRankSlice(static_cast<std::size_t>(group.rank()));
    for (SelectOperator *op : {op_s_cust, op_s_ord, op_s_line}) op->setBlocksPerWorkOrder(run_blocks);
    for (HashJoinOperator *op : {op_j_ord, op_j_line}) op->setBlocksPerWorkOrder(run_blocks);
    for (BuildHashOperator *op : {op_b_cust, op_b_ord}) op->setBlocksPerWorkOrder(run_blocks);
    op_agg->setBlocksPerWorkOrder(run_blocks);
    const auto s_cust = plan.addRelationalOperator(op_s_cust);
    const auto x_cust = plan.addRelationalOperator(op_x_cust);
    const auto b_cust = plan.addRelationalOperator(op_b_cust);
    const auto s_ord = plan.addRelationalOperator(op_s_ord);
    const auto j_ord = plan.addRelationalOperator(op_j_ord);
    const auto x_ord = plan.addRelationalOperator(op_x_ord);
    const auto b_ord = plan.addRelationalOperator(op_b_ord);
    const auto s_line = plan.addRelationalOperator(op_s_line);
    const auto j_line = plan.addRelationalOperator(op_j_line);
    const auto agg = plan.addRelationalOperator(op_agg);
    const auto x_agg = plan.addRelationalOperator(new ExchangeAggregationStatesOperator(0, state, 1, &group));
    const auto fin = plan.addRelationalOperator(op_fin);
    SortRunGenerationOperator *op_gen = new SortRunGenerationOperator(0, agg_out, runs, d_runs, sort_config, false);
    op_gen->setTopK(10);     // LIMIT 10: no run needs more than its first ten tuples
    const auto gen = plan.addRelationalOperator(op_gen);
    const auto merge = plan.addRelationalOperator(new SortMergeRunOperator(0, runs, top_out, d_top, runs, d_runs, sort_config, 4, /*top_k=*/10, false));
    const auto drop_cust = plan.addRelationalOperator(new DestroyHashOperator(0, 1, t_cust));
    const auto drop_ord = plan.addRelationalOperator(new DestroyHashOperator(0, 1, t_ord));
    const auto drop_state = plan.addRelationalOperator(new DestroyAggregationStateOperator(0, state));
    plan.addDirectDependency(x_cust, s_cust, false);
    plan.addDirectDependency(b_cust, x_cust, false);
    plan.addDirectDependency(s_ord, b_cust, true);      // the LIP filter the select probes is complete behind the build
    plan.addDirectDependency(j_ord, b_cust, true);
    plan.addDirectDependency(j_ord, s_ord, false);
    plan.addDirectDependency(x_ord, j_ord, false);
    plan.addDirectDependency(b_ord, x_ord, false);
    plan.addDirectDependency(s_line, b_ord, true);
    plan.addDirectDependency(j_line, b_ord, true);
    plan.addDirectDependency(j_line, s_line, false);
    plan.addDirectDependency(agg, j_line, false);
    plan.addDirectDependency(x_agg, agg, true);
    plan.addDirectDependency(fin, x_agg, true);
    plan.addDirectDependency(gen, fin, false);
    plan.addDirectDependency(merge, gen, false);
    plan.addDirectDependency(drop_cust, j_ord, true);
    plan.addDirectDependency(drop_ord, j_line, true);
    plan.addDirectDependency(drop_state, fin, true);
    ForemanSingleNode foreman(&plan, &ctx, &storage, workers);
    foreman.run();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (it >= warmup) times.add(ms);
    work_orders = foreman.getWorkOrderProfilingResults().size();
    if (std::getenv("QSX_BENCH_STEP_TIMES") != nullptr) std::fprintf(stderr, "step %d%s: %.3f ms\n", it, it < warmup ? " (warmup)" : "", ms);
    if (std::getenv("QSX_TEST_PROFILE") != nullptr && it == warmup + steps - 1) reportWorkOrders(plan, foreman, t0, ms);
    // ---- results ----
    auto rows_of = [&](QueryContext::insert_destination_id d) {
      std::int64_t rows = 0;
      for (block_id b : ctx.getInsertDestination(d)->getTouchedBlocks()) rows += storage.getBlock(b)->numTuples();
      return rows;
    };
    EXPECT_EQ(rows_of(d_cust_all), want_customers);
    EXPECT_EQ(rows_of(d_ord_all), want_orders);
    EXPECT_EQ(rows_of(d_joined), want_pairs);
    EXPECT_EQ(rows_of(d_agg), want_groups);
    std::vector<std::pair<double, std::int32_t>> got;
    for (block_id b : ctx.getInsertDestination(d_top)->getTouchedBlocks()) {
      BlockReference blk = storage.getBlock(b);
      const std::size_t k = static_cast<std::size_t>(blk->numTuples());
      std::vector<std::int32_t> key(k);
      std::vector<double> rev(k);
      blk->copyAttributeToHost(0, key.data());
      blk->copyAttributeToHost(1, rev.data());
      for (std::size_t i = 0; i < k; ++i) got.emplace_back(rev[i], key[i]);
    }
    EXPECT_EQ(got.size(), top.size());
    for (std::size_t i = 0; i < got.size() && i < top.size(); ++i) {
      EXPECT_NEAR(got[i].first, top[i].first, 1e-6 * top[i].first);   // DOUBLE sums: 1e-6 relative (north_star)
      if (i + 1 >= top.size() || top[i].first != top[i + 1].first) EXPECT_EQ(got[i].second, top[i].second);
    }
    for (QueryContext::insert_destination_id d : {d_cust, d_cust_all, d_ord, d_ord_ok, d_ord_all, d_li, d_joined, d_agg, d_runs, d_top}) {
      for (block_id b : ctx.getInsertDestination(d)->getTouchedBlocks()) storage.deleteBlockOrBlobFile(b);
    }
    ctx.destroyLIPFilter(lip_c);
    ctx.destroyLIPFilter(lip_o);
  }
  const double ms_per_step = times.total_ms / steps;
  const std::int64_t input_rows = n_c + n_o + n_l;
  std::printf("{\"path\": \"operators (Q3: Select / BuildHash / HashJoin / Aggregation / Finalize / Sort + PartitionExchange, ExchangeAggregationStates under "
              "ForemanSingleNode, one rank over RCCL)\", \"config\": \"c5\", \"rows_per_s\": %.6g, \"ms_per_step\": %.4f, \"best_ms\": %.4f, \"steps\": %d, "
              "\"warmup\": %d, \"workers\": %zu, \"blocks_per_work_order\": %zu, \"block_bytes\": %lld, \"sf_per_rank\": %.3f, \"input_rows\": %lld, "
              "\"qualifying_customers\": %lld, \"qualifying_orders\": %lld, \"joined_pairs\": %lld, \"groups\": %lld, \"work_orders_per_step\": %zu, \"checked\": %s}\n",
              static_cast<double>(input_rows) / (ms_per_step / 1e3), ms_per_step, times.best_ms, steps, warmup, workers, run_blocks,
              static_cast<long long>(kBlockBytes), sf, static_cast<long long>(input_rows), static_cast<long long>(want_customers),
              static_cast<long long>(want_orders), static_cast<long long>(want_pairs), static_cast<long long>(want_groups), work_orders,
              g_failures == 0 ? "true" : "false");
  return g_failures == 0 ? 0 : 1;
}
}  // namespace

int main(int argc, char **argv) {
  if (qsx_device_count() < 1) {
    std::fprintf(stderr, "partitioned_operators_bench needs an MI355X: %s\n", qsx_status_string(QSX_ERR_NO_DEVICE));
    return 2;
  }
  const std::string config = argc > 1 ? argv[1] : "c4";
  const int steps = argc > 3 ? std::atoi(argv[3]) : 5, warmup = argc > 4 ? std::atoi(argv[4]) : 3;
  const std::size_t workers = argc > 5 ? static_cast<std::size_t>(std::atoi(argv[5])) : 4;
  const std::size_t run_blocks = argc > 6 ? static_cast<std::size_t>(std::atoi(argv[6])) : 64;
  if (config == "c4") return runC4(argc > 2 ? std::atoll(argv[2]) : 18750000, steps, warmup, workers, run_blocks);
  if (config == "c5") return runC5(argc > 2 ? std::atof(argv[2]) : 37.5, steps, warmup, workers, run_blocks);
  std::fprintf(stderr, "usage: partitioned_operators_bench c4|c5 [size [steps warmup workers blocks_per_work_order]]\n");
  return 2;
}
