// TPC-H Q3 (benchmarks/tpch/queries/03.sql) as ONE query plan of the operator layer, run by ForemanSingleNode with four
// workers — every operator of the hot path in one DAG, over the reference's own attribute types:
//
//   Select(customer: c_mktsegment = 'BUILDING')            --stream--> BuildHash(c_custkey)
//   Select(orders: o_orderdate < DATE '1995-03-15')        --stream--> HashJoin(o_custkey = c_custkey)   [after the build]
//                                                          --stream--> BuildHash(o_orderkey)
//   Select(lineitem: l_shipdate > DATE '1995-03-15')       --stream--> HashJoin(l_orderkey = o_orderkey)  [after the build]
//   --stream--> Aggregation(GROUP BY l_orderkey, o_orderdate, o_shippriority; SUM(l_extendedprice * (1 - l_discount)))
//   --> FinalizeAggregation --stream--> SortRunGeneration --stream--> SortMergeRun(ORDER BY revenue DESC, o_orderdate; LIMIT 10)
//
// The group-by key is 16 bytes (INT, DATE, INT: a wide key), the aggregate's argument a Scalar tree, c_mktsegment a CHAR(10)
// attribute, the dates 8-byte DateLits.  Checked against the same query computed on the host columns.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <tuple>

#include "test_util.hpp"

using namespace quickstep;

namespace {
int kCustomers = 3000, kOrders = 30000;       // usage: tpch_q3_plan_test [orders [block rows]] (customers = orders / 10)
std::int64_t kBlock = 8192;

struct Db {
  std::vector<std::int32_t> c_custkey;
  std::vector<char> c_mktsegment;
  std::vector<std::int32_t> o_orderkey, o_custkey, o_shippriority;
  std::vector<DateLit> o_orderdate;
  std::vector<std::int32_t> l_orderkey;
  std::vector<double> l_extendedprice, l_discount;
  std::vector<DateLit> l_shipdate;
  Db() {
    std::uint64_t x = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    const char *segments[] = {"AUTOMOBILE", "BUILDING", "FURNITURE", "MACHINERY", "HOUSEHOLD"};
    c_mktsegment.assign(static_cast<std::size_t>(kCustomers) * 10, 0);
    for (int i = 0; i < kCustomers; ++i) {
      c_custkey.push_back(i + 1);
      std::strncpy(&c_mktsegment[static_cast<std::size_t>(i) * 10], segments[rnd() % 5], 10);
    }
    for (int i = 0; i < kOrders; ++i) {
      o_orderkey.push_back(i + 1);
      o_custkey.push_back(static_cast<std::int32_t>(rnd() % kCustomers) + 1);
      o_shippriority.push_back(0);
      o_orderdate.push_back(DateLit::Create(1992 + static_cast<int>(rnd() % 7), static_cast<std::uint8_t>(1 + rnd() % 12), static_cast<std::uint8_t>(1 + rnd() % 28)));
      const int lines = 1 + static_cast<int>(rnd() % 7);
      for (int l = 0; l < lines; ++l) {
        l_orderkey.push_back(i + 1);
        l_extendedprice.push_back(900.0 + static_cast<double>(rnd() % 10000000) / 100.0);
        l_discount.push_back(static_cast<double>(rnd() % 11) / 100.0);
        DateLit s = o_orderdate.back();   // shipped 0 .. 3 months after the order
        const int months = s.month - 1 + static_cast<int>(rnd() % 4);
        s.year += months / 12;
        s.month = static_cast<std::uint8_t>(months % 12 + 1);
        s.unused[0] = static_cast<std::uint8_t>(rnd());
        l_shipdate.push_back(s);
      }
    }
  }
};

template <typename T>
void loadBlocks(StorageManager *storage, CatalogRelation *rel, std::int64_t rows, const std::vector<std::pair<const char *, int>> &cols) {
  for (std::int64_t at = 0; at < rows; at += kBlock) {
    std::vector<const void *> ptrs;
    for (const auto &c : cols) ptrs.push_back(c.first + at * c.second);
    storage->loadBlock(rel, ptrs, std::min<std::int64_t>(kBlock, rows - at));
  }
}

// blocks_per_work_order > 1: Select, HashJoin and Aggregation work orders cover runs of the blocks that have arrived
// (setBlocksPerWorkOrder) — streaming edges deliver blocks one by one, so runs of every length up to the limit occur.
void runQ3(const Db &db, std::size_t blocks_per_work_order) {
  const std::int64_t lines = static_cast<std::int64_t>(db.l_orderkey.size());
  StorageManager storage;
  CatalogRelation customer(1, "customer"), orders(2, "orders"), lineitem(3, "lineitem");
  customer.addAttribute("c_custkey", Type::Int());
  customer.addAttribute("c_mktsegment", Type::Char(10));
  orders.addAttribute("o_orderkey", Type::Int());
  orders.addAttribute("o_custkey", Type::Int());
  orders.addAttribute("o_orderdate", Type::Date());
  orders.addAttribute("o_shippriority", Type::Int());
  lineitem.addAttribute("l_orderkey", Type::Int());
  lineitem.addAttribute("l_extendedprice", Type::Double());
  lineitem.addAttribute("l_discount", Type::Double());
  lineitem.addAttribute("l_shipdate", Type::Date());
  loadBlocks<void>(&storage, &customer, kCustomers, {{reinterpret_cast<const char *>(db.c_custkey.data()), 4}, {db.c_mktsegment.data(), 10}});
  loadBlocks<void>(&storage, &orders, kOrders, {{reinterpret_cast<const char *>(db.o_orderkey.data()), 4}, {reinterpret_cast<const char *>(db.o_custkey.data()), 4},
                                                 {reinterpret_cast<const char *>(db.o_orderdate.data()), 8}, {reinterpret_cast<const char *>(db.o_shippriority.data()), 4}});
  loadBlocks<void>(&storage, &lineitem, lines, {{reinterpret_cast<const char *>(db.l_orderkey.data()), 4}, {reinterpret_cast<const char *>(db.l_extendedprice.data()), 8},
                                                 {reinterpret_cast<const char *>(db.l_discount.data()), 8}, {reinterpret_cast<const char *>(db.l_shipdate.data()), 8}});

  // intermediate relations
  CatalogRelation cust_sel(10, "cust_sel"), ord_sel(11, "ord_sel"), ord_join(12, "ord_join"), li_sel(13, "li_sel"), joined(14, "joined"),
      agg_out(15, "agg_out"), runs(16, "runs"), top(17, "top");
  cust_sel.addAttribute("c_custkey", Type::Int());
  for (const char *n : {"o_orderkey", "o_custkey"}) ord_sel.addAttribute(n, Type::Int());
  ord_sel.addAttribute("o_orderdate", Type::Date());
  ord_sel.addAttribute("o_shippriority", Type::Int());
  ord_join.addAttribute("o_orderkey", Type::Int());
  ord_join.addAttribute("o_orderdate", Type::Date());
  ord_join.addAttribute("o_shippriority", Type::Int());
  li_sel.addAttribute("l_orderkey", Type::Int());
  li_sel.addAttribute("l_extendedprice", Type::Double());
  li_sel.addAttribute("l_discount", Type::Double());
  joined.addAttribute("l_orderkey", Type::Int());
  joined.addAttribute("l_extendedprice", Type::Double());
  joined.addAttribute("l_discount", Type::Double());
  joined.addAttribute("o_orderdate", Type::Date());
  joined.addAttribute("o_shippriority", Type::Int());
  for (CatalogRelation *r : {&agg_out, &runs, &top}) {
    r->addAttribute("l_orderkey", Type::Int());
    r->addAttribute("o_orderdate", Type::Date());
    r->addAttribute("o_shippriority", Type::Int());
    r->addAttribute("revenue", Type::Double());
  }

  QueryContext ctx;
  Predicate p_cust, p_ord, p_line;
  p_cust.conjuncts.push_back({1, ComparisonID::kEqual, TypedLiteral::Char("BUILDING")});
  p_ord.conjuncts.push_back({2, ComparisonID::kLess, TypedLiteral::Date(1995, 3, 15)});
  p_line.conjuncts.push_back({3, ComparisonID::kGreater, TypedLiteral::Date(1995, 3, 15)});
  const auto pid_cust = ctx.addPredicate(p_cust), pid_ord = ctx.addPredicate(p_ord), pid_line = ctx.addPredicate(p_line);
  const auto d_cust = ctx.addInsertDestination(&cust_sel, &storage), d_ord = ctx.addInsertDestination(&ord_sel, &storage),
             d_ordjoin = ctx.addInsertDestination(&ord_join, &storage), d_li = ctx.addInsertDestination(&li_sel, &storage),
             d_joined = ctx.addInsertDestination(&joined, &storage), d_agg = ctx.addInsertDestination(&agg_out, &storage),
             d_runs = ctx.addInsertDestination(&runs, &storage), d_top = ctx.addInsertDestination(&top, &storage);
  const QueryContext::ExactKeyRange cust_range{1, kCustomers}, order_range{1, kOrders};   // exact statistics of the primary keys
  const auto t_cust = ctx.addJoinHashTable(kInt, kCustomers, 1, &cust_range);
  const auto t_ord = ctx.addJoinHashTable(kInt, kOrders, 1, &order_range);
  const auto sel_ord = ctx.addScalarGroup({0, 2, 3});          // of ord_sel (the probe side)
  const std::vector<bool> sel_ord_on_build{false, false, false};
  const auto sel_joined = ctx.addScalarGroup({0, 1, 2, 1, 2});  // l_orderkey, price, discount of li_sel; o_orderdate, o_shippriority of ord_join
  const std::vector<bool> sel_joined_on_build{false, false, false, true, true};
  AggregationStateSpec spec;
  spec.input_relation = &joined;
  spec.group_by = {0, 3, 4};
  spec.aggregates = {AggregateSpec(AggregationID::kSum, Scalar::Binary(BinaryOperationID::kMultiply, Scalar::Attribute(1),
                                                                       Scalar::Binary(BinaryOperationID::kSubtract, Scalar::Literal(1.0), Scalar::Attribute(2))))};
  spec.strategy = QSX_AGG_GENERIC;
  spec.estimated_num_groups = 16;       // the real count is several hundred: the table grows
  const auto state = ctx.addAggregationState(spec);
  const auto sort_config = ctx.addSortConfig({{3, 1}, {false, true}});   // revenue DESC, o_orderdate ASC

  QueryPlan plan;
  SelectOperator *op_s_cust = new SelectOperator(0, customer, false, cust_sel, d_cust, pid_cust, std::vector<attribute_id>{0}, true);
  SelectOperator *op_s_ord = new SelectOperator(0, orders, false, ord_sel, d_ord, pid_ord, std::vector<attribute_id>{0, 1, 2, 3}, true);
  SelectOperator *op_s_line = new SelectOperator(0, lineitem, false, li_sel, d_li, pid_line, std::vector<attribute_id>{0, 1, 2}, true);
  HashJoinOperator *op_j_ord = new HashJoinOperator(0, cust_sel, ord_sel, false, {1}, false, 1, false, ord_join, d_ordjoin, t_cust,
                                                    QueryContext::kInvalidPredicateId, sel_ord, &sel_ord_on_build,
                                                    HashJoinOperator::JoinType::kInnerJoin);
  HashJoinOperator *op_j_line = new HashJoinOperator(0, ord_join, li_sel, false, {0}, false, 1, false, joined, d_joined, t_ord,
                                                     QueryContext::kInvalidPredicateId, sel_joined, &sel_joined_on_build,
                                                     HashJoinOperator::JoinType::kInnerJoin);
  AggregationOperator *op_agg = new AggregationOperator(0, joined, false, state);
  for (SelectOperator *op : {op_s_cust, op_s_ord, op_s_line}) op->setBlocksPerWorkOrder(blocks_per_work_order);
  for (HashJoinOperator *op : {op_j_ord, op_j_line}) op->setBlocksPerWorkOrder(blocks_per_work_order);
  op_agg->setBlocksPerWorkOrder(blocks_per_work_order);
  const auto s_cust = plan.addRelationalOperator(op_s_cust);
  BuildHashOperator *op_b_cust = new BuildHashOperator(0, cust_sel, false, {0}, false, 1, t_cust);
  BuildHashOperator *op_b_ord = new BuildHashOperator(0, ord_join, false, {0}, false, 1, t_ord);
  for (BuildHashOperator *op : {op_b_cust, op_b_ord}) op->setBlocksPerWorkOrder(blocks_per_work_order);
  const auto b_cust = plan.addRelationalOperator(op_b_cust);
  const auto s_ord = plan.addRelationalOperator(op_s_ord);
  const auto j_ord = plan.addRelationalOperator(op_j_ord);
  const auto b_ord = plan.addRelationalOperator(op_b_ord);
  const auto s_line = plan.addRelationalOperator(op_s_line);
  const auto j_line = plan.addRelationalOperator(op_j_line);
  const auto agg = plan.addRelationalOperator(op_agg);
  const auto fin = plan.addRelationalOperator(new FinalizeAggregationOperator(0, state, 1, false, 1, agg_out, d_agg));
  const auto gen = plan.addRelationalOperator(new SortRunGenerationOperator(0, agg_out, runs, d_runs, sort_config, false));
  const auto merge = plan.addRelationalOperator(new SortMergeRunOperator(0, runs, top, d_top, runs, d_runs, sort_config, 4, /*top_k=*/10, false));
  const auto drop_cust = plan.addRelationalOperator(new DestroyHashOperator(0, 1, t_cust));
  const auto drop_ord = plan.addRelationalOperator(new DestroyHashOperator(0, 1, t_ord));
  const auto drop_state = plan.addRelationalOperator(new DestroyAggregationStateOperator(0, state));
  plan.addDirectDependency(b_cust, s_cust, false);
  plan.addDirectDependency(j_ord, b_cust, true);      // pipeline breaker: every build work order before the first probe
  plan.addDirectDependency(j_ord, s_ord, false);
  plan.addDirectDependency(b_ord, j_ord, false);
  plan.addDirectDependency(j_line, b_ord, true);
  plan.addDirectDependency(j_line, s_line, false);
  plan.addDirectDependency(agg, j_line, false);
  plan.addDirectDependency(fin, agg, true);
  plan.addDirectDependency(gen, fin, false);
  plan.addDirectDependency(merge, gen, false);
  plan.addDirectDependency(drop_cust, j_ord, true);
  plan.addDirectDependency(drop_ord, j_line, true);
  plan.addDirectDependency(drop_state, fin, true);
  ForemanSingleNode foreman(&plan, &ctx, &storage, 4);
  const auto t0 = std::chrono::steady_clock::now();
  foreman.run();
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::printf("Q3 plan, %d orders / %lld lineitems in blocks of %lld rows, %zu block(s) per work order: %.2f ms, %zu work orders\n", kOrders,
              static_cast<long long>(lines), static_cast<long long>(kBlock), blocks_per_work_order, ms,
              foreman.getWorkOrderProfilingResults().size());
  if (std::getenv("QSX_TEST_PROFILE") != nullptr) {   // --profile_and_report_workorder_perf: time per operator
    std::map<std::size_t, std::pair<double, int>> per_op;
    std::uint64_t first = ~0ull, last = 0;
    for (const WorkOrderTimeEntry &e : foreman.getWorkOrderProfilingResults()) {
      per_op[e.operator_index].first += static_cast<double>(e.end_us - e.start_us) / 1e3;
      per_op[e.operator_index].second += 1;
      first = std::min(first, e.start_us);
      last = std::max(last, e.end_us);
    }
    for (const auto &kv : per_op) std::printf("  operator %zu: %d work orders, %.2f ms in total\n", kv.first, kv.second.second, kv.second.first);
    std::printf("  first work order start .. last end: %.2f ms\n", static_cast<double>(last - first) / 1e3);
  }

  // the same query on the host columns
  const DateLit cut = DateLit::Create(1995, 3, 15);
  std::vector<bool> building(kCustomers + 1, false);
  for (int i = 0; i < kCustomers; ++i) building[db.c_custkey[i]] = std::string(&db.c_mktsegment[static_cast<std::size_t>(i) * 10], strnlen(&db.c_mktsegment[static_cast<std::size_t>(i) * 10], 10)) == "BUILDING";
  std::vector<int> order_row(kOrders + 1, -1);
  for (int i = 0; i < kOrders; ++i) {
    if (db.o_orderdate[i] < cut && building[db.o_custkey[i]]) order_row[db.o_orderkey[i]] = i;
  }
  std::map<int, double> revenue;     // by l_orderkey (o_orderdate, o_shippriority are functions of it)
  for (std::int64_t i = 0; i < lines; ++i) {
    if (!(cut < db.l_shipdate[i]) || order_row[db.l_orderkey[i]] < 0) continue;
    revenue[db.l_orderkey[i]] += db.l_extendedprice[i] * (1.0 - db.l_discount[i]);
  }
  struct Out { int orderkey; DateLit date; int prio; double revenue; };
  std::vector<Out> want;
  for (const auto &kv : revenue) want.push_back({kv.first, db.o_orderdate[order_row[kv.first]], db.o_shippriority[order_row[kv.first]], kv.second});
  std::sort(want.begin(), want.end(), [](const Out &a, const Out &b) { return a.revenue != b.revenue ? a.revenue > b.revenue : a.date < b.date; });

  // group count of the aggregation and the ten result rows
  std::int64_t groups = 0;
  for (block_id b : ctx.getInsertDestination(d_agg)->getTouchedBlocks()) groups += storage.getBlock(b)->numTuples();
  EXPECT_EQ(groups, static_cast<std::int64_t>(want.size()));
  EXPECT_TRUE(want.size() > 100);
  std::vector<Out> got;
  for (block_id b : ctx.getInsertDestination(d_top)->getTouchedBlocks()) {
    BlockReference blk = storage.getBlock(b);
    const std::size_t k = static_cast<std::size_t>(blk->numTuples());
    std::vector<std::int32_t> key(k), prio(k);
    std::vector<DateLit> date(k);
    std::vector<double> rev(k);
    blk->copyAttributeToHost(0, key.data()); blk->copyAttributeToHost(1, date.data());
    blk->copyAttributeToHost(2, prio.data()); blk->copyAttributeToHost(3, rev.data());
    for (std::size_t i = 0; i < k; ++i) got.push_back({key[i], date[i], prio[i], rev[i]});
  }
  EXPECT_EQ(got.size(), static_cast<std::size_t>(10));
  for (std::size_t i = 0; i < got.size() && i < want.size(); ++i) {
    EXPECT_EQ(got[i].orderkey, want[i].orderkey);
    EXPECT_TRUE(got[i].date == want[i].date);
    EXPECT_EQ(got[i].prio, want[i].prio);
    EXPECT_NEAR(got[i].revenue, want[i].revenue, 1e-9 * want[i].revenue);
  }
}
}  // namespace

int main(int argc, char **argv) {
  if (qsx_device_count() < 1) {
    std::fprintf(stderr, "tpch_q3_plan_test needs an MI355X: %s\n", qsx_status_string(QSX_ERR_NO_DEVICE));
    return 2;
  }
  if (argc > 1) {
    kOrders = std::atoi(argv[1]);
    kCustomers = kOrders / 10;
    if (argc > 2) kBlock = std::atoll(argv[2]);
  }
  const Db db;
  runQ3(db, 1);
  runQ3(db, 4);     // work orders over runs of up to four blocks
  if (argc > 1) {   // a timing run: warm repetitions, then runs of 64
    runQ3(db, 1);
    runQ3(db, 64);
    runQ3(db, 64);
  }
  return finish("tpch_q3_plan_test");
}
