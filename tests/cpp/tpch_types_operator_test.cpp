// TPC-H's own attribute types through the operators: the reference's schema (benchmarks/tpch/create.sql) declares
// c_mktsegment CHAR(10), l_shipmode CHAR(10), l_returnflag / l_linestatus CHAR(1), l_shipdate / o_orderdate DATE
// (types/DatetimeLit.hpp:38-90: 8-byte DateLit compared year, month, day), and Q1 / Q3 (benchmarks/tpch/queries/01.sql,
// 03.sql) put predicates and group-by keys on exactly those: `c_mktsegment = 'BUILDING'`, `o_orderdate < DATE`,
// `l_shipdate <= DATE`, GROUP BY l_orderkey, o_orderdate, o_shippriority.  Every plan below runs over plain column-store
// blocks and over compressed ones (dictionary codes for the CHAR(10) and DATE attributes: predicates scan the codes,
// CompressedStoreUtil.cpp:51-140) and is checked against the same computation on the host columns.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <tuple>

#include "test_util.hpp"

using namespace quickstep;

namespace {
constexpr std::int64_t kRows = 120000;
constexpr std::int64_t kBlockRows = 40000;

struct Tables {
  // customer
  std::vector<std::int32_t> c_custkey;
  std::vector<char> c_mktsegment;   // CHAR(10)
  // orders
  std::vector<std::int32_t> o_orderkey, o_shippriority;
  std::vector<DateLit> o_orderdate;
  // lineitem
  std::vector<unsigned char> l_returnflag, l_linestatus;
  std::vector<DateLit> l_shipdate;
  std::vector<double> l_quantity, l_extendedprice, l_discount, l_tax;
  std::vector<char> l_shipmode;     // CHAR(10)
  Tables() {
    std::uint64_t x = 0x2545F4914F6CDD1Dull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    const char *segments[] = {"AUTOMOBILE", "BUILDING", "FURNITURE", "MACHINERY", "HOUSEHOLD"};
    const char *modes[] = {"REG AIR", "AIR", "RAIL", "SHIP", "TRUCK", "MAIL", "FOB"};
    c_mktsegment.assign(kRows * 10, 0);
    l_shipmode.assign(kRows * 10, 0);
    for (std::int64_t i = 0; i < kRows; ++i) {
      c_custkey.push_back(static_cast<std::int32_t>(i + 1));
      std::strncpy(&c_mktsegment[i * 10], segments[rnd() % 5], 10);   // AUTOMOBILE fills the field: no terminator
      o_orderkey.push_back(static_cast<std::int32_t>(rnd() % 50));
      o_shippriority.push_back(static_cast<std::int32_t>(rnd() % 2));
      DateLit d = DateLit::Create(1992 + static_cast<int>(rnd() % 7), static_cast<std::uint8_t>(1 + rnd() % 12), 1);
      d.unused[0] = static_cast<std::uint8_t>(rnd());   // garbage where the reference's struct has padding
      d.unused[1] = static_cast<std::uint8_t>(rnd());
      o_orderdate.push_back(d);
      l_returnflag.push_back("ANR"[rnd() % 3]);
      l_linestatus.push_back("FO"[rnd() % 2]);
      DateLit s = DateLit::Create(1992 + static_cast<int>(rnd() % 7), static_cast<std::uint8_t>(1 + rnd() % 12), static_cast<std::uint8_t>(1 + rnd() % 28));
      s.unused[1] = static_cast<std::uint8_t>(rnd());
      l_shipdate.push_back(s);
      l_quantity.push_back(static_cast<double>(rnd() % 50 + 1));
      l_extendedprice.push_back(900.0 + static_cast<double>(rnd() % 10000000) / 100.0);
      l_discount.push_back(static_cast<double>(rnd() % 11) / 100.0);
      l_tax.push_back(static_cast<double>(rnd() % 9) / 100.0);
      std::strncpy(&l_shipmode[i * 10], modes[rnd() % 7], 10);
    }
  }
};

std::string charAt(const std::vector<char> &col, std::int64_t i, int width) {
  return std::string(&col[static_cast<std::size_t>(i) * width], strnlen(&col[static_cast<std::size_t>(i) * width], width));
}
}  // namespace

int main() {
  if (qsx_device_count() < 1) {
    std::fprintf(stderr, "tpch_types_operator_test needs an MI355X: %s\n", qsx_status_string(QSX_ERR_NO_DEVICE));
    return 2;
  }
  const Tables t;
  for (const bool compressed : {false, true}) {
    StorageManager storage;
    CatalogRelation customer(1, "customer"), orders(2, "orders"), lineitem(3, "lineitem");
    customer.addAttribute("c_custkey", Type::Int());
    customer.addAttribute("c_mktsegment", Type::Char(10));
    orders.addAttribute("o_orderkey", Type::Int());
    orders.addAttribute("o_orderdate", Type::Date());
    orders.addAttribute("o_shippriority", Type::Int());
    lineitem.addAttribute("l_returnflag", Type::Char(1));
    lineitem.addAttribute("l_linestatus", Type::Char(1));
    lineitem.addAttribute("l_shipdate", Type::Date());
    lineitem.addAttribute("l_quantity", Type::Double());
    lineitem.addAttribute("l_shipmode", Type::Char(10));
    lineitem.addAttribute("l_extendedprice", Type::Double());
    lineitem.addAttribute("l_discount", Type::Double());
    lineitem.addAttribute("l_tax", Type::Double());
    const std::vector<bool> c_flags{false, true}, o_flags{false, true, false}, l_flags{false, false, true, true, true, true, true, true};
    for (std::int64_t at = 0; at < kRows; at += kBlockRows) {
      storage.loadBlock(&customer, {t.c_custkey.data() + at, t.c_mktsegment.data() + at * 10}, kBlockRows, 0, compressed ? &c_flags : nullptr);
      storage.loadBlock(&orders, {t.o_orderkey.data() + at, t.o_orderdate.data() + at, t.o_shippriority.data() + at}, kBlockRows, 0,
                        compressed ? &o_flags : nullptr);
      storage.loadBlock(&lineitem, {t.l_returnflag.data() + at, t.l_linestatus.data() + at, t.l_shipdate.data() + at, t.l_quantity.data() + at,
                                    t.l_shipmode.data() + at * 10, t.l_extendedprice.data() + at, t.l_discount.data() + at, t.l_tax.data() + at},
                        kBlockRows, 0, compressed ? &l_flags : nullptr);
    }
    if (compressed) {   // the CHAR(10) and DATE attributes really are dictionary-coded
      BlockReference c = storage.getBlock(customer.getBlocksSnapshot().front());
      EXPECT_TRUE(c->compressedAttribute(1) != nullptr && c->compressedAttribute(1)->kind == CompressedAttribute::kDictionary &&
                  c->compressedAttribute(1)->num_codes == 5 && c->compressedAttribute(1)->code_width == 1);
      BlockReference o = storage.getBlock(orders.getBlocksSnapshot().front());
      EXPECT_TRUE(o->compressedAttribute(1) != nullptr && o->compressedAttribute(1)->num_codes == 84);   // 7 years x 12 months
      BlockReference l = storage.getBlock(lineitem.getBlocksSnapshot().front());
      EXPECT_TRUE(l->compressedAttribute(2) != nullptr && l->compressedAttribute(2)->code_width == 2);
      EXPECT_TRUE(l->compressedAttribute(4) != nullptr && l->compressedAttribute(4)->num_codes == 7);
    }
    // ---- Q3: SELECT c_custkey FROM customer WHERE c_mktsegment = 'BUILDING' (and the other comparisons) ----------------
    for (const auto &probe : std::vector<std::pair<ComparisonID, std::string>>{{ComparisonID::kEqual, "BUILDING"},
                                                                                {ComparisonID::kEqual, "AUTOMOBILE"},
                                                                                {ComparisonID::kNotEqual, "MACHINERY"},
                                                                                {ComparisonID::kLess, "BUILDINGS"},
                                                                                {ComparisonID::kGreaterOrEqual, "FURN"},
                                                                                {ComparisonID::kEqual, "BUILD"}}) {
      CatalogRelation out(10, "out");
      out.addAttribute("c_custkey", Type::Int());
      QueryContext ctx;
      Predicate pred;
      pred.conjuncts.push_back({1, probe.first, TypedLiteral::Char(probe.second)});
      const auto pred_id = ctx.addPredicate(pred);
      const auto dest = ctx.addInsertDestination(&out, &storage);
      SelectOperator select(0, customer, false, out, dest, pred_id, std::vector<attribute_id>{0}, true);
      // (work orders over runs of three blocks for every other probe: on the compressed blocks the CHAR comparison is a scan
      // of the code stripes with the comparison rewritten per block, qsx_select_codes_blocks)
      if (probe.second.size() % 2 == 0) select.setBlocksPerWorkOrder(3);
      fetchAndExecuteWorkOrders(&select, &ctx, &storage);
      std::vector<std::int32_t> got;
      for (block_id b : ctx.getInsertDestination(dest)->getTouchedBlocks()) {
        BlockReference blk = storage.getBlock(b);
        const std::size_t at = got.size();
        got.resize(at + static_cast<std::size_t>(blk->numTuples()));
        blk->copyAttributeToHost(0, got.data() + at);
      }
      std::vector<std::int32_t> want;
      for (std::int64_t i = 0; i < kRows; ++i) {
        const int order = charAt(t.c_mktsegment, i, 10).compare(probe.second);
        bool keep = false;
        switch (probe.first) {
          case ComparisonID::kEqual: keep = order == 0; break;
          case ComparisonID::kNotEqual: keep = order != 0; break;
          case ComparisonID::kLess: keep = order < 0; break;
          default: keep = order >= 0; break;
        }
        if (keep) want.push_back(t.c_custkey[i]);
      }
      EXPECT_EQ(got.size(), want.size());
      EXPECT_TRUE(got == want);
    }
    // ---- a compressed store's SORT column: customer is sorted on c_custkey (1 .. kRows); in the compressed run the key is
    // truncated to 4-byte codes... per 40 000-row block to 2- or 4-byte codes, and the predicate is answered by two searches
    {
      CatalogRelation sorted_rel(20, "sorted_customer"), out(21, "out");
      sorted_rel.addAttribute("c_custkey", Type::Int());
      out.addAttribute("c_custkey", Type::Int());
      const std::vector<bool> flags{true};
      for (std::int64_t at = 0; at < kRows; at += kBlockRows) {
        const block_id id = storage.loadBlock(&sorted_rel, {t.c_custkey.data() + at}, kBlockRows, 0, compressed ? &flags : nullptr);
        storage.getBlock(id)->setSortColumn(0);
      }
      for (const auto &probe : std::vector<std::pair<ComparisonID, int>>{{ComparisonID::kLess, 50000}, {ComparisonID::kGreaterOrEqual, 100001},
                                                                         {ComparisonID::kEqual, 40001}, {ComparisonID::kNotEqual, 7}}) {
        QueryContext ctx;
        Predicate pred;
        pred.conjuncts.push_back({0, probe.first, TypedLiteral::Int(probe.second)});
        const auto pred_id = ctx.addPredicate(pred);
        const auto dest = ctx.addInsertDestination(&out, &storage);
        SelectOperator select(0, sorted_rel, false, out, dest, pred_id, std::vector<attribute_id>{0}, true);
        select.setBlocksPerWorkOrder(2);   // the sort-column search over a run of blocks
        fetchAndExecuteWorkOrders(&select, &ctx, &storage);
        std::int64_t got = 0, sum = 0;
        for (block_id b : ctx.getInsertDestination(dest)->getTouchedBlocks()) {
          BlockReference blk = storage.getBlock(b);
          std::vector<std::int32_t> keys(static_cast<std::size_t>(blk->numTuples()));
          blk->copyAttributeToHost(0, keys.data());
          got += blk->numTuples();
          for (std::int32_t k : keys) sum += k;
        }
        std::int64_t want = 0, want_sum = 0;
        for (std::int64_t i = 0; i < kRows; ++i) {
          const int k = t.c_custkey[i];
          const bool keep = probe.first == ComparisonID::kLess ? k < probe.second
                            : probe.first == ComparisonID::kGreaterOrEqual ? k >= probe.second
                            : probe.first == ComparisonID::kEqual ? k == probe.second : k != probe.second;
          if (keep) { ++want; want_sum += k; }
        }
        EXPECT_EQ(got, want);
        EXPECT_EQ(sum, want_sum);
      }
    }
    // ---- Q3: orders WHERE o_orderdate < DATE '1995-03-15', GROUP BY o_orderkey, o_orderdate, o_shippriority --------------
    {
      CatalogRelation result(11, "result");
      result.addAttribute("o_orderkey", Type::Int());
      result.addAttribute("o_orderdate", Type::Date());
      result.addAttribute("o_shippriority", Type::Int());
      result.addAttribute("count", Type::Long());
      QueryContext ctx;
      Predicate pred;
      pred.conjuncts.push_back({1, ComparisonID::kLess, TypedLiteral::Date(1995, 3, 15)});
      const auto pred_id = ctx.addPredicate(pred);
      AggregationStateSpec spec;
      spec.input_relation = &orders;
      spec.group_by = {0, 1, 2};             // INT + DATE + INT = 16 bytes: a wide key
      spec.aggregates = {{AggregationID::kCount, kInvalidAttributeID}};
      spec.predicate = ctx.getPredicate(pred_id);
      spec.strategy = QSX_AGG_GENERIC;
      spec.estimated_num_groups = 4096;
      const auto state = ctx.addAggregationState(spec);
      const auto dest = ctx.addInsertDestination(&result, &storage);
      AggregationOperator op(0, orders, true, state);
      FinalizeAggregationOperator fin(0, state, 1, false, 1, result, dest);
      fetchAndExecuteWorkOrders(&op, &ctx, &storage);
      fetchAndExecuteWorkOrders(&fin, &ctx, &storage);
      std::map<std::tuple<int, int, int, int, int>, std::int64_t> want, got;
      const DateLit cut = DateLit::Create(1995, 3, 15);
      for (std::int64_t i = 0; i < kRows; ++i) {
        if (t.o_orderdate[i] < cut) ++want[{t.o_orderkey[i], t.o_orderdate[i].year, t.o_orderdate[i].month, t.o_orderdate[i].day, t.o_shippriority[i]}];
      }
      for (block_id b : ctx.getInsertDestination(dest)->getTouchedBlocks()) {
        BlockReference blk = storage.getBlock(b);
        const std::size_t k = static_cast<std::size_t>(blk->numTuples());
        std::vector<std::int32_t> okey(k), prio(k);
        std::vector<DateLit> date(k);
        std::vector<std::int64_t> cnt(k);
        blk->copyAttributeToHost(0, okey.data()); blk->copyAttributeToHost(1, date.data());
        blk->copyAttributeToHost(2, prio.data()); blk->copyAttributeToHost(3, cnt.data());
        for (std::size_t i = 0; i < k; ++i) {
          EXPECT_TRUE(date[i].unused[0] == 0 && date[i].unused[1] == 0);   // the padding took no part and comes back zero
          got[{okey[i], date[i].year, date[i].month, date[i].day, prio[i]}] += cnt[i];
        }
      }
      EXPECT_EQ(got.size(), want.size());
      EXPECT_TRUE(got == want);
    }
    // ---- Q1: lineitem WHERE l_shipdate <= DATE '1998-09-02' [AND l_shipmode <> 'MAIL'] GROUP BY l_returnflag, l_linestatus --
    for (const bool with_shipmode : {false, true}) {
      CatalogRelation result(12, "result");
      result.addAttribute("l_returnflag", Type::Char(1));
      result.addAttribute("l_linestatus", Type::Char(1));
      result.addAttribute("sum_qty", Type::Double());
      result.addAttribute("count", Type::Long());
      QueryContext ctx;
      Predicate pred;
      pred.conjuncts.push_back({2, ComparisonID::kLessOrEqual, TypedLiteral::Date(1998, 9, 2)});
      if (with_shipmode) pred.conjuncts.push_back({4, ComparisonID::kNotEqual, TypedLiteral::Char("MAIL")});
      const auto pred_id = ctx.addPredicate(pred);
      AggregationStateSpec spec;
      spec.input_relation = &lineitem;
      spec.group_by = {0, 1};
      spec.aggregates = {{AggregationID::kSum, 3}, {AggregationID::kCount, kInvalidAttributeID}};
      spec.predicate = ctx.getPredicate(pred_id);
      spec.strategy = QSX_AGG_COMPACT_KEY;
      spec.estimated_num_groups = 6;
      const auto state = ctx.addAggregationState(spec);
      const auto dest = ctx.addInsertDestination(&result, &storage);
      AggregationOperator op(0, lineitem, true, state);
      FinalizeAggregationOperator fin(0, state, 1, false, 1, result, dest);
      fetchAndExecuteWorkOrders(&op, &ctx, &storage);
      fetchAndExecuteWorkOrders(&fin, &ctx, &storage);
      std::map<std::pair<char, char>, std::pair<double, std::int64_t>> want, got;
      const DateLit cut = DateLit::Create(1998, 9, 2);
      for (std::int64_t i = 0; i < kRows; ++i) {
        if (cut < t.l_shipdate[i]) continue;
        if (with_shipmode && charAt(t.l_shipmode, i, 10) == "MAIL") continue;
        auto &w = want[{static_cast<char>(t.l_returnflag[i]), static_cast<char>(t.l_linestatus[i])}];
        w.first += t.l_quantity[i];
        w.second += 1;
      }
      for (block_id b : ctx.getInsertDestination(dest)->getTouchedBlocks()) {
        BlockReference blk = storage.getBlock(b);
        const std::size_t k = static_cast<std::size_t>(blk->numTuples());
        std::vector<char> f(k), s(k);
        std::vector<double> sum(k);
        std::vector<std::int64_t> cnt(k);
        blk->copyAttributeToHost(0, f.data()); blk->copyAttributeToHost(1, s.data());
        blk->copyAttributeToHost(2, sum.data()); blk->copyAttributeToHost(3, cnt.data());
        for (std::size_t i = 0; i < k; ++i) got[{f[i], s[i]}] = {sum[i], cnt[i]};
      }
      EXPECT_EQ(got.size(), want.size());
      for (const auto &kv : want) {
        const auto it = got.find(kv.first);
        EXPECT_TRUE(it != got.end());
        if (it == got.end()) continue;
        EXPECT_EQ(it->second.second, kv.second.second);
        EXPECT_NEAR(it->second.first, kv.second.first, 1e-9 * kv.second.first);
      }
    }
  }
  // ---- Q1 as benchmarks/tpch/queries/01.sql writes it: every aggregate, the two arithmetic expressions as Scalar trees ----
  // sum(l_quantity), sum(l_extendedprice), sum(l_extendedprice * (1 - l_discount)), sum(l_extendedprice * (1 - l_discount) *
  // (1 + l_tax)), avg(l_quantity), avg(l_extendedprice), avg(l_discount), count(*) WHERE l_shipdate <= DATE GROUP BY flags;
  // and the projection of an expression by a SelectOperator (ScalarBinaryExpression::getAllValues on its own).
  for (const bool compressed : {false, true}) {
    StorageManager storage;
    CatalogRelation lineitem(30, "lineitem"), result(31, "q1"), projected(32, "projected");
    lineitem.addAttribute("l_returnflag", Type::Char(1));
    lineitem.addAttribute("l_linestatus", Type::Char(1));
    lineitem.addAttribute("l_shipdate", Type::Date());
    lineitem.addAttribute("l_quantity", Type::Double());
    lineitem.addAttribute("l_extendedprice", Type::Double());
    lineitem.addAttribute("l_discount", Type::Double());
    lineitem.addAttribute("l_tax", Type::Double());
    const std::vector<bool> flags{false, false, true, true, false, true, true};
    for (std::int64_t at = 0; at < kRows; at += kBlockRows) {
      storage.loadBlock(&lineitem, {t.l_returnflag.data() + at, t.l_linestatus.data() + at, t.l_shipdate.data() + at, t.l_quantity.data() + at,
                                    t.l_extendedprice.data() + at, t.l_discount.data() + at, t.l_tax.data() + at},
                        kBlockRows, 0, compressed ? &flags : nullptr);
    }
    for (const char *name : {"l_returnflag", "l_linestatus"}) result.addAttribute(name, Type::Char(1));
    for (const char *name : {"sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"}) {
      result.addAttribute(name, Type::Double());
    }
    result.addAttribute("count_order", Type::Long());
    const ScalarPtr price = Scalar::Attribute(4), disc = Scalar::Attribute(5), tax = Scalar::Attribute(6), one = Scalar::Literal(1.0);
    const ScalarPtr disc_price = Scalar::Binary(BinaryOperationID::kMultiply, price, Scalar::Binary(BinaryOperationID::kSubtract, one, disc));
    const ScalarPtr charge = Scalar::Binary(BinaryOperationID::kMultiply, disc_price, Scalar::Binary(BinaryOperationID::kAdd, one, tax));
    QueryContext ctx;
    Predicate pred;
    pred.conjuncts.push_back({2, ComparisonID::kLessOrEqual, TypedLiteral::Date(1998, 9, 2)});
    const auto pred_id = ctx.addPredicate(pred);
    AggregationStateSpec spec;
    spec.input_relation = &lineitem;
    spec.group_by = {0, 1};
    spec.aggregates = {AggregateSpec(AggregationID::kSum, 3), AggregateSpec(AggregationID::kSum, Scalar::Attribute(4)),
                       AggregateSpec(AggregationID::kSum, disc_price), AggregateSpec(AggregationID::kSum, charge),
                       AggregateSpec(AggregationID::kAvg, 3), AggregateSpec(AggregationID::kAvg, 4), AggregateSpec(AggregationID::kAvg, 5),
                       AggregateSpec(AggregationID::kCount, kInvalidAttributeID)};
    spec.predicate = ctx.getPredicate(pred_id);
    spec.strategy = QSX_AGG_COMPACT_KEY;
    spec.estimated_num_groups = 6;
    const auto state = ctx.addAggregationState(spec);
    const auto dest = ctx.addInsertDestination(&result, &storage);
    AggregationOperator op(0, lineitem, true, state);
    FinalizeAggregationOperator fin(0, state, 1, false, 1, result, dest);
    fetchAndExecuteWorkOrders(&op, &ctx, &storage);
    fetchAndExecuteWorkOrders(&fin, &ctx, &storage);
    struct Row { double v[7] = {}; std::int64_t count = 0; };
    std::map<std::pair<char, char>, Row> want;
    const DateLit cut = DateLit::Create(1998, 9, 2);
    for (std::int64_t i = 0; i < kRows; ++i) {
      if (cut < t.l_shipdate[i]) continue;
      Row &w = want[{static_cast<char>(t.l_returnflag[i]), static_cast<char>(t.l_linestatus[i])}];
      const double dp = t.l_extendedprice[i] * (1.0 - t.l_discount[i]);
      w.v[0] += t.l_quantity[i]; w.v[1] += t.l_extendedprice[i]; w.v[2] += dp; w.v[3] += dp * (1.0 + t.l_tax[i]);
      w.v[4] += t.l_quantity[i]; w.v[5] += t.l_extendedprice[i]; w.v[6] += t.l_discount[i];
      ++w.count;
    }
    std::size_t groups = 0;
    for (block_id b : ctx.getInsertDestination(dest)->getTouchedBlocks()) {
      BlockReference blk = storage.getBlock(b);
      const std::size_t k = static_cast<std::size_t>(blk->numTuples());
      std::vector<char> f(k), s(k);
      std::vector<std::vector<double>> vals(7, std::vector<double>(k));
      std::vector<std::int64_t> cnt(k);
      blk->copyAttributeToHost(0, f.data()); blk->copyAttributeToHost(1, s.data());
      for (int a = 0; a < 7; ++a) blk->copyAttributeToHost(static_cast<attribute_id>(2 + a), vals[a].data());
      blk->copyAttributeToHost(9, cnt.data());
      for (std::size_t i = 0; i < k; ++i, ++groups) {
        const auto it = want.find({f[i], s[i]});
        EXPECT_TRUE(it != want.end());
        if (it == want.end()) continue;
        EXPECT_EQ(cnt[i], it->second.count);
        for (int a = 0; a < 7; ++a) {
          const double expect = a < 4 ? it->second.v[a] : it->second.v[a] / static_cast<double>(it->second.count);
          EXPECT_NEAR(vals[a][i], expect, 1e-9 * std::fabs(expect));   // (north star: 1e-6 relative; summation order is all that differs)
        }
      }
    }
    EXPECT_EQ(groups, want.size());
    // SELECT l_quantity, l_extendedprice * (1 - l_discount) * (1 + l_tax) FROM lineitem WHERE l_shipdate <= DATE
    projected.addAttribute("l_quantity", Type::Double());
    projected.addAttribute("charge", Type::Double());
    const auto pdest = ctx.addInsertDestination(&projected, &storage);
    SelectOperator select(0, lineitem, false, projected, pdest, pred_id, std::vector<ScalarPtr>{Scalar::Attribute(3), charge}, true);
    fetchAndExecuteWorkOrders(&select, &ctx, &storage);
    std::vector<double> got_qty, got_charge;
    for (block_id b : ctx.getInsertDestination(pdest)->getTouchedBlocks()) {
      BlockReference blk = storage.getBlock(b);
      const std::size_t at = got_qty.size(), k = static_cast<std::size_t>(blk->numTuples());
      got_qty.resize(at + k); got_charge.resize(at + k);
      blk->copyAttributeToHost(0, got_qty.data() + at);
      blk->copyAttributeToHost(1, got_charge.data() + at);
    }
    std::size_t at = 0;
    bool same = true;
    for (std::int64_t i = 0; i < kRows; ++i) {
      if (cut < t.l_shipdate[i]) continue;
      const double dp = t.l_extendedprice[i] * (1.0 - t.l_discount[i]);
      same = same && at < got_qty.size() && got_qty[at] == t.l_quantity[i] && got_charge[at] == dp * (1.0 + t.l_tax[i]);   // bit-equal: same roundings
      ++at;
    }
    EXPECT_EQ(at, got_qty.size());
    EXPECT_TRUE(same);
  }
  // ---- work-order granularity: Q1 over blocks of the reference's size (4 MB of Q1 columns = 120 K rows), one
  // AggregationWorkOrder per block against one per run of 64 blocks (qsx_agg_update_blocks) — same result, and the run
  // form must not be slower (on the GPU box it is several times faster: the per-block form is launch-bound)
  {
    constexpr std::int64_t kBig = 6000000, kBigBlock = 120000;
    std::vector<unsigned char> flag(kBig), status(kBig);
    std::vector<double> qty(kBig), price(kBig), disc(kBig), tax(kBig);
    std::uint64_t x = 0x853C49E6748FEA9Bull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (std::int64_t i = 0; i < kBig; ++i) {
      flag[i] = "ANR"[rnd() % 3]; status[i] = "FO"[rnd() % 2];
      qty[i] = static_cast<double>(rnd() % 50 + 1); price[i] = 900.0 + static_cast<double>(rnd() % 10000000) / 100.0;
      disc[i] = static_cast<double>(rnd() % 11) / 100.0; tax[i] = static_cast<double>(rnd() % 9) / 100.0;
    }
    StorageManager storage;
    CatalogRelation lineitem(40, "lineitem");
    lineitem.addAttribute("l_returnflag", Type::Char(1));
    lineitem.addAttribute("l_linestatus", Type::Char(1));
    for (const char *name : {"l_quantity", "l_extendedprice", "l_discount", "l_tax"}) lineitem.addAttribute(name, Type::Double());
    for (std::int64_t at = 0; at < kBig; at += kBigBlock) {
      storage.loadBlock(&lineitem, {flag.data() + at, status.data() + at, qty.data() + at, price.data() + at, disc.data() + at, tax.data() + at}, kBigBlock);
    }
    const ScalarPtr p = Scalar::Attribute(3), d = Scalar::Attribute(4), tx = Scalar::Attribute(5), one = Scalar::Literal(1.0);
    const ScalarPtr disc_price = Scalar::Binary(BinaryOperationID::kMultiply, p, Scalar::Binary(BinaryOperationID::kSubtract, one, d));
    const ScalarPtr charge = Scalar::Binary(BinaryOperationID::kMultiply, disc_price, Scalar::Binary(BinaryOperationID::kAdd, one, tx));
    std::map<std::pair<char, char>, std::vector<double>> results[2];
    double millis[2] = {0, 0};
    for (int runs = 0; runs < 2; ++runs) {
      for (int rep = 0; rep < 3; ++rep) {   // the last repetition is the one compared and timed
        CatalogRelation result(41 + runs, "q1");
        for (const char *name : {"l_returnflag", "l_linestatus"}) result.addAttribute(name, Type::Char(1));
        for (const char *name : {"sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_disc"}) result.addAttribute(name, Type::Double());
        result.addAttribute("count_order", Type::Long());
        QueryContext ctx;
        AggregationStateSpec spec;
        spec.input_relation = &lineitem;
        spec.group_by = {0, 1};
        spec.aggregates = {AggregateSpec(AggregationID::kSum, 2), AggregateSpec(AggregationID::kSum, 3), AggregateSpec(AggregationID::kSum, disc_price),
                           AggregateSpec(AggregationID::kSum, charge), AggregateSpec(AggregationID::kAvg, 4),
                           AggregateSpec(AggregationID::kCount, kInvalidAttributeID)};
        spec.strategy = QSX_AGG_COMPACT_KEY;
        spec.estimated_num_groups = 6;
        const auto state = ctx.addAggregationState(spec);
        const auto dest = ctx.addInsertDestination(&result, &storage);
        AggregationOperator op(0, lineitem, true, state);
        if (runs == 1) op.setBlocksPerWorkOrder(64);
        FinalizeAggregationOperator fin(0, state, 1, false, 1, result, dest);
        const auto t0 = std::chrono::steady_clock::now();
        fetchAndExecuteWorkOrders(&op, &ctx, &storage);
        millis[runs] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        fetchAndExecuteWorkOrders(&fin, &ctx, &storage);
        results[runs].clear();
        for (block_id b : ctx.getInsertDestination(dest)->getTouchedBlocks()) {
          BlockReference blk = storage.getBlock(b);
          const std::size_t k = static_cast<std::size_t>(blk->numTuples());
          std::vector<char> f(k), s2(k);
          std::vector<std::vector<double>> vals(5, std::vector<double>(k));
          std::vector<std::int64_t> cnt(k);
          blk->copyAttributeToHost(0, f.data()); blk->copyAttributeToHost(1, s2.data());
          for (int a = 0; a < 5; ++a) blk->copyAttributeToHost(static_cast<attribute_id>(2 + a), vals[a].data());
          blk->copyAttributeToHost(7, cnt.data());
          for (std::size_t i = 0; i < k; ++i) {
            results[runs][{f[i], s2[i]}] = {vals[0][i], vals[1][i], vals[2][i], vals[3][i], vals[4][i], static_cast<double>(cnt[i])};
          }
        }
      }
    }
    EXPECT_EQ(results[0].size(), static_cast<std::size_t>(6));
    EXPECT_EQ(results[1].size(), results[0].size());
    for (const auto &kv : results[0]) {
      const auto it = results[1].find(kv.first);
      EXPECT_TRUE(it != results[1].end());
      if (it == results[1].end()) continue;
      EXPECT_TRUE(it->second[5] == kv.second[5]);                                                       // counts: exact
      for (int a = 0; a < 5; ++a) EXPECT_NEAR(it->second[a], kv.second[a], 1e-9 * std::fabs(kv.second[a]));
    }
    std::printf("Q1 over %lld rows in %lld blocks: one work order per block %.2f ms, per run of 64 blocks %.2f ms\n",
                static_cast<long long>(kBig), static_cast<long long>(kBig / kBigBlock), millis[0], millis[1]);
    EXPECT_TRUE(millis[1] < millis[0] * 1.5);
  }
  return finish("tpch_types_operator_test");
}
