// Compressed column-store blocks through the operators (the reference's TPC-H DDL stores lineitem,
// orders and partsupp as compressed column stores, benchmarks/tpch/create.sql:69-121).  The same plan — Select with a
// conjunction on compressed attributes, then a grouped aggregation over compressed inputs — runs over a relation loaded
// with compression and over a plain copy; results must be identical, and the blocks must really hold codes
// (truncation for small non-negative integers, dictionaries for few distinct values: CompressedBlockBuilder's choice).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <tuple>

#include "test_util.hpp"

using namespace quickstep;

namespace {
constexpr std::int64_t kRows = 200000;
constexpr std::int64_t kBlockRows = 50000;

struct Lineitem {
  std::vector<std::int32_t> linenumber, shipdate, partkey;
  std::vector<std::int64_t> orderkey;
  std::vector<double> quantity, discount, price;
  Lineitem() {
    std::uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (std::int64_t i = 0; i < kRows; ++i) {
      linenumber.push_back(static_cast<std::int32_t>(rnd() % 7 + 1));            // truncated to 1 byte
      shipdate.push_back(static_cast<std::int32_t>(19920101 + rnd() % 2500));     // 2500 distinct: 2-byte dictionary
      partkey.push_back(static_cast<std::int32_t>(rnd() % 2000000) - 1000000);    // negative values, many distinct: stays plain
      orderkey.push_back(static_cast<std::int64_t>(i / 4));                       // truncated to 2 or 4 bytes per block
      quantity.push_back(static_cast<double>(rnd() % 50 + 1));                    // 50 distinct doubles: 1-byte dictionary
      discount.push_back(static_cast<double>(rnd() % 11) / 100.0);                // 11 distinct: 1-byte dictionary
      price.push_back(900.0 + static_cast<double>(rnd() % 10000000) / 100.0);     // ~all distinct: plain
    }
  }
};

void load(const Lineitem &li, CatalogRelation *rel, StorageManager *storage, bool compressed) {
  for (const char *name : {"l_linenumber", "l_shipdate", "l_partkey"}) rel->addAttribute(name, Type::Int());
  rel->addAttribute("l_orderkey", Type::Long());
  for (const char *name : {"l_quantity", "l_discount", "l_extendedprice"}) rel->addAttribute(name, Type::Double());
  const std::vector<bool> all(7, true);
  for (std::int64_t at = 0; at < kRows; at += kBlockRows) {
    storage->loadBlock(rel, {li.linenumber.data() + at, li.shipdate.data() + at, li.partkey.data() + at, li.orderkey.data() + at,
                             li.quantity.data() + at, li.discount.data() + at, li.price.data() + at},
                       kBlockRows, 0, compressed ? &all : nullptr);
  }
}

struct Output {
  std::vector<std::int32_t> linenumber;
  std::vector<double> quantity, price;
  std::vector<std::int64_t> agg_key_count;   // per l_linenumber group: count
  std::vector<double> agg_sum_disc_price, agg_min_qty;
};

std::size_t g_blocks_per_work_order = 1;

Output run(const Lineitem &li, bool compressed, bool use_foreman) {
  CatalogRelation lineitem(1, "lineitem"), selected(2, "selected"), result(3, "result");
  StorageManager storage;
  load(li, &lineitem, &storage, compressed);
  if (compressed) {
    // what CompressedBlockBuilder decides for these columns
    BlockReference b = storage.getBlock(lineitem.getBlocksSnapshot().front());
    EXPECT_TRUE(b->compressedAttribute(0) != nullptr && b->compressedAttribute(0)->kind == CompressedAttribute::kTruncated &&
                b->compressedAttribute(0)->code_width == 1);
    EXPECT_TRUE(b->compressedAttribute(1) != nullptr && b->compressedAttribute(1)->kind == CompressedAttribute::kDictionary &&
                b->compressedAttribute(1)->code_width == 2);
    EXPECT_TRUE(b->compressedAttribute(2) == nullptr);
    EXPECT_TRUE(b->compressedAttribute(3) != nullptr && b->compressedAttribute(3)->kind == CompressedAttribute::kTruncated);
    EXPECT_TRUE(b->compressedAttribute(4) != nullptr && b->compressedAttribute(4)->kind == CompressedAttribute::kDictionary &&
                b->compressedAttribute(4)->code_width == 1 && b->compressedAttribute(4)->num_codes == 50);
    EXPECT_TRUE(b->compressedAttribute(5) != nullptr && b->compressedAttribute(5)->num_codes == 11);
    EXPECT_TRUE(b->compressedAttribute(6) == nullptr);
    EXPECT_TRUE(!b->valuesMaterialized(4));   // nothing decoded before an operator asks for the values
  }
  selected.addAttribute("l_linenumber", Type::Int());
  selected.addAttribute("l_quantity", Type::Double());
  selected.addAttribute("l_extendedprice", Type::Double());
  result.addAttribute("l_linenumber", Type::Int());
  result.addAttribute("count", Type::Long());
  result.addAttribute("sum_disc", Type::Double());
  result.addAttribute("min_qty", Type::Double());
  QueryContext ctx;
  Predicate pred;   // l_shipdate <= 19980902-ish AND l_quantity < 24 AND l_discount >= 0.045 AND l_orderkey >= 10
  pred.conjuncts.push_back({1, ComparisonID::kLessOrEqual, TypedLiteral::Int(19920101 + 2000)});
  pred.conjuncts.push_back({4, ComparisonID::kLess, TypedLiteral::Double(24.0)});
  pred.conjuncts.push_back({5, ComparisonID::kGreaterOrEqual, TypedLiteral::Double(0.045)});   // between two dictionary values
  pred.conjuncts.push_back({3, ComparisonID::kGreaterOrEqual, TypedLiteral::Long(10)});
  const auto pred_id = ctx.addPredicate(pred);   // (the aggregation state takes at most QSX_MAX_PRED_TERMS = 4 terms)
  const auto sel_dest = ctx.addInsertDestination(&selected, &storage);
  const auto agg_dest = ctx.addInsertDestination(&result, &storage);
  AggregationStateSpec spec;
  spec.input_relation = &lineitem;
  spec.group_by = {0};
  spec.aggregates = {{AggregationID::kCount, kInvalidAttributeID}, {AggregationID::kSum, 5}, {AggregationID::kMin, 4}};
  spec.predicate = ctx.getPredicate(pred_id);
  spec.strategy = QSX_AGG_COMPACT_KEY;
  spec.estimated_num_groups = 8;
  const auto state = ctx.addAggregationState(spec);
  auto *select = new SelectOperator(0, lineitem, false, selected, sel_dest, pred_id, std::vector<attribute_id>{0, 4, 6}, true);
  auto *aggregate = new AggregationOperator(0, lineitem, true, state);
  auto *finalize = new FinalizeAggregationOperator(0, state, 1, false, 1, result, agg_dest);
  select->setBlocksPerWorkOrder(g_blocks_per_work_order);      // (predicates on codes: the Select goes block by block inside)
  aggregate->setBlocksPerWorkOrder(g_blocks_per_work_order);   // compressed blocks: one qsx_agg_update_coded_blocks per run
  std::vector<std::unique_ptr<RelationalOperator>> owned;
  if (use_foreman) {
    QueryPlan plan;
    plan.addRelationalOperator(select);
    const auto a = plan.addRelationalOperator(aggregate);
    const auto z = plan.addRelationalOperator(finalize);
    plan.addDirectDependency(z, a, true);
    ForemanSingleNode foreman(&plan, &ctx, &storage, 4);
    foreman.run();
  } else {
    for (RelationalOperator *op : {static_cast<RelationalOperator *>(select), static_cast<RelationalOperator *>(aggregate),
                                   static_cast<RelationalOperator *>(finalize)}) {
      owned.emplace_back(op);
      fetchAndExecuteWorkOrders(op, &ctx, &storage);
    }
  }
  // compressed blocks are aggregated on their code stripes (key, predicate and argument attributes all coded here)
  EXPECT_TRUE((ctx.getAggregationState(state, 0)->numBlocksAggregatedOnCodes() > 0) == compressed);
  Output out;
  for (block_id b : ctx.getInsertDestination(sel_dest)->getTouchedBlocks()) {
    BlockReference blk = storage.getBlock(b);
    const std::size_t at = out.linenumber.size(), k = static_cast<std::size_t>(blk->numTuples());
    out.linenumber.resize(at + k); out.quantity.resize(at + k); out.price.resize(at + k);
    blk->copyAttributeToHost(0, out.linenumber.data() + at);
    blk->copyAttributeToHost(1, out.quantity.data() + at);
    blk->copyAttributeToHost(2, out.price.data() + at);
  }
  out.agg_key_count.assign(8, 0);
  out.agg_sum_disc_price.assign(8, 0.0);
  out.agg_min_qty.assign(8, 0.0);
  for (block_id b : ctx.getInsertDestination(agg_dest)->getTouchedBlocks()) {
    BlockReference blk = storage.getBlock(b);
    const std::size_t k = static_cast<std::size_t>(blk->numTuples());
    std::vector<std::int32_t> key(k);
    std::vector<std::int64_t> cnt(k);
    std::vector<double> sum(k), mn(k);
    blk->copyAttributeToHost(0, key.data()); blk->copyAttributeToHost(1, cnt.data());
    blk->copyAttributeToHost(2, sum.data()); blk->copyAttributeToHost(3, mn.data());
    for (std::size_t i = 0; i < k; ++i) {
      out.agg_key_count[key[i]] = cnt[i]; out.agg_sum_disc_price[key[i]] = sum[i]; out.agg_min_qty[key[i]] = mn[i];
    }
  }
  return out;
}

// ---- TPC-H Q1's shape over compressed blocks: aggregates factored through the dictionary codes, block by block ----------------
// (csrc/agg_factored.hpp; benchmarks/tpch/queries/01.sql over create.sql:69-121).  Plain CHAR(1) keys, l_quantity and
// l_discount dictionary-coded with a dictionary of ITS OWN in every block (storage/CompressedBlockBuilder.cpp:300-368): the
// blocks draw their discounts from different value sets, so one code means different discounts from block to block.
extern "C" long long qsx_debug_agg_factored_launches(void);   // test hook of libqsx.so (aggregate.hip)

struct Q1Rows {
  std::vector<char> flag, status;
  std::vector<double> quantity, price, discount;
  Q1Rows() {
    std::uint64_t x = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (std::int64_t i = 0; i < kRows; ++i) {
      const std::int64_t block = i / kBlockRows;
      flag.push_back("ANR"[rnd() % 3]);
      status.push_back("FO"[rnd() % 2]);
      quantity.push_back(static_cast<double>(rnd() % (20 + 10 * block) + 1));                        // 20 / 30 / 40 / 50 entries
      discount.push_back(static_cast<double>(rnd() % (4 + 2 * block) + static_cast<std::uint64_t>(block)) / 100.0);   // shifted value sets
      price.push_back(900.0 + static_cast<double>(rnd() % 10000000) / 100.0);
    }
  }
};
struct Q1Group {
  std::int64_t count = 0;
  double sum_qty = 0, sum_price = 0, sum_disc_price = 0, avg_disc = 0;
};

// with_predicate: a predicate inside the aggregation (Q1's l_shipdate <= DATE is such a predicate_ of the operator).  The state
// over code stripes leaves it to the scans (AggregationOperationState::externalizeCodedPredicate): terms on compressed attributes
// are rewritten on every block's own codes and scanned on the code stripes, the state takes the TupleIdSequence as its filter
// and keeps factoring.
//   1: 20000 <= l_extendedprice < 95000 — two terms on a plain attribute;
//   2: l_quantity < 24 AND l_extendedprice >= 20000 — a term on a dictionary-coded attribute.
bool Q1RowPasses(const Q1Rows &rows, std::int64_t i, int with_predicate) {
  return with_predicate == 1 ? rows.price[i] >= 20000.0 && rows.price[i] < 95000.0 : rows.quantity[i] < 24.0 && rows.price[i] >= 20000.0;
}

std::vector<Q1Group> runQ1(const Q1Rows &rows, bool compressed, std::size_t blocks_per_work_order, bool use_foreman, int with_predicate = 0) {
  CatalogRelation lineitem(1, "lineitem"), result(2, "result");
  StorageManager storage;
  lineitem.addAttribute("l_returnflag", Type::Char(1));
  lineitem.addAttribute("l_linestatus", Type::Char(1));
  for (const char *name : {"l_quantity", "l_extendedprice", "l_discount"}) lineitem.addAttribute(name, Type::Double());
  const std::vector<bool> all(5, true);
  for (std::int64_t at = 0; at < kRows; at += kBlockRows) {
    storage.loadBlock(&lineitem, {rows.flag.data() + at, rows.status.data() + at, rows.quantity.data() + at, rows.price.data() + at,
                                  rows.discount.data() + at}, kBlockRows, 0, compressed ? &all : nullptr);
  }
  if (compressed) {   // every block its own dictionaries
    const auto ids = lineitem.getBlocksSnapshot();
    BlockReference first = storage.getBlock(ids.front()), last = storage.getBlock(ids.back());
    EXPECT_TRUE(first->compressedAttribute(2) != nullptr && first->compressedAttribute(2)->num_codes == 20);
    EXPECT_TRUE(last->compressedAttribute(2) != nullptr && last->compressedAttribute(2)->num_codes == 50);
    EXPECT_TRUE(first->compressedAttribute(4) != nullptr && first->compressedAttribute(4)->num_codes == 4);
    EXPECT_TRUE(last->compressedAttribute(4) != nullptr && last->compressedAttribute(4)->num_codes == 10);
    EXPECT_TRUE(first->compressedAttribute(3) == nullptr);
  }
  result.addAttribute("l_returnflag", Type::Char(1));
  result.addAttribute("l_linestatus", Type::Char(1));
  for (const char *name : {"sum_qty", "sum_price", "sum_disc_price", "avg_disc"}) result.addAttribute(name, Type::Double());
  result.addAttribute("count", Type::Long());
  QueryContext ctx;
  const auto dest = ctx.addInsertDestination(&result, &storage);
  AggregationStateSpec spec;
  spec.input_relation = &lineitem;
  spec.group_by = {0, 1};
  const ScalarPtr disc_price = Scalar::Binary(BinaryOperationID::kMultiply, Scalar::Attribute(3),
                                              Scalar::Binary(BinaryOperationID::kSubtract, Scalar::Literal(1.0), Scalar::Attribute(4)));
  spec.aggregates = {AggregateSpec(AggregationID::kSum, 2), AggregateSpec(AggregationID::kSum, 3), AggregateSpec(AggregationID::kSum, disc_price),
                     AggregateSpec(AggregationID::kAvg, 4), AggregateSpec(AggregationID::kCount, kInvalidAttributeID)};
  spec.strategy = QSX_AGG_COMPACT_KEY;
  spec.estimated_num_groups = 6;
  if (with_predicate != 0) {
    Predicate pred;
    if (with_predicate == 1) {
      pred.conjuncts.push_back({3, ComparisonID::kGreaterOrEqual, TypedLiteral::Double(20000.0)});
      pred.conjuncts.push_back({3, ComparisonID::kLess, TypedLiteral::Double(95000.0)});
    } else {
      pred.conjuncts.push_back({2, ComparisonID::kLess, TypedLiteral::Double(24.0)});
      pred.conjuncts.push_back({3, ComparisonID::kGreaterOrEqual, TypedLiteral::Double(20000.0)});
    }
    spec.predicate = ctx.getPredicate(ctx.addPredicate(pred));
  }
  const auto state = ctx.addAggregationState(spec);
  auto *aggregate = new AggregationOperator(0, lineitem, true, state);
  auto *finalize = new FinalizeAggregationOperator(0, state, 1, false, 1, result, dest);
  aggregate->setBlocksPerWorkOrder(blocks_per_work_order);
  const long long launches_before = qsx_debug_agg_factored_launches();
  std::vector<std::unique_ptr<RelationalOperator>> owned;
  if (use_foreman) {
    QueryPlan plan;
    const auto a = plan.addRelationalOperator(aggregate);
    const auto z = plan.addRelationalOperator(finalize);
    plan.addDirectDependency(z, a, true);
    ForemanSingleNode foreman(&plan, &ctx, &storage, 4);
    foreman.run();
  } else {
    for (RelationalOperator *op : {static_cast<RelationalOperator *>(aggregate), static_cast<RelationalOperator *>(finalize)}) {
      owned.emplace_back(op);
      fetchAndExecuteWorkOrders(op, &ctx, &storage);
    }
  }
  const long long launches = qsx_debug_agg_factored_launches() - launches_before;
  if (compressed) {
    const long long work_orders = static_cast<long long>((kRows / kBlockRows + blocks_per_work_order - 1) / blocks_per_work_order);
    EXPECT_EQ(launches, work_orders);           // every work order — a block, or a run of blocks — through the factored kernel
    EXPECT_TRUE(ctx.getAggregationState(state, 0)->numBlocksAggregatedOnCodes() == kRows / kBlockRows);
  } else {
    EXPECT_EQ(launches, 0ll);
  }
  std::vector<Q1Group> out(6);
  for (block_id b : ctx.getInsertDestination(dest)->getTouchedBlocks()) {
    BlockReference blk = storage.getBlock(b);
    const std::size_t k = static_cast<std::size_t>(blk->numTuples());
    std::vector<char> f(k), st(k);
    std::vector<double> q(k), p(k), dp(k), ad(k);
    std::vector<std::int64_t> cnt(k);
    blk->copyAttributeToHost(0, f.data()); blk->copyAttributeToHost(1, st.data());
    blk->copyAttributeToHost(2, q.data()); blk->copyAttributeToHost(3, p.data());
    blk->copyAttributeToHost(4, dp.data()); blk->copyAttributeToHost(5, ad.data());
    blk->copyAttributeToHost(6, cnt.data());
    for (std::size_t i = 0; i < k; ++i) {
      const std::size_t g = (f[i] == 'A' ? 0 : (f[i] == 'N' ? 1 : 2)) * 2 + (st[i] == 'F' ? 0 : 1);
      out[g].count = cnt[i]; out[g].sum_qty = q[i]; out[g].sum_price = p[i]; out[g].sum_disc_price = dp[i]; out[g].avg_disc = ad[i];
    }
  }
  return out;
}

void testQ1OverCompressedBlocks() {
  setenv("QSX_AGG_FACTORED_MIN_ROWS", "0", 1);   // (a test relation of 200 K rows: the default keeps calls below 256 Ki rows on the decoding kernels)
  const Q1Rows rows;
  std::vector<Q1Group> want(6);
  std::vector<double> disc_sum(6, 0.0);
  for (std::int64_t i = 0; i < kRows; ++i) {
    const std::size_t g = (rows.flag[i] == 'A' ? 0 : (rows.flag[i] == 'N' ? 1 : 2)) * 2 + (rows.status[i] == 'F' ? 0 : 1);
    ++want[g].count;
    want[g].sum_qty += rows.quantity[i];
    want[g].sum_price += rows.price[i];
    want[g].sum_disc_price += rows.price[i] * (1.0 - rows.discount[i]);
    disc_sum[g] += rows.discount[i];
  }
  // the same under a predicate inside the aggregation
  std::vector<Q1Group> want_pred[3] = {std::vector<Q1Group>(6), std::vector<Q1Group>(6), std::vector<Q1Group>(6)};
  std::vector<double> disc_sum_pred[3] = {std::vector<double>(6, 0.0), std::vector<double>(6, 0.0), std::vector<double>(6, 0.0)};
  for (int kind = 1; kind <= 2; ++kind) {
    for (std::int64_t i = 0; i < kRows; ++i) {
      if (!Q1RowPasses(rows, i, kind)) continue;
      const std::size_t g = (rows.flag[i] == 'A' ? 0 : (rows.flag[i] == 'N' ? 1 : 2)) * 2 + (rows.status[i] == 'F' ? 0 : 1);
      ++want_pred[kind][g].count;
      want_pred[kind][g].sum_qty += rows.quantity[i];
      want_pred[kind][g].sum_price += rows.price[i];
      want_pred[kind][g].sum_disc_price += rows.price[i] * (1.0 - rows.discount[i]);
      disc_sum_pred[kind][g] += rows.discount[i];
    }
  }
  for (const int variant : {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10}) {
    const int with_predicate = variant < 6 ? 0 : (variant <= 8 ? 1 : 2);
    const bool compressed = variant != 0 && variant != 6;
    const std::size_t per_work_order = variant <= 1 || variant == 6 ? 1 : (variant <= 3 || variant == 9 ? 3 : 4);
    const bool use_foreman = variant == 3 || variant == 5 || variant == 8;
    const std::vector<Q1Group> got = runQ1(rows, compressed, per_work_order, use_foreman, with_predicate);
    const std::vector<Q1Group> &expect = with_predicate != 0 ? want_pred[with_predicate] : want;
    const std::vector<double> &expect_disc = with_predicate != 0 ? disc_sum_pred[with_predicate] : disc_sum;
    for (std::size_t g = 0; g < 6; ++g) {
      EXPECT_EQ(got[g].count, expect[g].count);
      EXPECT_TRUE(got[g].sum_qty == expect[g].sum_qty);                       // integer-valued doubles: exact
      EXPECT_NEAR(got[g].sum_price, expect[g].sum_price, 1e-9 * expect[g].sum_price);
      EXPECT_NEAR(got[g].sum_disc_price, expect[g].sum_disc_price, 1e-9 * expect[g].sum_disc_price);
      EXPECT_NEAR(got[g].avg_disc, expect_disc[g] / static_cast<double>(expect[g].count), 1e-9);
    }
  }
  unsetenv("QSX_AGG_FACTORED_MIN_ROWS");
}

// ---- a run whose blocks compressed an operand differently ------------------------------------------------------------------------
// The reference compresses every block on its own (storage/CompressedBlockBuilder.cpp:300-368): one attribute may be 1-byte
// dictionary codes in most blocks, 2-byte codes in a block with more distinct values and plain in a block where compression does
// not pay.  The state over code stripes is created for the first block's widths; the blocks that agree with it make one run on
// their codes, the others join the work order's run of plain stripes (decoded once) — two launches per work order, not one per
// odd block — and both states are merged at finalization.
void testAggregationOverBlocksThatCompressedDifferently() {
  const std::int64_t blocks = 6;
  std::vector<std::int32_t> key;
  std::vector<double> value;
  std::uint64_t x = 0x2545F4914F6CDD1Dull;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  std::vector<std::int64_t> want_count(5, 0);
  std::vector<double> want_sum(5, 0.0);
  for (std::int64_t i = 0; i < blocks * kBlockRows; ++i) {
    const std::int64_t b = i / kBlockRows;
    key.push_back(static_cast<std::int32_t>(rnd() % 5));
    value.push_back(static_cast<double>(b == 2 ? rnd() % 1000 : (b == 5 ? rnd() % 5000000 : rnd() % 50)));
    ++want_count[key.back()];
    want_sum[key.back()] += value.back();
  }
  for (const std::size_t per_work_order : {std::size_t(1), std::size_t(4), std::size_t(6)}) {
    CatalogRelation rel(1, "r"), result(2, "result");
    StorageManager storage;
    rel.addAttribute("k", Type::Int());
    rel.addAttribute("v", Type::Double());
    const std::vector<bool> all(2, true);
    for (std::int64_t at = 0; at < blocks * kBlockRows; at += kBlockRows) storage.loadBlock(&rel, {key.data() + at, value.data() + at}, kBlockRows, 0, &all);
    const auto ids = rel.getBlocksSnapshot();
    EXPECT_TRUE(storage.getBlock(ids[0])->compressedAttribute(1) != nullptr && storage.getBlock(ids[0])->compressedAttribute(1)->code_width == 1);
    EXPECT_TRUE(storage.getBlock(ids[2])->compressedAttribute(1) != nullptr && storage.getBlock(ids[2])->compressedAttribute(1)->code_width == 2);
    EXPECT_TRUE(storage.getBlock(ids[5])->compressedAttribute(1) == nullptr);
    result.addAttribute("k", Type::Int());
    result.addAttribute("sum_v", Type::Double());
    result.addAttribute("count", Type::Long());
    QueryContext ctx;
    const auto dest = ctx.addInsertDestination(&result, &storage);
    AggregationStateSpec spec;
    spec.input_relation = &rel;
    spec.group_by = {0};
    spec.aggregates = {AggregateSpec(AggregationID::kSum, 1), AggregateSpec(AggregationID::kCount, kInvalidAttributeID)};
    spec.strategy = QSX_AGG_COMPACT_KEY;
    spec.estimated_num_groups = 5;
    const auto state = ctx.addAggregationState(spec);
    AggregationOperator aggregate(0, rel, true, state);
    FinalizeAggregationOperator finalize(0, state, 1, false, 1, result, dest);
    aggregate.setBlocksPerWorkOrder(per_work_order);
    fetchAndExecuteWorkOrders(&aggregate, &ctx, &storage);
    fetchAndExecuteWorkOrders(&finalize, &ctx, &storage);
    EXPECT_TRUE(ctx.getAggregationState(state, 0)->numBlocksAggregatedOnCodes() == 4);   // blocks 0, 1, 3, 4: the first block's coding
    std::vector<std::int64_t> got_count(5, -1);
    std::vector<double> got_sum(5, -1.0);
    for (block_id b : ctx.getInsertDestination(dest)->getTouchedBlocks()) {
      BlockReference blk = storage.getBlock(b);
      const std::size_t n = static_cast<std::size_t>(blk->numTuples());
      std::vector<std::int32_t> k(n);
      std::vector<double> sum(n);
      std::vector<std::int64_t> cnt(n);
      blk->copyAttributeToHost(0, k.data());
      blk->copyAttributeToHost(1, sum.data());
      blk->copyAttributeToHost(2, cnt.data());
      for (std::size_t i = 0; i < n; ++i) {
        got_sum[k[i]] = sum[i];
        got_count[k[i]] = cnt[i];
      }
    }
    EXPECT_TRUE(got_count == want_count);
    EXPECT_TRUE(got_sum == want_sum);     // integer-valued doubles below 2^53: exact in any order
  }
}

// ---- joins directly on compressed key stripes ------------------------------------------------------------------------------------
// (csrc/block_runs.hpp "Coded key stripes", include/qsx.h qsx_key_coding_t.)  BuildHash and HashJoin work orders over runs of
// CompressedColumnStore blocks hand the join attribute to the kernels as it lies — per block: 2- or 4-byte truncated values,
// 1-byte dictionary codes — where the reference reads it through CompressedTupleStorageSubBlock::getAttributeValue
// (storage/CompressedTupleStorageSubBlock.hpp:225-300).  Same output as over plain blocks, and no block's key attribute is
// ever decoded into a stripe of values (valuesMaterialized stays false) — neither by the build, the probe, the LIP filters,
// nor by the projection of the join attribute itself.
constexpr std::int64_t kDimRows = 120000, kDimBlock = 40000, kFactRows = 250000, kFactBlock = 50000;
struct JoinRows {
  std::vector<std::int64_t> d_key, f_key;
  std::vector<double> d_val;
  std::vector<std::int32_t> f_qty;
  std::vector<std::int32_t> d_lo, d_hi, f_lo, f_hi;   // key % 5 and key / 5: a composite key (INT, INT) that joins like the key itself
  JoinRows() {
    std::uint64_t x = 0x2545F4914F6CDD1Dull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (std::int64_t i = 0; i < kDimRows; ++i) {
      d_key.push_back(i);                       // block 0: < 2^16, truncated to 2 bytes; blocks 1, 2: 17 bits, 4 bytes
      d_val.push_back(0.5 * static_cast<double>(i) + 7.0);
    }
    std::vector<std::int64_t> few;
    for (int i = 0; i < 200; ++i) few.push_back(static_cast<std::int64_t>(rnd() % 150000));
    for (std::int64_t i = 0; i < kFactRows; ++i) {
      const std::int64_t block = i / kFactBlock;
      // blocks 1 and 3 draw from 200 keys (1-byte dictionary codes), the others from 0 .. 149999 (4-byte truncation; a fifth misses)
      f_key.push_back(block % 2 == 1 ? few[rnd() % few.size()] : static_cast<std::int64_t>(rnd() % 150000));
      f_qty.push_back(static_cast<std::int32_t>(rnd() % 50));
    }
    for (std::int64_t k : d_key) { d_lo.push_back(static_cast<std::int32_t>(k % 5)); d_hi.push_back(static_cast<std::int32_t>(k / 5)); }
    for (std::int64_t k : f_key) { f_lo.push_back(static_cast<std::int32_t>(k % 5)); f_hi.push_back(static_cast<std::int32_t>(k / 5)); }
  }
};
struct JoinOut {
  std::vector<std::int64_t> key;
  std::vector<std::int32_t> qty;
  std::vector<double> val;
};
// composite: the join key is (lo, hi) = (key % 5, key / 5) — two INT attributes, both compressed (1-byte and 2-byte truncation):
// the packed key of a run is made from the code stripes (qsx_join_key_pack_blocks_coded).
JoinOut runCompressedJoin(const JoinRows &rows, bool compressed, bool exact_stats, std::size_t per_work_order, bool use_foreman,
                          HashJoinOperator::JoinType join_type, bool with_lip, bool composite = false) {
  CatalogRelation dim(1, "dim"), fact(2, "fact"), result(3, "result");
  StorageManager storage;
  dim.addAttribute("d_key", Type::Long());
  dim.addAttribute("d_val", Type::Double());
  fact.addAttribute("f_key", Type::Long());
  fact.addAttribute("f_qty", Type::Int());
  for (CatalogRelation *r : {&dim, &fact}) {
    r->addAttribute("lo", Type::Int());
    r->addAttribute("hi", Type::Int());
  }
  const std::vector<bool> both(4, true);
  for (std::int64_t at = 0; at < kDimRows; at += kDimBlock) {
    storage.loadBlock(&dim, {rows.d_key.data() + at, rows.d_val.data() + at, rows.d_lo.data() + at, rows.d_hi.data() + at}, kDimBlock, 0,
                      compressed ? &both : nullptr);
  }
  for (std::int64_t at = 0; at < kFactRows; at += kFactBlock) {
    storage.loadBlock(&fact, {rows.f_key.data() + at, rows.f_qty.data() + at, rows.f_lo.data() + at, rows.f_hi.data() + at}, kFactBlock, 0,
                      compressed ? &both : nullptr);
  }
  if (compressed) {
    const auto d = dim.getBlocksSnapshot(), f = fact.getBlocksSnapshot();
    const CompressedAttribute *d0 = storage.getBlock(d[0])->compressedAttribute(0), *d2 = storage.getBlock(d[2])->compressedAttribute(0);
    const CompressedAttribute *f0 = storage.getBlock(f[0])->compressedAttribute(0), *f1 = storage.getBlock(f[1])->compressedAttribute(0);
    EXPECT_TRUE(d0 != nullptr && d0->kind == CompressedAttribute::kTruncated && d0->code_width == 2);
    EXPECT_TRUE(d2 != nullptr && d2->kind == CompressedAttribute::kTruncated && d2->code_width == 4);
    EXPECT_TRUE(f0 != nullptr && f0->kind == CompressedAttribute::kTruncated && f0->code_width == 4);
    EXPECT_TRUE(f1 != nullptr && f1->kind == CompressedAttribute::kDictionary && f1->code_width == 1);
  }
  const bool inner = join_type == HashJoinOperator::JoinType::kInnerJoin;
  QueryContext ctx;
  const QueryContext::ExactKeyRange range{0, kDimRows - 1};
  const auto table = ctx.addJoinHashTable(kLong, kDimRows, 1, exact_stats ? &range : nullptr);
  const auto dest = ctx.addInsertDestination(&result, &storage);
  std::vector<bool> on_build;
  QueryContext::scalar_group_id selection;
  if (inner) {
    result.addAttribute("f_key", Type::Long());
    result.addAttribute("f_qty", Type::Int());
    result.addAttribute("d_val", Type::Double());
    selection = ctx.addScalarGroup({0, 1, 1});      // the join attribute itself, a probe attribute, a build attribute
    on_build = {false, false, true};
  } else {
    // (a semi / anti join that projected f_key would decode it for the OUTPUT — the compaction copies values; the join does not)
    result.addAttribute("f_qty", Type::Int());
    selection = ctx.addScalarGroup({1});
    on_build = {false};
  }
  const std::vector<attribute_id> key_attrs = composite ? std::vector<attribute_id>{2, 3} : std::vector<attribute_id>{0};
  auto *builder = new BuildHashOperator(0, dim, true, key_attrs, false, 1, table);
  auto *prober = new HashJoinOperator(0, dim, fact, true, key_attrs, false, 1, false, result, dest, table, QueryContext::kInvalidPredicateId,
                                      selection, &on_build, join_type);
  auto *cleaner = new DestroyHashOperator(0, 1, table);
  if (with_lip) {   // a LIP filter on the join attribute: built from the dim's key stripes, probed with the fact's
    // (the anti join probes no filter: a LIP filter drops the tuples it is after)
    const auto lip = ctx.addLIPFilter(QSX_LIP_BITVECTOR_EXACT, kDimRows, 0);
    QueryContext::LIPFilterDeployment build_dep, probe_dep;
    build_dep.build_entries.push_back({lip, 0});
    probe_dep.probe_entries.push_back({lip, 0});
    builder->deployLIPFilters(ctx.addLIPDeployment(build_dep));
    if (join_type != HashJoinOperator::JoinType::kLeftAntiJoin) prober->deployLIPFilters(ctx.addLIPDeployment(probe_dep));
  }
  builder->setBlocksPerWorkOrder(per_work_order);
  prober->setBlocksPerWorkOrder(per_work_order);
  std::vector<std::unique_ptr<RelationalOperator>> owned;
  if (use_foreman) {
    QueryPlan plan;
    const auto b = plan.addRelationalOperator(builder);
    const auto pr = plan.addRelationalOperator(prober);
    const auto c = plan.addRelationalOperator(cleaner);
    plan.addDirectDependency(pr, b, true);
    plan.addDirectDependency(c, pr, true);
    ForemanSingleNode foreman(&plan, &ctx, &storage, 4);
    foreman.run();
  } else {
    owned.emplace_back(builder); owned.emplace_back(prober); owned.emplace_back(cleaner);
    fetchAndExecuteWorkOrders(builder, &ctx, &storage);
    fetchAndExecuteWorkOrders(prober, &ctx, &storage);
  }
  JoinOut out;
  for (block_id b : ctx.getInsertDestination(dest)->getTouchedBlocks()) {
    BlockReference blk = storage.getBlock(b);
    const std::size_t at = out.key.size(), k = static_cast<std::size_t>(blk->numTuples());
    out.key.resize(at + k, 0); out.qty.resize(at + k);
    blk->copyAttributeToHost(inner ? 1 : 0, out.qty.data() + at);
    if (inner) {
      out.val.resize(at + k);
      blk->copyAttributeToHost(0, out.key.data() + at);
      blk->copyAttributeToHost(2, out.val.data() + at);
    }
  }
  if (!use_foreman) fetchAndExecuteWorkOrders(cleaner, &ctx, &storage);
  if (compressed && per_work_order > 1) {   // (a work order over ONE block takes the per-block entry points, which read values)
    int decoded_dim = 0, decoded_fact = 0;
    for (attribute_id a : key_attrs) {
      for (block_id b : dim.getBlocksSnapshot()) decoded_dim += storage.getBlock(b)->valuesMaterialized(a) ? 1 : 0;
      for (block_id b : fact.getBlocksSnapshot()) decoded_fact += storage.getBlock(b)->valuesMaterialized(a) ? 1 : 0;
    }
    if (decoded_dim != 0 || decoded_fact != 0) {
      std::fprintf(stderr, "join type %d, exact_stats %d, per work order %zu, foreman %d, lip %d: %d dim and %d fact key stripes were decoded\n",
                   static_cast<int>(join_type), exact_stats ? 1 : 0, per_work_order, use_foreman ? 1 : 0, with_lip ? 1 : 0, decoded_dim, decoded_fact);
    }
    EXPECT_EQ(decoded_dim, 0);
    EXPECT_EQ(decoded_fact, 0);
  }
  return out;
}

void testJoinsOverCompressedKeys() {
  const JoinRows rows;
  for (const auto join_type : {HashJoinOperator::JoinType::kInnerJoin, HashJoinOperator::JoinType::kLeftSemiJoin,
                               HashJoinOperator::JoinType::kLeftAntiJoin}) {
    const bool inner = join_type == HashJoinOperator::JoinType::kInnerJoin;
    // expected: (f_key, f_qty[, d_val]) of the fact rows with / without a partner, as sorted triples
    std::vector<std::tuple<std::int64_t, std::int32_t, double>> want;
    for (std::int64_t i = 0; i < kFactRows; ++i) {
      const bool found = rows.f_key[i] < kDimRows;
      if (found == (join_type != HashJoinOperator::JoinType::kLeftAntiJoin)) {
        want.emplace_back(inner ? rows.f_key[i] : 0, rows.f_qty[i], inner ? rows.d_val[rows.f_key[i]] : 0.0);
      }
    }
    std::sort(want.begin(), want.end());
    for (const int variant : {0, 1, 2, 3, 4, 5, 6, 7}) {
      const bool compressed = variant != 0;
      const bool composite = variant == 7;
      const bool exact_stats = variant == 2 || variant == 4 || variant == 6;
      const std::size_t per_work_order = variant == 5 ? 1 : (variant <= 2 ? 3 : 8);
      const bool use_foreman = variant == 3 || variant == 4;
      const bool with_lip = variant == 6 || variant == 3;
      const JoinOut got = runCompressedJoin(rows, compressed, exact_stats, per_work_order, use_foreman, join_type, with_lip, composite);
      std::vector<std::tuple<std::int64_t, std::int32_t, double>> have;
      for (std::size_t i = 0; i < got.key.size(); ++i) have.emplace_back(got.key[i], got.qty[i], inner ? got.val[i] : 0.0);
      std::sort(have.begin(), have.end());
      EXPECT_EQ(have.size(), want.size());
      EXPECT_TRUE(have == want);
    }
  }
}
}  // namespace

int main() {
  if (qsx_device_count() < 1) {
    std::fprintf(stderr, "compressed_block_operator_test needs an MI355X: %s\n", qsx_status_string(QSX_ERR_NO_DEVICE));
    return 2;
  }
  testAggregationOverBlocksThatCompressedDifferently();
  const Lineitem li;
  // expected, straight from the host columns
  std::vector<std::int64_t> want_count(8, 0);
  std::vector<double> want_sum(8, 0.0), want_min(8, 1e300);
  std::vector<std::int64_t> want_rows;
  for (std::int64_t i = 0; i < kRows; ++i) {
    if (li.shipdate[i] <= 19920101 + 2000 && li.quantity[i] < 24.0 && li.discount[i] >= 0.045 && li.orderkey[i] >= 10) {
      want_rows.push_back(i);
      ++want_count[li.linenumber[i]];
      want_sum[li.linenumber[i]] += li.discount[i];
      want_min[li.linenumber[i]] = std::min(want_min[li.linenumber[i]], li.quantity[i]);
    }
  }
  for (const int variant : {0, 1, 2, 3}) {
    const bool use_foreman = (variant & 1) != 0;
    g_blocks_per_work_order = (variant & 2) != 0 ? 3 : 1;
    const Output plain = run(li, false, use_foreman);
    const Output comp = run(li, true, use_foreman);
    for (const Output *o : {&plain, &comp}) {
      EXPECT_EQ(o->linenumber.size(), want_rows.size());
      if (!use_foreman && o->linenumber.size() == want_rows.size()) {   // the synchronous driver keeps block order
        for (std::size_t i = 0; i < want_rows.size(); ++i) {
          EXPECT_EQ(o->linenumber[i], li.linenumber[want_rows[i]]);
          EXPECT_TRUE(o->quantity[i] == li.quantity[want_rows[i]] && o->price[i] == li.price[want_rows[i]]);
        }
      }
      for (int g = 1; g <= 7; ++g) {
        EXPECT_EQ(o->agg_key_count[g], want_count[g]);
        EXPECT_NEAR(o->agg_sum_disc_price[g], want_sum[g], 1e-9 * want_sum[g] + 1e-12);
        if (want_count[g] > 0) EXPECT_TRUE(o->agg_min_qty[g] == want_min[g]);
      }
    }
  }
  testQ1OverCompressedBlocks();
  testJoinsOverCompressedKeys();
  return finish("compressed_block_operator_test");
}
