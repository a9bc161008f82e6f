// Work orders over runs of blocks (SelectOperator / HashJoinOperator / AggregationOperator::setBlocksPerWorkOrder): the
// operator decides how many blocks a work order covers (relational_operators/RelationalOperator.hpp:117-119; the
// reference makes one per block, SelectOperator.cpp:83-150, HashJoinOperator.cpp:203-260).  The run forms must give the
// tuples of the block-by-block forms — the same multiset for the join, the same sequence for the select (output order =
// block order, row order) — including the shapes the run form hands back to the per-block path (a nullable attribute).
#include <algorithm>
#include <chrono>
#include <cstring>
#include <random>

#include "test_util.hpp"

using namespace quickstep;

namespace {
constexpr int kBlocks = 150;
constexpr std::int64_t kBlockRows = 40000;       // ragged: block b holds kBlockRows - 13 * (b % 7) rows, block 5 none

struct Lineitem {
  CatalogRelation rel{1, "lineitem"};
  std::vector<std::int32_t> orderkey, quantity;
  std::vector<double> price;
  std::vector<std::int64_t> block_rows;
  // sorted_mode 1: every block sorted on l_quantity and declared so (sort column); 2: the sort column also compressed
  Lineitem(StorageManager *storage, bool nullable_quantity, int sorted_mode = 0) {
    rel.addAttribute("l_orderkey", Type::Int());
    rel.addAttribute("l_quantity", nullable_quantity ? Type::Int().getNullableVersion() : Type::Int());
    rel.addAttribute("l_extendedprice", Type::Double());
    std::mt19937_64 rng(11);
    for (int b = 0; b < kBlocks; ++b) {
      const std::int64_t n = b == 5 ? 0 : kBlockRows - 13 * (b % 7);
      std::vector<std::int32_t> k(n), q(n);
      std::vector<double> p(n);
      for (std::int64_t i = 0; i < n; ++i) {
        k[i] = static_cast<std::int32_t>(rng() % 300000);
        q[i] = static_cast<std::int32_t>(rng() % 50) + 1;
        p[i] = static_cast<double>(rng() % 10000000) / 100.0;
      }
      if (sorted_mode != 0) {
        std::vector<std::int64_t> order(n);
        for (std::int64_t i = 0; i < n; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](std::int64_t x, std::int64_t y) { return q[x] < q[y]; });
        std::vector<std::int32_t> k2(n), q2(n);
        std::vector<double> p2(n);
        for (std::int64_t i = 0; i < n; ++i) { k2[i] = k[order[i]]; q2[i] = q[order[i]]; p2[i] = p[order[i]]; }
        k.swap(k2); q.swap(q2); p.swap(p2);
      }
      std::vector<std::uint64_t> nulls(static_cast<std::size_t>((n + 63) / 64) + 1, 0);
      for (std::int64_t i = 0; i < n; i += 17) nulls[i >> 6] |= 1ull << (63 - (i & 63));
      const std::vector<const std::uint64_t *> null_bitmaps = {nullptr, nullable_quantity ? nulls.data() : nullptr, nullptr};
      const std::vector<bool> compress = {false, sorted_mode == 2, false};
      const block_id id = storage->loadBlock(&rel, {k.data(), q.data(), p.data()}, n, 0, &compress, &null_bitmaps);
      if (sorted_mode != 0) storage->getBlock(id)->setSortColumn(1);
      if (sorted_mode == 2 && n > 1000) EXPECT_TRUE(storage->getBlock(id)->compressedAttribute(1) != nullptr);
      orderkey.insert(orderkey.end(), k.begin(), k.end());
      quantity.insert(quantity.end(), q.begin(), q.end());
      price.insert(price.end(), p.begin(), p.end());
      block_rows.push_back(n);
    }
  }
};

struct Rows {
  std::vector<std::int32_t> key;
  std::vector<double> price;
};

Rows collect(QueryContext &ctx, QueryContext::insert_destination_id dest, StorageManager &storage, std::size_t *blocks_out) {
  Rows r;
  const std::vector<block_id> touched = ctx.getInsertDestination(dest)->getTouchedBlocks();
  *blocks_out = touched.size();
  for (block_id b : touched) {
    BlockReference blk = storage.getBlock(b);
    const std::size_t at = r.key.size(), k = static_cast<std::size_t>(blk->numTuples());
    r.key.resize(at + k);
    r.price.resize(at + k);
    if (k == 0) continue;
    blk->copyAttributeToHost(0, r.key.data() + at);
    blk->copyAttributeToHost(1, r.price.data() + at);
  }
  return r;
}

double g_build_ms = 0;   // the BuildHashOperator of the last runJoin

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// A LIP filter over l_orderkey holding the keys below kLipLimit, built by a BuildHashOperator over a relation of those keys
// (in 10 blocks; the build runs in the run form too).
constexpr std::int32_t kLipLimit = 150000;
QueryContext::lip_deployment_id deployLip(QueryContext *ctx, StorageManager *storage, CatalogRelation *small, std::size_t blocks_per_order) {
  small->addAttribute("k", Type::Int());
  std::vector<std::int32_t> ks(kLipLimit);
  for (std::int32_t i = 0; i < kLipLimit; ++i) ks[i] = i;
  for (std::size_t at = 0; at < ks.size(); at += 15000) storage->loadBlock(small, {ks.data() + at}, 15000);
  const auto filter = ctx->addLIPFilter(QSX_LIP_BITVECTOR_EXACT, kLipLimit, 0);
  QueryContext::LIPFilterDeployment build_dep, probe_dep;
  build_dep.build_entries.push_back({filter, 0});
  probe_dep.probe_entries.push_back({filter, 0});   // l_orderkey
  const auto build_id = ctx->addLIPDeployment(build_dep);
  const auto probe_id = ctx->addLIPDeployment(probe_dep);
  const auto table = ctx->addJoinHashTable(kInt, kLipLimit);
  BuildHashOperator builder(0, *small, true, {0}, false, 1, table);
  builder.deployLIPFilters(build_id);
  builder.setBlocksPerWorkOrder(blocks_per_order);
  fetchAndExecuteWorkOrders(&builder, ctx, storage);
  return probe_id;
}

// select l_orderkey, l_extendedprice from lineitem where l_quantity < 24 and l_extendedprice >= 20000.0
// [with_lip: and l_orderkey passes the LIP filter]
Rows runSelect(bool nullable_quantity, std::size_t blocks_per_order, std::size_t *out_blocks, double *ms, bool with_lip = false,
               int sorted_mode = 0) {
  StorageManager storage;
  Lineitem li(&storage, nullable_quantity, sorted_mode);
  CatalogRelation small(7, "small");
  CatalogRelation out(2, "selected");
  out.addAttribute("l_orderkey", Type::Int());
  out.addAttribute("l_extendedprice", Type::Double());
  QueryContext ctx;
  Predicate p;
  p.conjuncts.push_back(ComparisonPredicate(1, ComparisonID::kLess, TypedLiteral::Int(24)));
  p.conjuncts.push_back(ComparisonPredicate(2, ComparisonID::kGreaterOrEqual, TypedLiteral::Double(20000.0)));
  const auto pred = ctx.addPredicate(p);
  const auto dest = ctx.addInsertDestination(&out, &storage);
  SelectOperator op(0, li.rel, false, out, dest, pred, std::vector<attribute_id>{0, 2}, true);
  if (with_lip) op.deployLIPFilters(deployLip(&ctx, &storage, &small, blocks_per_order));
  op.setBlocksPerWorkOrder(blocks_per_order);
  const double t0 = now_ms();
  fetchAndExecuteWorkOrders(&op, &ctx, &storage);
  *ms = now_ms() - t0;
  Rows got = collect(ctx, dest, storage, out_blocks);
  // the oracle is plain arithmetic here
  Rows want;
  std::size_t row = 0;
  for (std::int64_t n : li.block_rows) {
    for (std::int64_t i = 0; i < n; ++i, ++row) {
      const bool is_null = nullable_quantity && i % 17 == 0;
      if (!is_null && li.quantity[row] < 24 && li.price[row] >= 20000.0 && (!with_lip || li.orderkey[row] < kLipLimit)) {
        want.key.push_back(li.orderkey[row]);
        want.price.push_back(li.price[row]);
      }
    }
  }
  EXPECT_EQ(got.key.size(), want.key.size());
  EXPECT_TRUE(got.key == want.key);        // block order, row order
  EXPECT_TRUE(got.price == want.price);
  return got;
}

// select l_orderkey [build side: o_orderkey], l_extendedprice from orders join lineitem on o_orderkey = l_orderkey,
// orders = the even keys below 200000, each once, in 40 blocks
Rows runJoin(bool exact_stats, std::size_t blocks_per_order, std::size_t *out_blocks, double *ms, bool with_lip = false) {
  StorageManager storage;
  Lineitem li(&storage, false);
  CatalogRelation small(7, "small");
  CatalogRelation orders(3, "orders");
  orders.addAttribute("o_orderkey", Type::Int());
  std::vector<std::int32_t> okeys;
  for (std::int32_t k = 0; k < 200000; k += 2) okeys.push_back(k);
  std::shuffle(okeys.begin(), okeys.end(), std::mt19937_64(3));
  for (std::size_t at = 0; at < okeys.size(); at += 2500) storage.loadBlock(&orders, {okeys.data() + at}, 2500);
  CatalogRelation out(4, "joined");
  out.addAttribute("o_orderkey", Type::Int());
  out.addAttribute("l_extendedprice", Type::Double());
  QueryContext ctx;
  const QueryContext::ExactKeyRange range{0, 199998};
  const auto table = ctx.addJoinHashTable(kInt, 100000, 1, exact_stats ? &range : nullptr);
  const auto dest = ctx.addInsertDestination(&out, &storage);
  const auto selection = ctx.addScalarGroup({0, 2});
  const std::vector<bool> on_build = {true, false};
  BuildHashOperator builder(0, orders, true, {0}, false, 1, table);
  HashJoinOperator prober(0, orders, li.rel, true, {0}, false, 1, false, out, dest, table, QueryContext::kInvalidPredicateId, selection,
                          &on_build, HashJoinOperator::JoinType::kInnerJoin);
  if (with_lip) prober.deployLIPFilters(deployLip(&ctx, &storage, &small, blocks_per_order));
  prober.setBlocksPerWorkOrder(blocks_per_order);
  builder.setBlocksPerWorkOrder(blocks_per_order);
  const double tb = now_ms();
  fetchAndExecuteWorkOrders(&builder, &ctx, &storage);
  g_build_ms = now_ms() - tb;
  const double t0 = now_ms();
  fetchAndExecuteWorkOrders(&prober, &ctx, &storage);
  *ms = now_ms() - t0;
  Rows got = collect(ctx, dest, storage, out_blocks);
  std::vector<std::pair<std::int32_t, double>> g, w;
  for (std::size_t i = 0; i < got.key.size(); ++i) g.emplace_back(got.key[i], got.price[i]);
  for (std::size_t i = 0; i < li.orderkey.size(); ++i) {
    if (li.orderkey[i] < (with_lip ? kLipLimit : 200000) && (li.orderkey[i] & 1) == 0) w.emplace_back(li.orderkey[i], li.price[i]);
  }
  std::sort(g.begin(), g.end());
  std::sort(w.begin(), w.end());
  EXPECT_EQ(g.size(), w.size());
  EXPECT_TRUE(g == w);
  return got;
}

// The same join with every order THREE times in the build relation: three matches per probe tuple, i.e. more output tuples than
// the projecting probe has room for (one per probe tuple) — the work order must notice (count > room), drop the block it
// started and produce the result through the pair list.
void runJoinDuplicateBuildKeys(bool exact_stats, std::size_t blocks_per_order) {
  StorageManager storage;
  Lineitem li(&storage, false);
  CatalogRelation orders(3, "orders");
  orders.addAttribute("o_orderkey", Type::Int());
  orders.addAttribute("o_copy", Type::Long());
  std::vector<std::int32_t> okeys;
  std::vector<std::int64_t> copies;
  for (std::int32_t copy = 0; copy < 3; ++copy) {
    for (std::int32_t k = 0; k < 200000; k += 2) {
      okeys.push_back(k);
      copies.push_back(static_cast<std::int64_t>(k) * 10 + copy);
    }
  }
  std::vector<std::size_t> order(okeys.size());
  for (std::size_t i = 0; i < order.size(); ++i) order[i] = i;
  std::shuffle(order.begin(), order.end(), std::mt19937_64(11));
  std::vector<std::int32_t> k2(okeys.size());
  std::vector<std::int64_t> c2(okeys.size());
  for (std::size_t i = 0; i < order.size(); ++i) {
    k2[i] = okeys[order[i]];
    c2[i] = copies[order[i]];
  }
  for (std::size_t at = 0; at < k2.size(); at += 7500) storage.loadBlock(&orders, {k2.data() + at, c2.data() + at}, 7500);
  CatalogRelation out(4, "joined");
  out.addAttribute("o_copy", Type::Long());
  out.addAttribute("l_extendedprice", Type::Double());
  QueryContext ctx;
  const QueryContext::ExactKeyRange range{0, 199998};
  const auto table = ctx.addJoinHashTable(kInt, 300000, 1, exact_stats ? &range : nullptr);
  const auto dest = ctx.addInsertDestination(&out, &storage);
  const auto selection = ctx.addScalarGroup({1, 2});
  const std::vector<bool> on_build = {true, false};
  BuildHashOperator builder(0, orders, true, {0}, false, 1, table);
  HashJoinOperator prober(0, orders, li.rel, true, {0}, false, 1, false, out, dest, table, QueryContext::kInvalidPredicateId, selection,
                          &on_build, HashJoinOperator::JoinType::kInnerJoin);
  prober.setBlocksPerWorkOrder(blocks_per_order);
  builder.setBlocksPerWorkOrder(blocks_per_order);
  fetchAndExecuteWorkOrders(&builder, &ctx, &storage);
  fetchAndExecuteWorkOrders(&prober, &ctx, &storage);
  std::vector<std::pair<std::int64_t, double>> g, w;
  for (block_id b : ctx.getInsertDestination(dest)->getTouchedBlocks()) {
    BlockReference blk = storage.getBlock(b);
    const std::size_t k = static_cast<std::size_t>(blk->numTuples());
    if (k == 0) continue;
    std::vector<std::int64_t> copy(k);
    std::vector<double> price(k);
    blk->copyAttributeToHost(0, copy.data());
    blk->copyAttributeToHost(1, price.data());
    for (std::size_t i = 0; i < k; ++i) g.emplace_back(copy[i], price[i]);
  }
  for (std::size_t i = 0; i < li.orderkey.size(); ++i) {
    if (li.orderkey[i] < 200000 && (li.orderkey[i] & 1) == 0) {
      for (int copy = 0; copy < 3; ++copy) w.emplace_back(static_cast<std::int64_t>(li.orderkey[i]) * 10 + copy, li.price[i]);
    }
  }
  std::sort(g.begin(), g.end());
  std::sort(w.begin(), w.end());
  EXPECT_TRUE(w.size() > 100000);
  EXPECT_EQ(g.size(), w.size());
  EXPECT_TRUE(g == w);
}

// select l_orderkey, l_extendedprice from lineitem where [not] exists (select * from orders where o_orderkey = l_orderkey and
// o_limit > l_quantity): a semi / an anti join with a residual predicate between the two sides — the pairs of the run, the
// residual on them, the probe tuples that kept (semi) or never had (anti) a pair; every order twice with different limits, so
// that a probe tuple can keep one pair and lose the other
void runSemiAntiResidual(bool anti, bool exact_stats, std::size_t blocks_per_order, std::size_t *out_blocks, bool with_lip = false) {
  StorageManager storage;
  Lineitem li(&storage, false);
  CatalogRelation small(7, "small");
  CatalogRelation orders(3, "orders");
  orders.addAttribute("o_orderkey", Type::Int());
  orders.addAttribute("o_limit", Type::Int());
  std::vector<std::int32_t> okeys, limits;
  for (std::int32_t copy = 0; copy < 2; ++copy) {
    for (std::int32_t k = 0; k < 200000; k += 2) {
      okeys.push_back(k);
      limits.push_back(copy == 0 ? k % 50 : (k / 2) % 7);
    }
  }
  for (std::size_t at = 0; at < okeys.size(); at += 5000) storage.loadBlock(&orders, {okeys.data() + at, limits.data() + at}, 5000);
  CatalogRelation out(4, "kept");
  out.addAttribute("l_orderkey", Type::Int());
  out.addAttribute("l_extendedprice", Type::Double());
  QueryContext ctx;
  const QueryContext::ExactKeyRange range{0, 199998};
  const auto table = ctx.addJoinHashTable(kInt, 200000, 1, exact_stats ? &range : nullptr);
  const auto dest = ctx.addInsertDestination(&out, &storage);
  const auto selection = ctx.addScalarGroup({0, 2});
  const std::vector<bool> on_build = {false, false};
  Predicate residual;   // o_limit (build side) > l_quantity (probe side)
  residual.conjuncts.push_back(ComparisonPredicate::Attributes(1, true, ComparisonID::kGreater, 1, false));
  const auto pred = ctx.addPredicate(residual);
  BuildHashOperator builder(0, orders, true, {0}, false, 1, table);
  HashJoinOperator prober(0, orders, li.rel, true, {0}, false, 1, false, out, dest, table, pred, selection, &on_build,
                          anti ? HashJoinOperator::JoinType::kLeftAntiJoin : HashJoinOperator::JoinType::kLeftSemiJoin);
  // under a LIP filter the work order is only ABOUT the tuples the filter lets through: an anti join must not bring the
  // others back through the complement of the paired tuples
  if (with_lip) prober.deployLIPFilters(deployLip(&ctx, &storage, &small, blocks_per_order));
  prober.setBlocksPerWorkOrder(blocks_per_order);
  builder.setBlocksPerWorkOrder(blocks_per_order);
  fetchAndExecuteWorkOrders(&builder, &ctx, &storage);
  fetchAndExecuteWorkOrders(&prober, &ctx, &storage);
  Rows got = collect(ctx, dest, storage, out_blocks);
  std::vector<std::pair<std::int32_t, double>> g, w;
  for (std::size_t i = 0; i < got.key.size(); ++i) g.emplace_back(got.key[i], got.price[i]);
  for (std::size_t i = 0; i < li.orderkey.size(); ++i) {
    const std::int32_t k = li.orderkey[i];
    if (with_lip && k >= kLipLimit) continue;
    const bool has_order = k < 200000 && (k & 1) == 0;
    const bool kept = has_order && (k % 50 > li.quantity[i] || (k / 2) % 7 > li.quantity[i]);
    if (kept != anti) w.emplace_back(k, li.price[i]);
  }
  std::sort(g.begin(), g.end());
  std::sort(w.begin(), w.end());
  EXPECT_TRUE(w.size() > 1000);
  EXPECT_EQ(g.size(), w.size());
  EXPECT_TRUE(g == w);
}

// select l_orderkey, l_extendedprice, o_limit from lineitem left outer join orders on o_orderkey = l_orderkey: every lineitem
// once, o_limit NULL where the order does not exist (the odd keys and those from 200000 on)
void runOuterJoin(bool exact_stats, std::size_t blocks_per_order, std::size_t *out_blocks) {
  StorageManager storage;
  Lineitem li(&storage, false);
  CatalogRelation orders(3, "orders");
  orders.addAttribute("o_orderkey", Type::Int());
  orders.addAttribute("o_limit", Type::Int());
  std::vector<std::int32_t> okeys, limits;
  for (std::int32_t k = 0; k < 200000; k += 2) {
    okeys.push_back(k);
    limits.push_back(k % 50 + 1);
  }
  std::vector<std::size_t> order(okeys.size());
  for (std::size_t i = 0; i < order.size(); ++i) order[i] = i;
  std::shuffle(order.begin(), order.end(), std::mt19937_64(13));
  std::vector<std::int32_t> k2(okeys.size()), l2(okeys.size());
  for (std::size_t i = 0; i < order.size(); ++i) {
    k2[i] = okeys[order[i]];
    l2[i] = limits[order[i]];
  }
  for (std::size_t at = 0; at < k2.size(); at += 2500) storage.loadBlock(&orders, {k2.data() + at, l2.data() + at}, 2500);
  CatalogRelation out(4, "joined");
  out.addAttribute("l_orderkey", Type::Int());
  out.addAttribute("l_extendedprice", Type::Double());
  out.addAttribute("o_limit", Type::Int().getNullableVersion());
  QueryContext ctx;
  const QueryContext::ExactKeyRange range{0, 199998};
  const auto table = ctx.addJoinHashTable(kInt, 100000, 1, exact_stats ? &range : nullptr);
  const auto dest = ctx.addInsertDestination(&out, &storage);
  const auto selection = ctx.addScalarGroup({0, 2, 1});
  const std::vector<bool> on_build = {false, false, true};
  BuildHashOperator builder(0, orders, true, {0}, false, 1, table);
  HashJoinOperator prober(0, orders, li.rel, true, {0}, false, 1, false, out, dest, table, QueryContext::kInvalidPredicateId, selection,
                          &on_build, HashJoinOperator::JoinType::kLeftOuterJoin);
  prober.setBlocksPerWorkOrder(blocks_per_order);
  builder.setBlocksPerWorkOrder(blocks_per_order);
  fetchAndExecuteWorkOrders(&builder, &ctx, &storage);
  fetchAndExecuteWorkOrders(&prober, &ctx, &storage);
  struct Row { std::int32_t key; double price; std::int32_t limit; bool null; };
  std::vector<Row> g, w;
  const std::vector<block_id> touched = ctx.getInsertDestination(dest)->getTouchedBlocks();
  *out_blocks = touched.size();
  for (block_id b : touched) {
    BlockReference blk = storage.getBlock(b);
    const std::size_t k = static_cast<std::size_t>(blk->numTuples());
    if (k == 0) continue;
    std::vector<std::int32_t> key(k), limit(k);
    std::vector<double> price(k);
    std::vector<std::uint64_t> nulls((k + 63) / 64);
    blk->copyAttributeToHost(0, key.data());
    blk->copyAttributeToHost(1, price.data());
    blk->copyAttributeToHost(2, limit.data());
    EXPECT_TRUE(blk->nullBitmap(2) != nullptr);
    blk->copyNullBitmapToHost(2, nulls.data());
    for (std::size_t i = 0; i < k; ++i) {
      const bool is_null = (nulls[i >> 6] >> (63 - (i & 63))) & 1ull;   // TupleIdSequence order: first tuple = most significant bit
      g.push_back({key[i], price[i], is_null ? 0 : limit[i], is_null});
    }
  }
  for (std::size_t i = 0; i < li.orderkey.size(); ++i) {
    const std::int32_t k = li.orderkey[i];
    const bool has_order = k < 200000 && (k & 1) == 0;
    w.push_back({k, li.price[i], has_order ? k % 50 + 1 : 0, !has_order});
  }
  auto less = [](const Row &a, const Row &b) { return std::tie(a.key, a.price, a.limit, a.null) < std::tie(b.key, b.price, b.limit, b.null); };
  std::sort(g.begin(), g.end(), less);
  std::sort(w.begin(), w.end(), less);
  EXPECT_EQ(g.size(), w.size());
  bool same = g.size() == w.size();
  for (std::size_t i = 0; same && i < g.size(); ++i) same = g[i].key == w[i].key && g[i].price == w[i].price && g[i].limit == w[i].limit && g[i].null == w[i].null;
  EXPECT_TRUE(same);
}

// select o_orderkey, l_extendedprice, o_flag from orders join lineitem on o_orderkey = l_orderkey where o_flag = 'KEEP':
// a CHAR(10) attribute of the build side in the residual predicate (compared on the pair list by qsx_select_cmp_char) and in
// the projection (gathered byte by byte) — the run form and the block-by-block form
void runJoinCharResidual(std::size_t blocks_per_order, std::size_t *out_blocks) {
  StorageManager storage;
  Lineitem li(&storage, false);
  CatalogRelation orders(3, "orders");
  orders.addAttribute("o_orderkey", Type::Int());
  orders.addAttribute("o_flag", Type::Char(10));
  std::vector<std::int32_t> okeys;
  for (std::int32_t k = 0; k < 200000; k += 2) okeys.push_back(k);
  std::shuffle(okeys.begin(), okeys.end(), std::mt19937_64(5));
  std::vector<char> flags(okeys.size() * 10, 0);
  for (std::size_t i = 0; i < okeys.size(); ++i) std::strncpy(&flags[i * 10], okeys[i] % 3 == 0 ? "KEEP" : "KEEPER", 10);
  for (std::size_t at = 0; at < okeys.size(); at += 2500) storage.loadBlock(&orders, {okeys.data() + at, flags.data() + at * 10}, 2500);
  CatalogRelation out(4, "joined");
  out.addAttribute("o_orderkey", Type::Int());
  out.addAttribute("l_extendedprice", Type::Double());
  out.addAttribute("o_flag", Type::Char(10));
  QueryContext ctx;
  const auto table = ctx.addJoinHashTable(kInt, 100000);
  const auto dest = ctx.addInsertDestination(&out, &storage);
  const auto selection = ctx.addScalarGroup({0, 2, 1});
  const std::vector<bool> on_build = {true, false, true};
  Predicate residual;
  residual.conjuncts.push_back(ComparisonPredicate(1, ComparisonID::kEqual, TypedLiteral::Char("KEEP"), /*build_side=*/true));
  const auto pred = ctx.addPredicate(residual);
  BuildHashOperator builder(0, orders, true, {0}, false, 1, table);
  HashJoinOperator prober(0, orders, li.rel, true, {0}, false, 1, false, out, dest, table, pred, selection, &on_build,
                          HashJoinOperator::JoinType::kInnerJoin);
  prober.setBlocksPerWorkOrder(blocks_per_order);
  fetchAndExecuteWorkOrders(&builder, &ctx, &storage);
  fetchAndExecuteWorkOrders(&prober, &ctx, &storage);
  std::vector<std::pair<std::int32_t, double>> g, w;
  const std::vector<block_id> touched = ctx.getInsertDestination(dest)->getTouchedBlocks();
  *out_blocks = touched.size();
  bool flags_ok = true;
  for (block_id b : touched) {
    BlockReference blk = storage.getBlock(b);
    const std::size_t k = static_cast<std::size_t>(blk->numTuples());
    if (k == 0) continue;
    std::vector<std::int32_t> key(k);
    std::vector<double> price(k);
    std::vector<char> flag(k * 10);
    blk->copyAttributeToHost(0, key.data());
    blk->copyAttributeToHost(1, price.data());
    blk->copyAttributeToHost(2, flag.data());
    for (std::size_t i = 0; i < k; ++i) {
      g.emplace_back(key[i], price[i]);
      flags_ok = flags_ok && std::strncmp(&flag[i * 10], "KEEP", 10) == 0;
    }
  }
  for (std::size_t i = 0; i < li.orderkey.size(); ++i) {
    if (li.orderkey[i] < 200000 && (li.orderkey[i] & 1) == 0 && li.orderkey[i] % 3 == 0) w.emplace_back(li.orderkey[i], li.price[i]);
  }
  std::sort(g.begin(), g.end());
  std::sort(w.begin(), w.end());
  EXPECT_TRUE(flags_ok);
  EXPECT_TRUE(w.size() > 10000);
  EXPECT_EQ(g.size(), w.size());
  EXPECT_TRUE(g == w);
}

// select l_orderkey * 2 + l_quantity [INT], l_orderkey * 5000000000 - l_quantity [LONG], l_quantity / 7 [INT, truncating],
//        l_extendedprice * l_quantity [DOUBLE] from lineitem where l_quantity < 24 — integer operands in integer arithmetic
// NULLs on the probe side — what every outer-join chain hands to the next join: a nullable join key (NULL in every 11th row of
// the odd blocks; the even blocks carry no bitmap at all) and a nullable output attribute (NULL in every 17th row).  A tuple
// with a NULL key is not looked up (HashTable.hpp:2158-2160): out of an inner / semi join, kept by an anti join, NULL-padded
// by an outer join; the null bits of the projected probe attribute follow the output tuples.  Tuples as (quantity or -1 for
// NULL, second attribute or a marker for NULL).
void runNullableProbeSide(HashJoinOperator::JoinType type, bool exact_stats, std::size_t blocks_per_order, std::size_t *out_blocks) {
  using JoinType = HashJoinOperator::JoinType;
  StorageManager storage;
  CatalogRelation li(1, "lineitem"), orders(3, "orders"), out(4, "joined");
  li.addAttribute("l_orderkey", Type::Int().getNullableVersion());
  li.addAttribute("l_quantity", Type::Int().getNullableVersion());
  li.addAttribute("l_extendedprice", Type::Double());
  orders.addAttribute("o_orderkey", Type::Int());
  const bool existence = type == JoinType::kLeftSemiJoin || type == JoinType::kLeftAntiJoin;
  out.addAttribute("l_quantity", Type::Int().getNullableVersion());
  if (existence) out.addAttribute("l_extendedprice", Type::Double());
  else out.addAttribute("o_orderkey", type == JoinType::kLeftOuterJoin ? Type::Int().getNullableVersion() : Type::Int());
  std::vector<std::pair<std::int64_t, std::int64_t>> want;
  constexpr std::int64_t kNullMark = -7777777;
  std::mt19937_64 rng(23);
  for (int b = 0; b < kBlocks; ++b) {
    const std::int64_t n = b == 5 ? 0 : kBlockRows - 13 * (b % 7);
    std::vector<std::int32_t> k(n), q(n);
    std::vector<double> p(n);
    std::vector<std::uint64_t> key_nulls(static_cast<std::size_t>((n + 63) / 64) + 1, 0), qty_nulls(key_nulls.size(), 0);
    for (std::int64_t i = 0; i < n; ++i) {
      k[i] = static_cast<std::int32_t>(rng() % 300000);
      q[i] = static_cast<std::int32_t>(rng() % 50) + 1;
      p[i] = static_cast<double>(rng() % 10000000);
      const bool key_null = (b & 1) != 0 && i % 11 == 0, qty_null = i % 17 == 0;
      if (key_null) key_nulls[i >> 6] |= 1ull << (63 - (i & 63));
      if (qty_null) qty_nulls[i >> 6] |= 1ull << (63 - (i & 63));
      const bool matched = !key_null && k[i] < 200000 && (k[i] & 1) == 0;
      const std::int64_t qty = qty_null ? -1 : q[i];
      switch (type) {
        case JoinType::kInnerJoin: if (matched) want.emplace_back(qty, k[i]); break;
        case JoinType::kLeftSemiJoin: if (matched) want.emplace_back(qty, static_cast<std::int64_t>(p[i])); break;
        case JoinType::kLeftAntiJoin: if (!matched) want.emplace_back(qty, static_cast<std::int64_t>(p[i])); break;
        default: want.emplace_back(qty, matched ? k[i] : kNullMark); break;
      }
    }
    const std::vector<const std::uint64_t *> null_bitmaps = {(b & 1) != 0 ? key_nulls.data() : nullptr, qty_nulls.data(), nullptr};
    storage.loadBlock(&li, {k.data(), q.data(), p.data()}, n, 0, nullptr, &null_bitmaps);
  }
  std::vector<std::int32_t> okeys;
  for (std::int32_t k = 0; k < 200000; k += 2) okeys.push_back(k);
  std::shuffle(okeys.begin(), okeys.end(), std::mt19937_64(3));
  for (std::size_t at = 0; at < okeys.size(); at += 2500) storage.loadBlock(&orders, {okeys.data() + at}, 2500);
  QueryContext ctx;
  const QueryContext::ExactKeyRange range{0, 199998};
  const auto table = ctx.addJoinHashTable(kInt, 100000, 1, exact_stats ? &range : nullptr);
  const auto dest = ctx.addInsertDestination(&out, &storage);
  const auto selection = ctx.addScalarGroup(existence ? std::vector<attribute_id>{1, 2} : std::vector<attribute_id>{1, 0});
  const std::vector<bool> on_build = existence ? std::vector<bool>{false, false} : std::vector<bool>{false, true};
  BuildHashOperator builder(0, orders, true, {0}, false, 1, table);
  HashJoinOperator prober(0, orders, li, true, {0}, true, 1, false, out, dest, table, QueryContext::kInvalidPredicateId, selection, &on_build, type);
  prober.setBlocksPerWorkOrder(blocks_per_order);
  fetchAndExecuteWorkOrders(&builder, &ctx, &storage);
  fetchAndExecuteWorkOrders(&prober, &ctx, &storage);
  std::vector<std::pair<std::int64_t, std::int64_t>> got;
  const std::vector<block_id> touched = ctx.getInsertDestination(dest)->getTouchedBlocks();
  *out_blocks = touched.size();
  for (block_id id : touched) {
    BlockReference blk = storage.getBlock(id);
    const std::size_t n = static_cast<std::size_t>(blk->numTuples());
    if (n == 0) continue;
    std::vector<std::int32_t> qty(n), key(n);
    std::vector<double> price(n);
    std::vector<std::uint64_t> qn((n + 63) / 64), kn((n + 63) / 64);
    blk->copyAttributeToHost(0, qty.data());
    blk->copyNullBitmapToHost(0, qn.data());
    if (existence) {
      blk->copyAttributeToHost(1, price.data());
    } else {
      blk->copyAttributeToHost(1, key.data());
      blk->copyNullBitmapToHost(1, kn.data());
    }
    for (std::size_t i = 0; i < n; ++i) {
      const bool qty_null = ((qn[i >> 6] >> (63 - (i & 63))) & 1u) != 0, key_null = ((kn[i >> 6] >> (63 - (i & 63))) & 1u) != 0;
      got.emplace_back(qty_null ? -1 : qty[i], existence ? static_cast<std::int64_t>(price[i]) : (key_null ? kNullMark : key[i]));
    }
  }
  std::sort(got.begin(), got.end());
  std::sort(want.begin(), want.end());
  EXPECT_EQ(got.size(), want.size());
  EXPECT_TRUE(got == want);
}

// Semi / anti joins over a HASHED composite key: (LONG, LONG) does not fit 8 bytes, the table is keyed by the reference's
// CombineHashes fold and a tuple only counts as matched when the components of a pair are equal (compositeKeyCollisionCheck,
// SeparateChainingHashTable.hpp:1046) — so the run form goes through the pair list and the component equalities, not
// through an existence probe.  Build: (a, b) for a in [0, 400), b = 7 a mod 13 and b + 13 (two rows per a); probe rows pick
// a in [0, 600) and b in [0, 26): few match by both components.
void runCompositeSemiAnti(bool anti, std::size_t blocks_per_order, std::size_t *out_blocks) {
  StorageManager storage;
  CatalogRelation probe(1, "probe"), build(3, "build"), out(4, "kept");
  probe.addAttribute("a", Type::Long());
  probe.addAttribute("b", Type::Long());
  probe.addAttribute("v", Type::Double());
  build.addAttribute("a", Type::Long());
  build.addAttribute("b", Type::Long());
  out.addAttribute("a", Type::Long());
  out.addAttribute("v", Type::Double());
  std::vector<std::pair<std::int64_t, double>> want;
  std::mt19937_64 rng(29);
  for (int blk = 0; blk < kBlocks; ++blk) {
    const std::int64_t n = blk == 5 ? 0 : 4000 - 13 * (blk % 7);
    std::vector<std::int64_t> a(n), b(n);
    std::vector<double> v(n);
    for (std::int64_t i = 0; i < n; ++i) {
      a[i] = static_cast<std::int64_t>(rng() % 600) * 1000003;       // (wide values: the pair does not pack into 8 bytes)
      b[i] = static_cast<std::int64_t>(rng() % 26) * 1000003;
      v[i] = static_cast<double>(rng() % 100000);
      const std::int64_t ka = a[i] / 1000003, kb = b[i] / 1000003;
      const bool matched = ka < 400 && (kb == (7 * ka) % 13 || kb == (7 * ka) % 13 + 13);
      if (matched != anti) want.emplace_back(a[i], v[i]);
    }
    storage.loadBlock(&probe, {a.data(), b.data(), v.data()}, n);
  }
  std::vector<std::int64_t> ba, bb;
  for (std::int64_t ka = 0; ka < 400; ++ka) {
    for (std::int64_t kb : {(7 * ka) % 13, (7 * ka) % 13 + 13}) {
      ba.push_back(ka * 1000003);
      bb.push_back(kb * 1000003);
    }
  }
  for (std::size_t at = 0; at < ba.size(); at += 100) storage.loadBlock(&build, {ba.data() + at, bb.data() + at}, 100);
  QueryContext ctx;
  const auto table = ctx.addJoinHashTable(kLong, 1000);
  const auto dest = ctx.addInsertDestination(&out, &storage);
  const auto selection = ctx.addScalarGroup({0, 2});
  const std::vector<bool> on_build = {false, false};
  BuildHashOperator builder(0, build, true, {0, 1}, false, 1, table);
  HashJoinOperator prober(0, build, probe, true, {0, 1}, false, 1, false, out, dest, table, QueryContext::kInvalidPredicateId, selection, &on_build,
                          anti ? HashJoinOperator::JoinType::kLeftAntiJoin : HashJoinOperator::JoinType::kLeftSemiJoin);
  prober.setBlocksPerWorkOrder(blocks_per_order);
  builder.setBlocksPerWorkOrder(blocks_per_order);
  fetchAndExecuteWorkOrders(&builder, &ctx, &storage);
  fetchAndExecuteWorkOrders(&prober, &ctx, &storage);
  std::vector<std::pair<std::int64_t, double>> got;
  const std::vector<block_id> touched = ctx.getInsertDestination(dest)->getTouchedBlocks();
  *out_blocks = touched.size();
  for (block_id id : touched) {
    BlockReference blk = storage.getBlock(id);
    const std::size_t n = static_cast<std::size_t>(blk->numTuples());
    if (n == 0) continue;
    std::vector<std::int64_t> a(n);
    std::vector<double> v(n);
    blk->copyAttributeToHost(0, a.data());
    blk->copyAttributeToHost(1, v.data());
    for (std::size_t i = 0; i < n; ++i) got.emplace_back(a[i], v[i]);
  }
  std::sort(got.begin(), got.end());
  std::sort(want.begin(), want.end());
  EXPECT_TRUE(want.size() > 1000);
  EXPECT_EQ(got.size(), want.size());
  EXPECT_TRUE(got == want);
}

void runTypedExpressions() {
  StorageManager storage;
  Lineitem li(&storage, false);
  CatalogRelation out(2, "projected");
  out.addAttribute("a", Type::Int());
  out.addAttribute("b", Type::Long());
  out.addAttribute("c", Type::Int());
  out.addAttribute("d", Type::Double());
  QueryContext ctx;
  Predicate p;
  p.conjuncts.push_back(ComparisonPredicate(1, ComparisonID::kLess, TypedLiteral::Int(24)));
  const auto pred = ctx.addPredicate(p);
  const auto dest = ctx.addInsertDestination(&out, &storage);
  using B = BinaryOperationID;
  std::vector<ScalarPtr> selection = {
      Scalar::Binary(B::kAdd, Scalar::Binary(B::kMultiply, Scalar::Attribute(0), Scalar::IntLiteral(2)), Scalar::Attribute(1)),
      Scalar::Binary(B::kSubtract, Scalar::Binary(B::kMultiply, Scalar::Attribute(0), Scalar::IntLiteral(5000000000ll)), Scalar::Attribute(1)),
      Scalar::Binary(B::kDivide, Scalar::Attribute(1), Scalar::IntLiteral(7)),
      Scalar::Binary(B::kMultiply, Scalar::Attribute(2), Scalar::Attribute(1))};
  EXPECT_EQ(ScalarResultType(selection[0], li.rel), kInt);
  EXPECT_EQ(ScalarResultType(selection[1], li.rel), kLong);
  EXPECT_EQ(ScalarResultType(selection[3], li.rel), kDouble);
  SelectOperator op(0, li.rel, false, out, dest, pred, std::move(selection), true);
  fetchAndExecuteWorkOrders(&op, &ctx, &storage);
  std::vector<std::int32_t> a, c;
  std::vector<std::int64_t> b;
  std::vector<double> d;
  for (block_id id : ctx.getInsertDestination(dest)->getTouchedBlocks()) {
    BlockReference blk = storage.getBlock(id);
    const std::size_t at = a.size(), k = static_cast<std::size_t>(blk->numTuples());
    a.resize(at + k); b.resize(at + k); c.resize(at + k); d.resize(at + k);
    if (k == 0) continue;
    blk->copyAttributeToHost(0, a.data() + at); blk->copyAttributeToHost(1, b.data() + at);
    blk->copyAttributeToHost(2, c.data() + at); blk->copyAttributeToHost(3, d.data() + at);
  }
  std::size_t at = 0;
  bool same = true;
  for (std::size_t i = 0; i < li.orderkey.size(); ++i) {
    if (li.quantity[i] >= 24) continue;
    if (at >= a.size()) { same = false; break; }
    same = same && a[at] == li.orderkey[i] * 2 + li.quantity[i] &&
           b[at] == static_cast<std::int64_t>(li.orderkey[i]) * 5000000000ll - li.quantity[i] && c[at] == li.quantity[i] / 7 &&
           d[at] == li.price[i] * static_cast<double>(li.quantity[i]);
    ++at;
  }
  EXPECT_EQ(at, a.size());
  EXPECT_TRUE(same);
  // an output attribute of the wrong type for its expression is refused
  CatalogRelation wrong(3, "wrong");
  wrong.addAttribute("a", Type::Double());
  const auto wrong_dest = ctx.addInsertDestination(&wrong, &storage);
  SelectOperator bad(0, li.rel, false, wrong, wrong_dest, pred,
                     std::vector<ScalarPtr>{Scalar::Binary(B::kAdd, Scalar::Attribute(0), Scalar::Attribute(1))}, true);
  bool threw = false;
  try {
    fetchAndExecuteWorkOrders(&bad, &ctx, &storage);
  } catch (const ExecutionError &e) {
    threw = e.status() == QSX_ERR_INVALID_ARGUMENT;
  }
  EXPECT_TRUE(threw);
}
}  // namespace

int main() {
  if (qsx_device_count() < 1) {
    std::fprintf(stderr, "work_order_runs_test needs an MI355X: %s\n", qsx_status_string(QSX_ERR_NO_DEVICE));
    return 2;
  }
  std::size_t blocks_one = 0, blocks_run = 0;
  double ms_one = 0, ms_run = 0;
  for (int rep = 0; rep < 2; ++rep) {   // second repetition: warm allocator / modules
    runSelect(false, 1, &blocks_one, &ms_one);
    runSelect(false, 64, &blocks_run, &ms_run);
  }
  EXPECT_EQ(blocks_one, static_cast<std::size_t>(kBlocks));
  EXPECT_EQ(blocks_run, static_cast<std::size_t>((kBlocks + 63) / 64));      // one output block per run
  std::printf("select over %d blocks of ~%lld rows: one work order per block %.2f ms, per run of 64 blocks %.2f ms\n", kBlocks,
              static_cast<long long>(kBlockRows), ms_one, ms_run);
  // a nullable predicate attribute: the run work order executes its blocks one by one
  runSelect(true, 1, &blocks_one, &ms_one);
  runSelect(true, 64, &blocks_run, &ms_run);
  EXPECT_EQ(blocks_run, static_cast<std::size_t>(kBlocks));
  for (const bool exact_stats : {true, false}) {
    for (int rep = 0; rep < 2; ++rep) {
      runJoin(exact_stats, 1, &blocks_one, &ms_one);
      runJoin(exact_stats, 64, &blocks_run, &ms_run);
    }
    EXPECT_EQ(blocks_one, static_cast<std::size_t>(kBlocks));
    EXPECT_EQ(blocks_run, static_cast<std::size_t>((kBlocks + 63) / 64));
    double build_one = 0, build_run = 0;
    for (int rep = 0; rep < 2; ++rep) {
      runJoin(exact_stats, 1, &blocks_one, &ms_one);
      build_one = g_build_ms;
      runJoin(exact_stats, 64, &blocks_run, &ms_run);
      build_run = g_build_ms;
    }
    std::printf("hash join (%s table) probing %d blocks: one work order per block %.2f ms, per run of 64 blocks %.2f ms; "
                "building from 40 blocks: %.2f ms / %.2f ms\n",
                exact_stats ? "directly addressed" : "hashed", kBlocks, ms_one, ms_run, build_one, build_run);
  }
  // blocks sorted on l_quantity: the first predicate term is a binary search per block — on the values, or on the code
  // stripe of the compressed sort column (the reference's TPC-H layout: lineitem SORT l_shipdate, COMPRESS ALL)
  for (const int sorted_mode : {1, 2}) {
    runSelect(false, 1, &blocks_one, &ms_one, false, sorted_mode);
    runSelect(false, 64, &blocks_run, &ms_run, false, sorted_mode);
    EXPECT_EQ(blocks_run, static_cast<std::size_t>((kBlocks + 63) / 64));
    std::printf("select on the %ssort column: one work order per block %.2f ms, per run of 64 blocks %.2f ms\n",
                sorted_mode == 2 ? "compressed " : "", ms_one, ms_run);
  }
  // LIP filters in the run forms: one qsx_lip_probe_blocks per filter in front of the predicate terms / the probe
  runSelect(false, 1, &blocks_one, &ms_one, true);
  runSelect(false, 64, &blocks_run, &ms_run, true);
  EXPECT_EQ(blocks_run, static_cast<std::size_t>((kBlocks + 63) / 64));
  std::printf("select under a LIP filter: one work order per block %.2f ms, per run of 64 blocks %.2f ms\n", ms_one, ms_run);
  runJoin(true, 1, &blocks_one, &ms_one, true);
  runJoin(true, 64, &blocks_run, &ms_run, true);
  EXPECT_EQ(blocks_run, static_cast<std::size_t>((kBlocks + 63) / 64));
  std::printf("hash join under a LIP filter: one work order per block %.2f ms, per run of 64 blocks %.2f ms\n", ms_one, ms_run);
  runTypedExpressions();
  // left outer join: block by block and over runs
  {
    std::size_t one = 0, run = 0;
    runOuterJoin(true, 1, &one);
    runOuterJoin(true, 64, &run);
    runOuterJoin(false, 64, &run);
    EXPECT_EQ(one, static_cast<std::size_t>(kBlocks));
    EXPECT_EQ(run, static_cast<std::size_t>((kBlocks + 63) / 64));
  }
  // semi / anti joins with a residual predicate: block by block and over runs (ragged blocks: word-aligned tuple ids)
  for (bool anti : {false, true}) {
    std::size_t one = 0, run = 0;
    runSemiAntiResidual(anti, true, 1, &one);
    runSemiAntiResidual(anti, true, 64, &run);
    runSemiAntiResidual(anti, false, 64, &run);
    EXPECT_EQ(one, static_cast<std::size_t>(kBlocks));
    EXPECT_EQ(run, static_cast<std::size_t>((kBlocks + 63) / 64));
  }
  // NULL join keys and a nullable projected attribute on the probe side: every join type, both table kinds, both forms —
  // the run form covers them (one output block per run), it no longer hands these shapes to the block-by-block path
  for (const HashJoinOperator::JoinType type : {HashJoinOperator::JoinType::kInnerJoin, HashJoinOperator::JoinType::kLeftSemiJoin,
                                                HashJoinOperator::JoinType::kLeftAntiJoin, HashJoinOperator::JoinType::kLeftOuterJoin}) {
    for (const bool exact_stats : {true, false}) {
      std::size_t one = 0, run = 0;
      runNullableProbeSide(type, exact_stats, 1, &one);
      runNullableProbeSide(type, exact_stats, 64, &run);
      EXPECT_EQ(one, static_cast<std::size_t>(kBlocks));
      EXPECT_EQ(run, static_cast<std::size_t>((kBlocks + 63) / 64));
    }
  }
  // an anti / semi join with a residual predicate UNDER A LIP FILTER, and semi / anti joins over a hashed composite key:
  // run forms since round 4
  for (bool anti : {false, true}) {
    std::size_t one = 0, run = 0;
    runSemiAntiResidual(anti, true, 1, &one, /*with_lip=*/true);
    runSemiAntiResidual(anti, true, 64, &run, /*with_lip=*/true);
    EXPECT_EQ(one, static_cast<std::size_t>(kBlocks));
    EXPECT_EQ(run, static_cast<std::size_t>((kBlocks + 63) / 64));
    runCompositeSemiAnti(anti, 1, &one);
    runCompositeSemiAnti(anti, 64, &run);
    EXPECT_EQ(one, static_cast<std::size_t>(kBlocks));
    EXPECT_EQ(run, static_cast<std::size_t>((kBlocks + 63) / 64));
  }
  // duplicate build keys: the projecting probe overflows its block and the work order falls back to the pair list
  for (bool exact_stats : {true, false}) {
    runJoinDuplicateBuildKeys(exact_stats, 1);
    runJoinDuplicateBuildKeys(exact_stats, 64);
  }
  // a CHAR(10) build attribute in the residual predicate and in the projection: both forms
  runJoinCharResidual(1, &blocks_one);
  runJoinCharResidual(64, &blocks_run);
  EXPECT_EQ(blocks_one, static_cast<std::size_t>(kBlocks));
  EXPECT_EQ(blocks_run, static_cast<std::size_t>((kBlocks + 63) / 64));
  return finish("work_order_runs_test");
}
