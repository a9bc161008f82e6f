// The headline workload of bench.py (BASELINE.json configs 2 + 3) through the OPERATOR BOUNDARY instead of the raw C ABI:
//
//   BuildHashOperator(customer.c_custkey)  --breaker-->  HashJoinOperator(orders.o_custkey = c_custkey) -> output relation
//   AggregationOperator(lineitem: Q1's GROUP BY l_returnflag, l_linestatus + its eight aggregates)
//       --breaker--> FinalizeAggregationOperator -> output relation
//   DestroyHashOperator, DestroyAggregationStateOperator
//
// scheduled by ForemanSingleNode on process-wide Worker threads (query_execution/Worker.cpp:119-148), the relations stored
// as reference-sized 4 MB column-store blocks (1 M INT rows; 123 361 rows of the six Q1 attributes), work orders over runs
// of blocks (setBlocksPerWorkOrder, RelationalOperator.hpp:117-119).  One step = one such query; the operators allocate
// their join table, aggregation state and output blocks inside the step like the reference's query admission does.
//
// The blocks are loaded from a handful of host templates (no 20 GB host copy of lineitem): every template block recurs
// n / templates times, so the expected result of any number of blocks is known exactly (counts) or to rounding (sums).
//
// lineitem_store = 1: lineitem arrives as the reference stores it (benchmarks/tpch/create.sql:69-121): CompressedColumnStore blocks
// sorted on l_shipdate, l_quantity / l_discount / l_tax as 1-byte dictionary codes with a dictionary per block — block IMAGES in
// the reference's own layout (tests/cpp/block_image_util.hpp) adopted where they lie in device memory
// (StorageManager::adoptBlockImage), the stripes at whatever byte offsets the layout gives them.  The aggregation then reads
// 13 instead of 34 bytes per row and its aggregates are factored through the dictionary codes (csrc/agg_factored.hpp).
//
// usage: headline_operators_bench [build_rows probe_rows agg_rows [steps warmup workers blocks_per_work_order [lineitem_store]]]
// prints one JSON line {"rows_per_s": ..., "ms_per_step": ..., ...}; exit code 0 only when every step's results check out.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>

#include "block_image_util.hpp"
#include "test_util.hpp"

using namespace quickstep;

extern "C" long long qsx_debug_agg_factored_launches(void);   // test hook of libqsx.so (aggregate.hip)

namespace {
constexpr std::int64_t kBlockBytes = 4ll << 20;
constexpr int kTemplates = 8;

struct Xorshift {
  std::uint64_t x;
  explicit Xorshift(std::uint64_t seed) : x(seed * 0x9E3779B97F4A7C15ull + 1) {}
  std::uint64_t next() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; }
};

struct Q1Template {
  std::vector<char> flag, status;
  std::vector<double> qty, price, disc, tax;
  std::map<std::pair<char, char>, std::int64_t> count;
  std::map<std::pair<char, char>, double> sum_qty, sum_price, sum_disc_price;
};
}  // namespace

int main(int argc, char **argv) {
  if (qsx_device_count() < 1) {
    std::fprintf(stderr, "headline_operators_bench needs an MI355X: %s\n", qsx_status_string(QSX_ERR_NO_DEVICE));
    return 2;
  }
  const std::int64_t build_rows = argc > 1 ? std::atoll(argv[1]) : 1000000, probe_rows = argc > 2 ? std::atoll(argv[2]) : 100000000,
                     agg_rows = argc > 3 ? std::atoll(argv[3]) : 600000000;
  const int steps = argc > 4 ? std::atoi(argv[4]) : 5, warmup = argc > 5 ? std::atoi(argv[5]) : 2;
  const std::size_t workers = argc > 6 ? static_cast<std::size_t>(std::atoi(argv[6])) : 4;
  const std::size_t run_blocks = argc > 7 ? static_cast<std::size_t>(std::atoi(argv[7])) : 64;
  const int lineitem_store = argc > 8 ? std::atoi(argv[8]) : 0;
  const bool compressed_lineitem = lineitem_store == 1 || lineitem_store == 2;
  // lineitem_store = 2: the block's sort column is l_orderkey, l_shipdate is one more dictionary-coded attribute (2-byte codes, a
  // dictionary per block: COMPRESS ALL) and Q1's predicate l_shipdate <= DATE sits inside the aggregation, as in the reference's
  // plan — scanned on the code stripes (rewritten on every block's own dictionary), the aggregation takes the TupleIdSequence.
  const bool q1_predicate = lineitem_store == 2;
  const std::int32_t shipdate_cutoff = 19920101 + 2475;   // of 19920101 .. 19920101 + 2525: ~98 % of the tuples, as in Q1

  StorageManager storage;
  CatalogRelation customer(1, "customer"), orders(2, "orders"), lineitem(3, "lineitem");
  customer.addAttribute("c_custkey", Type::Int());
  orders.addAttribute("o_custkey", Type::Int());
  lineitem.addAttribute("l_returnflag", Type::Char(1));
  lineitem.addAttribute("l_linestatus", Type::Char(1));
  for (const char *n : {"l_quantity", "l_extendedprice", "l_discount", "l_tax"}) lineitem.addAttribute(n, Type::Double());
  if (compressed_lineitem) lineitem.addAttribute("l_shipdate", Type::Int());   // store 1: the block's sort column
  if (q1_predicate) lineitem.addAttribute("l_orderkey", Type::Int());          // store 2: the block's sort column (create.sql: SORT l_orderkey)

  // ---- customer: a permutation of [0, build_rows) in 4 MB blocks ----------------------------------------------------------
  const std::int64_t int_block = kBlockBytes / 4;
  {
    std::vector<std::int32_t> keys(static_cast<std::size_t>(build_rows));
    for (std::int64_t i = 0; i < build_rows; ++i) keys[i] = static_cast<std::int32_t>(i);
    Xorshift rng(2);
    for (std::int64_t i = build_rows - 1; i > 0; --i) std::swap(keys[i], keys[rng.next() % static_cast<std::uint64_t>(i + 1)]);
    for (std::int64_t at = 0; at < build_rows; at += int_block) {
      storage.loadBlock(&customer, {keys.data() + at}, std::min(int_block, build_rows - at));
    }
  }
  // ---- orders: uniform foreign keys, kTemplates distinct blocks --------------------------------------------------------------
  {
    std::vector<std::vector<std::int32_t>> tmpl(kTemplates, std::vector<std::int32_t>(static_cast<std::size_t>(int_block)));
    Xorshift rng(3);
    for (auto &t : tmpl) for (auto &k : t) k = static_cast<std::int32_t>(rng.next() % static_cast<std::uint64_t>(build_rows));
    int b = 0;
    for (std::int64_t at = 0; at < probe_rows; at += int_block, ++b) {
      storage.loadBlock(&orders, {tmpl[b % kTemplates].data()}, std::min(int_block, probe_rows - at));
    }
  }
  // ---- lineitem: Q1 attributes, 4 MB = 123 361 rows of 34 bytes (plain) -----------------------------------------------------
  // compressed: the rows a 4 MB CompressedColumnStore block holds at 13 + 4 bytes per tuple behind its header and dictionaries
  const int image_attrs = q1_predicate ? 8 : 7, sort_attr = q1_predicate ? 7 : 6;
  std::vector<block_image::Coding> coding(static_cast<std::size_t>(image_attrs));
  for (int a : {2, 4, 5}) {
    coding[a].kind = block_image::Coding::kDictionary;
    coding[a].code_width = 1;
  }
  if (q1_predicate) {
    coding[6].kind = block_image::Coding::kDictionary;
    coding[6].code_width = 2;
  }
  std::int64_t q1_block = kBlockBytes / 34;
  std::vector<void *> device_images;
  if (compressed_lineitem) {   // (the capacity depends on the dictionaries' sizes only: ask the builder with a small block)
    Q1Template t;
    std::vector<std::int32_t> shipdate, orderkey;
    for (int i = 0; i < 4000; ++i) {
      t.flag.push_back('A'); t.status.push_back('F'); t.qty.push_back(1.0 + i % 50); t.price.push_back(1000.0 + i);
      t.disc.push_back((i % 11) / 100.0); t.tax.push_back((i % 9) / 100.0); shipdate.push_back(q1_predicate ? 19920101 + i % 2526 : i);
      orderkey.push_back(i);
    }
    std::int64_t capacity = 0;
    std::vector<const void *> columns = {t.flag.data(), t.status.data(), t.qty.data(), t.price.data(), t.disc.data(), t.tax.data(), shipdate.data()};
    if (q1_predicate) columns.push_back(orderkey.data());
    (void)block_image::BuildCompressed(lineitem, columns, std::vector<std::vector<bool>>(static_cast<std::size_t>(image_attrs)), 4000, kBlockBytes, sort_attr,
                                       coding, &capacity);
    q1_block = capacity;
  }
  std::vector<Q1Template> q1(kTemplates);
  std::vector<std::int32_t> shipdate_of_row;   // l_shipdate of row i of every block (the templates share the column)
  std::vector<std::int64_t> q1_uses(kTemplates, 0);
  std::int64_t last_block_rows = 0;
  int last_template = 0;
  {
    Xorshift rng(4);
    for (Q1Template &t : q1) {
      for (std::int64_t i = 0; i < q1_block; ++i) {
        const double u = static_cast<double>(rng.next() % 1000000) / 1e6;
        const char f = u < 0.2466 ? 'A' : (u < 0.2531 ? 'N' : (u < 0.7536 ? 'N' : 'R'));
        const char s = u < 0.2466 ? 'F' : (u < 0.2531 ? 'F' : (u < 0.7536 ? 'O' : 'F'));
        t.flag.push_back(f);
        t.status.push_back(s);
        t.qty.push_back(static_cast<double>(1 + rng.next() % 50));
        t.price.push_back(std::round((900.0 + static_cast<double>(rng.next() % 10410000) / 100.0) * 100.0) / 100.0);
        t.disc.push_back(static_cast<double>(rng.next() % 11) / 100.0);
        t.tax.push_back(static_cast<double>(rng.next() % 9) / 100.0);
      }
    }
    // compressed: one image per template in device memory; every block of the relation is a device copy of its template's
    std::vector<void *> template_images;
    std::vector<std::int32_t> shipdate(static_cast<std::size_t>(q1_block)), orderkey(static_cast<std::size_t>(q1_block));
    {
      Xorshift dates(9);
      for (std::int64_t i = 0; i < q1_block; ++i) {
        // store 1: ascending, the sort column; store 2: any of 2526 days (a 2-byte dictionary), the order keys ascend instead
        shipdate[i] = q1_predicate ? static_cast<std::int32_t>(19920101 + dates.next() % 2526) : static_cast<std::int32_t>(19920101 + i / 128);
        orderkey[i] = static_cast<std::int32_t>(i / 4);
      }
    }
    shipdate_of_row = shipdate;
    auto build_image = [&](const Q1Template &t, std::int64_t rows) {
      std::int64_t capacity = 0;
      std::vector<const void *> columns = {t.flag.data(), t.status.data(), t.qty.data(), t.price.data(), t.disc.data(), t.tax.data(), shipdate.data()};
      if (q1_predicate) columns.push_back(orderkey.data());
      const std::vector<unsigned char> image = block_image::BuildCompressed(
          lineitem, columns, std::vector<std::vector<bool>>(static_cast<std::size_t>(image_attrs)), rows, kBlockBytes, sort_attr, coding, &capacity);
      EXPECT_TRUE(capacity >= rows);
      void *dev = nullptr;
      CheckStatus(qsx_device_alloc(image.size(), &dev), "qsx_device_alloc");
      CheckStatus(qsx_copy_to_device(dev, image.data(), image.size(), nullptr), "qsx_copy_to_device");
      CheckStatus(qsx_stream_synchronize(nullptr), "qsx_stream_synchronize");
      return dev;
    };
    if (compressed_lineitem) for (const Q1Template &t : q1) template_images.push_back(build_image(t, q1_block));
    int b = 0;
    for (std::int64_t at = 0; at < agg_rows; at += q1_block, ++b) {
      const Q1Template &t = q1[b % kTemplates];
      const std::int64_t rows = std::min(q1_block, agg_rows - at);
      if (compressed_lineitem) {
        void *dev = nullptr;
        if (rows == q1_block) {
          CheckStatus(qsx_device_alloc(static_cast<std::size_t>(kBlockBytes), &dev), "qsx_device_alloc");
          CheckStatus(qsx_copy_on_device(dev, template_images[b % kTemplates], static_cast<std::size_t>(kBlockBytes), nullptr), "qsx_copy_on_device");
        } else {
          dev = build_image(t, rows);
        }
        device_images.push_back(dev);
        const block_id id = storage.adoptBlockImage(&lineitem, dev, static_cast<std::size_t>(kBlockBytes));
        EXPECT_EQ(storage.getBlock(id)->numTuples(), rows);
      } else {
        storage.loadBlock(&lineitem, {t.flag.data(), t.status.data(), t.qty.data(), t.price.data(), t.disc.data(), t.tax.data()}, rows);
      }
      if (rows == q1_block) {
        ++q1_uses[b % kTemplates];
      } else {
        last_block_rows = rows;
        last_template = b % kTemplates;
      }
    }
    CheckStatus(qsx_stream_synchronize(nullptr), "qsx_stream_synchronize");
    for (void *p : template_images) qsx_device_free(p);
    if (compressed_lineitem) {
      BlockReference first = storage.getBlock(lineitem.getBlocksSnapshot().front());
      EXPECT_TRUE(first->compressedAttribute(2) != nullptr && first->compressedAttribute(2)->kind == CompressedAttribute::kDictionary &&
                  first->compressedAttribute(2)->num_codes == 50 && first->compressedAttribute(4) != nullptr &&
                  first->compressedAttribute(4)->num_codes == 11 && first->compressedAttribute(5) != nullptr && first->compressedAttribute(5)->num_codes == 9);
      EXPECT_TRUE(first->compressedAttribute(3) == nullptr && first->sortColumn() == sort_attr);
      if (q1_predicate) {
        EXPECT_TRUE(first->compressedAttribute(6) != nullptr && first->compressedAttribute(6)->kind == CompressedAttribute::kDictionary &&
                    first->compressedAttribute(6)->code_width == 2);
      }
    }
  }
  // expected Q1 groups from the templates
  std::map<std::pair<char, char>, std::int64_t> want_count;
  std::map<std::pair<char, char>, double> want_qty, want_price, want_disc_price;
  for (int ti = 0; ti < kTemplates; ++ti) {
    const Q1Template &t = q1[ti];
    auto add = [&](std::int64_t rows, std::int64_t times) {
      std::map<std::pair<char, char>, std::int64_t> c;
      std::map<std::pair<char, char>, double> sq, sp, sd;
      for (std::int64_t i = 0; i < rows; ++i) {
        if (q1_predicate && shipdate_of_row[static_cast<std::size_t>(i)] > shipdate_cutoff) continue;
        const auto k = std::make_pair(t.flag[i], t.status[i]);
        c[k] += 1;
        sq[k] += t.qty[i];
        sp[k] += t.price[i];
        sd[k] += t.price[i] * (1.0 - t.disc[i]);
      }
      for (const auto &kv : c) {
        want_count[kv.first] += kv.second * times;
        want_qty[kv.first] += sq[kv.first] * static_cast<double>(times);
        want_price[kv.first] += sp[kv.first] * static_cast<double>(times);
        want_disc_price[kv.first] += sd[kv.first] * static_cast<double>(times);
      }
    };
    if (q1_uses[ti] > 0) add(q1_block, q1_uses[ti]);
    if (last_block_rows > 0 && ti == last_template) add(last_block_rows, 1);
  }

  CatalogRelation joined(10, "joined"), agg_out(11, "agg_out");
  joined.addAttribute("o_custkey", Type::Int());
  joined.addAttribute("c_custkey", Type::Int());
  agg_out.addAttribute("l_returnflag", Type::Char(1));
  agg_out.addAttribute("l_linestatus", Type::Char(1));
  for (const char *n : {"sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"}) agg_out.addAttribute(n, Type::Double());
  agg_out.addAttribute("count_order", Type::Long());

  double total_ms = 0.0, best_ms = 1e30;
  std::size_t work_orders = 0;
  const long long factored_before = qsx_debug_agg_factored_launches();
  for (int it = 0; it < warmup + steps; ++it) {
    const auto t0 = std::chrono::steady_clock::now();
    QueryContext ctx;
    const auto d_join = ctx.addInsertDestination(&joined, &storage), d_agg = ctx.addInsertDestination(&agg_out, &storage);
    const QueryContext::ExactKeyRange key_range{0, build_rows - 1};   // exact statistics of the primary key
    const auto table = ctx.addJoinHashTable(kInt, build_rows, 1, &key_range);
    const auto selection = ctx.addScalarGroup({0, 0});                  // o_custkey of the probe side, c_custkey of the build side
    const std::vector<bool> on_build{false, true};
    AggregationStateSpec spec;
    spec.input_relation = &lineitem;
    spec.group_by = {0, 1};
    const ScalarPtr disc_price = Scalar::Binary(BinaryOperationID::kMultiply, Scalar::Attribute(3),
                                                Scalar::Binary(BinaryOperationID::kSubtract, Scalar::Literal(1.0), Scalar::Attribute(4)));
    const ScalarPtr charge = Scalar::Binary(BinaryOperationID::kMultiply, disc_price,
                                            Scalar::Binary(BinaryOperationID::kAdd, Scalar::Literal(1.0), Scalar::Attribute(5)));
    spec.aggregates = {AggregateSpec(AggregationID::kSum, 2), AggregateSpec(AggregationID::kSum, 3), AggregateSpec(AggregationID::kSum, disc_price),
                       AggregateSpec(AggregationID::kSum, charge), AggregateSpec(AggregationID::kAvg, 2), AggregateSpec(AggregationID::kAvg, 3),
                       AggregateSpec(AggregationID::kAvg, 4), AggregateSpec(AggregationID::kCount, kInvalidAttributeID)};
    spec.strategy = QSX_AGG_COMPACT_KEY;
    spec.estimated_num_groups = 6;
    if (q1_predicate) {   // WHERE l_shipdate <= DATE: the aggregation operator's own predicate (benchmarks/tpch/queries/01.sql)
      Predicate where;
      where.conjuncts.push_back({6, ComparisonID::kLessOrEqual, TypedLiteral::Int(shipdate_cutoff)});
      spec.predicate = ctx.getPredicate(ctx.addPredicate(where));
    }
    const auto state = ctx.addAggregationState(spec);

    QueryPlan plan;
    BuildHashOperator *op_build = new BuildHashOperator(0, customer, true, {0}, false, 1, table);
    HashJoinOperator *op_join = new HashJoinOperator(0, customer, orders, true, {0}, false, 1, false, joined, d_join, table,
                                                     QueryContext::kInvalidPredicateId, selection, &on_build, HashJoinOperator::JoinType::kInnerJoin);
    AggregationOperator *op_agg = new AggregationOperator(0, lineitem, true, state);
    op_build->setBlocksPerWorkOrder(run_blocks);
    op_join->setBlocksPerWorkOrder(run_blocks);
    op_agg->setBlocksPerWorkOrder(run_blocks);
    const auto i_build = plan.addRelationalOperator(op_build);
    const auto i_join = plan.addRelationalOperator(op_join);
    const auto i_agg = plan.addRelationalOperator(op_agg);
    const auto i_fin = plan.addRelationalOperator(new FinalizeAggregationOperator(0, state, 1, false, 1, agg_out, d_agg));
    const auto i_drop_table = plan.addRelationalOperator(new DestroyHashOperator(0, 1, table));
    const auto i_drop_state = plan.addRelationalOperator(new DestroyAggregationStateOperator(0, state));
    plan.addDirectDependency(i_join, i_build, true);
    plan.addDirectDependency(i_fin, i_agg, true);
    plan.addDirectDependency(i_drop_table, i_join, true);
    plan.addDirectDependency(i_drop_state, i_fin, true);
    ForemanSingleNode foreman(&plan, &ctx, &storage, workers);
    foreman.run();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (it >= warmup) {
      total_ms += ms;
      best_ms = std::min(best_ms, ms);
    }
    if (std::getenv("QSX_BENCH_STEP_TIMES") != nullptr) std::fprintf(stderr, "step %d%s: %.3f ms\n", it, it < warmup ? " (warmup)" : "", ms);
    work_orders = foreman.getWorkOrderProfilingResults().size();
    if ((std::getenv("QSX_TEST_PROFILE") != nullptr && it == warmup + steps - 1) ||
        (std::getenv("QSX_BENCH_STEP_TIMES") != nullptr && it >= warmup && ms > 8.0)) {   // --profile_and_report_workorder_perf; a slow step
      const std::uint64_t t0_us = static_cast<std::uint64_t>(std::chrono::duration_cast<std::chrono::microseconds>(t0.time_since_epoch()).count());
      std::map<std::size_t, std::tuple<double, int, std::uint64_t, std::uint64_t>> per_op;
      for (const WorkOrderTimeEntry &e : foreman.getWorkOrderProfilingResults()) {
        auto &p = per_op[e.operator_index];
        if (std::get<1>(p) == 0) std::get<2>(p) = ~0ull;
        std::get<0>(p) += static_cast<double>(e.end_us - e.start_us) / 1e3;
        std::get<1>(p) += 1;
        std::get<2>(p) = std::min(std::get<2>(p), e.start_us);
        std::get<3>(p) = std::max(std::get<3>(p), e.end_us);
      }
      if (ms > 8.0) {
        for (const WorkOrderTimeEntry &e : foreman.getWorkOrderProfilingResults()) {
          std::fprintf(stderr, "    op %zu worker %zu: %+8.3f .. %+8.3f ms\n", e.operator_index, e.worker_id,
                       (static_cast<double>(e.start_us) - static_cast<double>(t0_us)) / 1e3, (static_cast<double>(e.end_us) - static_cast<double>(t0_us)) / 1e3);
        }
      }
      for (const auto &kv : per_op) {
        std::fprintf(stderr, "  %-34s %3d work orders, %7.3f ms summed, first start %+8.3f ms, last end %+8.3f ms\n",
                     plan.getOperator(kv.first)->getName().c_str(), std::get<1>(kv.second), std::get<0>(kv.second),
                     (static_cast<double>(std::get<2>(kv.second)) - static_cast<double>(t0_us)) / 1e3,
                     (static_cast<double>(std::get<3>(kv.second)) - static_cast<double>(t0_us)) / 1e3);
      }
      std::fprintf(stderr, "  step wall %.3f ms\n", ms);
    }

    // ---- results of this step ----------------------------------------------------------------------------------------------
    std::int64_t joined_rows = 0;
    bool join_ok = true;
    const bool verify_columns = it == warmup + steps - 1;   // the join condition on every output row: last step only
    for (block_id b : ctx.getInsertDestination(d_join)->getTouchedBlocks()) {
      BlockReference blk = storage.getBlock(b);
      joined_rows += blk->numTuples();
      if (verify_columns) {
        const std::size_t k = static_cast<std::size_t>(blk->numTuples());
        std::vector<std::int32_t> o(k), c(k);
        blk->copyAttributeToHost(0, o.data());
        blk->copyAttributeToHost(1, c.data());
        for (std::size_t i = 0; i < k; ++i) join_ok = join_ok && o[i] == c[i];
      }
      storage.deleteBlockOrBlobFile(b);
    }
    EXPECT_EQ(joined_rows, probe_rows);   // every foreign key has exactly one customer
    EXPECT_TRUE(join_ok);
    std::size_t groups = 0;
    for (block_id b : ctx.getInsertDestination(d_agg)->getTouchedBlocks()) {
      BlockReference blk = storage.getBlock(b);
      const std::size_t k = static_cast<std::size_t>(blk->numTuples());
      std::vector<char> f(k), s(k);
      std::vector<double> sq(k), sp(k), sd(k), aq(k);
      std::vector<std::int64_t> cnt(k);
      blk->copyAttributeToHost(0, f.data()); blk->copyAttributeToHost(1, s.data());
      blk->copyAttributeToHost(2, sq.data()); blk->copyAttributeToHost(3, sp.data()); blk->copyAttributeToHost(4, sd.data());
      blk->copyAttributeToHost(6, aq.data()); blk->copyAttributeToHost(9, cnt.data());
      for (std::size_t i = 0; i < k; ++i, ++groups) {
        const auto key = std::make_pair(f[i], s[i]);
        EXPECT_EQ(cnt[i], want_count[key]);                                   // COUNT(*): exact
        EXPECT_EQ(static_cast<std::int64_t>(sq[i]), static_cast<std::int64_t>(want_qty[key]));   // integer-valued doubles: exact
        EXPECT_NEAR(sp[i], want_price[key], 1e-6 * want_price[key]);
        EXPECT_NEAR(sd[i], want_disc_price[key], 1e-6 * want_disc_price[key]);
        EXPECT_NEAR(aq[i], want_qty[key] / static_cast<double>(want_count[key]), 1e-6 * aq[i]);
      }
      storage.deleteBlockOrBlobFile(b);
    }
    EXPECT_EQ(groups, want_count.size());
  }
  const double ms_per_step = total_ms / steps;
  std::printf("{\"path\": \"operators (BuildHash / HashJoin / Aggregation / FinalizeAggregation under ForemanSingleNode)\", "
              "\"rows_per_s\": %.6g, \"ms_per_step\": %.4f, \"best_ms\": %.4f, \"steps\": %d, \"warmup\": %d, \"workers\": %zu, "
              "\"blocks_per_work_order\": %zu, \"block_bytes\": %lld, \"probe_blocks\": %lld, \"aggregate_blocks\": %lld, \"work_orders_per_step\": %zu, "
              "\"build_rows\": %lld, \"probe_rows\": %lld, \"aggregate_rows\": %lld, \"lineitem_store\": \"%s\", \"lineitem_rows_per_block\": %lld, "
              "\"factored_aggregation_launches_per_step\": %.1f, \"checked\": %s}\n",
              static_cast<double>(probe_rows + agg_rows) / (ms_per_step / 1e3), ms_per_step, best_ms, steps, warmup, workers, run_blocks,
              static_cast<long long>(kBlockBytes), static_cast<long long>((probe_rows + int_block - 1) / int_block),
              static_cast<long long>((agg_rows + q1_block - 1) / q1_block), work_orders, static_cast<long long>(build_rows),
              static_cast<long long>(probe_rows), static_cast<long long>(agg_rows),
              q1_predicate ? "CompressedColumnStore images sorted on l_orderkey, l_shipdate 2-byte dictionary codes; Q1's l_shipdate <= DATE inside the aggregation (13 + 2 B/row)"
                           : (compressed_lineitem ? "CompressedColumnStore images (13 B/row aggregated, per-block dictionaries)" : "plain column stripes (34 B/row)"),
              static_cast<long long>(q1_block), static_cast<double>(qsx_debug_agg_factored_launches() - factored_before) / (warmup + steps),
              g_failures == 0 ? "true" : "false");
  return g_failures == 0 ? 0 : 1;
}
