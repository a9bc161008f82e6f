// Mirrors relational_operators/tests/SortRunGenerationOperator_unittest.cpp (1Column / 3Column, Asc / Desc, non-null:
// :442-478, :564-616) and SortMergeRunOperator_unittest.cpp (RunMergerTest 1Column / 3Column, Asc / Desc, with and
// without TopK: :923-963, :1049-1093): every run is sorted by the configuration; the merged output is the total order
// of the input (a permutation of it), truncated to top_k.  GPU work orders, synchronous driver and Foreman/Worker.
#include <algorithm>
#include <tuple>

#include "test_util.hpp"

using namespace quickstep;

namespace {
constexpr std::int64_t kRows = 60000;
constexpr std::int64_t kBlockRows = 7000;   // 9 runs, the last one short

struct Row {
  std::int32_t a;
  double b;
  std::int64_t c;
};

std::vector<Row> makeRows() {
  std::vector<Row> rows;
  std::uint64_t x = 1234567ull;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (std::int64_t i = 0; i < kRows; ++i) {
    rows.push_back({static_cast<std::int32_t>(rnd() % 200) - 100, static_cast<double>(rnd() % 1000) / 8.0 - 60.0,
                    static_cast<std::int64_t>(rnd() % 5) - 2});
  }
  return rows;
}

bool Before(const Row &x, const Row &y, const std::vector<attribute_id> &order_by, const std::vector<bool> &ordering) {
  for (std::size_t k = 0; k < order_by.size(); ++k) {
    int cmp = 0;
    switch (order_by[k]) {
      case 0: cmp = x.a < y.a ? -1 : (x.a > y.a ? 1 : 0); break;
      case 1: cmp = x.b < y.b ? -1 : (x.b > y.b ? 1 : 0); break;
      default: cmp = x.c < y.c ? -1 : (x.c > y.c ? 1 : 0); break;
    }
    if (!ordering[k]) cmp = -cmp;
    if (cmp != 0) return cmp < 0;
  }
  return false;
}

std::vector<Row> readRows(const std::vector<block_id> &blocks, StorageManager &storage, std::vector<std::size_t> *block_sizes) {
  std::vector<Row> out;
  for (block_id b : blocks) {
    BlockReference blk = storage.getBlock(b);
    const std::size_t k = static_cast<std::size_t>(blk->numTuples());
    std::vector<std::int32_t> a(k);
    std::vector<double> bb(k);
    std::vector<std::int64_t> c(k);
    blk->copyAttributeToHost(0, a.data()); blk->copyAttributeToHost(1, bb.data()); blk->copyAttributeToHost(2, c.data());
    for (std::size_t i = 0; i < k; ++i) out.push_back({a[i], bb[i], c[i]});
    if (block_sizes != nullptr) block_sizes->push_back(k);
  }
  return out;
}

// limit_runs: the LIMIT is handed to the run generation too (SortRunGenerationOperator::setTopK): runs of at most top_k tuples
void runCase(const std::vector<Row> &rows, const std::vector<attribute_id> &order_by, const std::vector<bool> &ordering,
             std::size_t top_k, bool use_foreman, bool limit_runs = false) {
  CatalogRelation input(1, "input"), runs(2, "runs"), output(3, "output");
  StorageManager storage;
  for (CatalogRelation *r : {&input, &runs, &output}) {
    r->addAttribute("a", Type::Int());
    r->addAttribute("b", Type::Double());
    r->addAttribute("c", Type::Long());
  }
  for (std::int64_t at = 0; at < kRows; at += kBlockRows) {
    const std::int64_t k = std::min(kBlockRows, kRows - at);
    std::vector<std::int32_t> a; std::vector<double> b; std::vector<std::int64_t> c;
    for (std::int64_t i = at; i < at + k; ++i) { a.push_back(rows[i].a); b.push_back(rows[i].b); c.push_back(rows[i].c); }
    storage.loadBlock(&input, {a.data(), b.data(), c.data()}, k);
  }
  QueryContext ctx;
  const auto config = ctx.addSortConfig({order_by, ordering});
  const auto run_dest = ctx.addInsertDestination(&runs, &storage);
  const auto out_dest = ctx.addInsertDestination(&output, &storage);
  auto *generate = new SortRunGenerationOperator(0, input, runs, run_dest, config, true);
  if (limit_runs) generate->setTopK(top_k);
  auto *merge = new SortMergeRunOperator(0, runs, output, out_dest, runs, run_dest, config, /*merge_factor=*/4, top_k, false);
  std::unique_ptr<RelationalOperator> g, m;
  if (use_foreman) {
    QueryPlan plan;
    const auto gi = plan.addRelationalOperator(generate);
    const auto mi = plan.addRelationalOperator(merge);
    plan.addDirectDependency(mi, gi, false);   // runs stream into the merge (SortMergeRunOperator::feedInputBlock)
    ForemanSingleNode foreman(&plan, &ctx, &storage, 4);
    foreman.run();
  } else {
    g.reset(generate); m.reset(merge);
    fetchAndExecuteWorkOrders(g.get(), &ctx, &storage);
    for (block_id b : ctx.getInsertDestination(run_dest)->getTouchedBlocks()) m->feedInputBlock(b, runs.getID(), 0);
    m->doneFeedingInputBlocks(runs.getID());
    fetchAndExecuteWorkOrders(m.get(), &ctx, &storage);
  }
  // every run is sorted (SortRunGenerationOperator_unittest: checkOutput per block)
  std::vector<std::size_t> run_sizes;
  const std::vector<Row> run_rows = readRows(ctx.getInsertDestination(run_dest)->getTouchedBlocks(), storage, &run_sizes);
  if (!limit_runs || top_k == 0) EXPECT_EQ(run_rows.size(), static_cast<std::size_t>(kRows));
  EXPECT_EQ(run_sizes.size(), static_cast<std::size_t>((kRows + kBlockRows - 1) / kBlockRows));
  if (limit_runs && top_k != 0) {
    for (std::size_t sz : run_sizes) EXPECT_TRUE(sz <= top_k);
  }
  std::size_t at = 0;
  for (std::size_t sz : run_sizes) {
    for (std::size_t i = at + 1; i < at + sz; ++i) EXPECT_TRUE(!Before(run_rows[i], run_rows[i - 1], order_by, ordering));
    at += sz;
  }
  // the merged output: total order, and exactly the first top_k tuples of the sorted input
  const std::vector<Row> out = readRows(ctx.getInsertDestination(out_dest)->getTouchedBlocks(), storage, nullptr);
  std::vector<Row> want = rows;
  std::stable_sort(want.begin(), want.end(), [&](const Row &x, const Row &y) { return Before(x, y, order_by, ordering); });
  const std::size_t expect_n = top_k != 0 && top_k < want.size() ? top_k : want.size();
  EXPECT_EQ(out.size(), expect_n);
  for (std::size_t i = 0; i < out.size() && i < expect_n; ++i) {
    // ties may come in any order among equal keys: compare the ORDER BY keys position by position ...
    EXPECT_TRUE(!Before(out[i], want[i], order_by, ordering) && !Before(want[i], out[i], order_by, ordering));
    if (i > 0) EXPECT_TRUE(!Before(out[i], out[i - 1], order_by, ordering));
  }
  if (top_k == 0) {   // ... and the whole output is a permutation of the input
    auto key = [](const Row &r) { return std::make_tuple(r.a, r.b, r.c); };
    std::vector<std::tuple<std::int32_t, double, std::int64_t>> x, y;
    for (const Row &r : out) x.push_back(key(r));
    for (const Row &r : rows) y.push_back(key(r));
    std::sort(x.begin(), x.end());
    std::sort(y.begin(), y.end());
    EXPECT_TRUE(x == y);
  }
}
}  // namespace

int main() {
  if (qsx_device_count() < 1) {
    std::fprintf(stderr, "sort_operator_test needs an MI355X: %s\n", qsx_status_string(QSX_ERR_NO_DEVICE));
    return 2;
  }
  const std::vector<Row> rows = makeRows();
  for (const bool use_foreman : {false, true}) {
    for (const std::size_t top_k : {static_cast<std::size_t>(0), static_cast<std::size_t>(10), static_cast<std::size_t>(12345)}) {
      runCase(rows, {0}, {true}, top_k, use_foreman);                          // 1Column_NonNull_Asc[_TopK]
      runCase(rows, {1}, {false}, top_k, use_foreman);                         // 1Column_NonNull_Desc[_TopK]
      runCase(rows, {2, 0, 1}, {true, true, true}, top_k, use_foreman);        // 3Column_NonNull_Asc[_TopK]
      runCase(rows, {2, 1, 0}, {false, false, false}, top_k, use_foreman);     // 3Column_NonNull_Desc[_TopK]
      runCase(rows, {0, 2, 1}, {true, false, true}, top_k, use_foreman);       // mixed ordering (:754-800 without NULLs)
      runCase(rows, {1}, {false}, top_k, use_foreman, /*limit_runs=*/true);    // ORDER BY b DESC LIMIT k with the limit in the runs
      runCase(rows, {2, 0, 1}, {true, false, true}, top_k, use_foreman, true);
    }
  }
  // ---- predicates on the sort column of a sorted block: binary search (qsx_select_cmp_sorted) == scan ----------------
  {
    StorageManager storage;
    CatalogRelation sorted(130, "sorted"), plain(131, "plain"), out_a(132, "out_a"), out_b(133, "out_b");
    for (CatalogRelation *r : {&sorted, &plain, &out_a, &out_b}) {
      r->addAttribute("a", Type::Int());
      r->addAttribute("b", Type::Double());
    }
    std::vector<Row> by_b = rows;
    std::sort(by_b.begin(), by_b.end(), [](const Row &x, const Row &y) { return x.b < y.b; });
    std::vector<std::int32_t> a;
    std::vector<double> b;
    for (const Row &r : by_b) { a.push_back(r.a); b.push_back(r.b); }
    const block_id sorted_id = storage.loadBlock(&sorted, {a.data(), b.data()}, static_cast<std::int64_t>(a.size()));
    storage.loadBlock(&plain, {a.data(), b.data()}, static_cast<std::int64_t>(a.size()));
    storage.getBlock(sorted_id)->setSortColumn(1);
    for (const ComparisonID cmp : {ComparisonID::kEqual, ComparisonID::kNotEqual, ComparisonID::kLess, ComparisonID::kLessOrEqual,
                                   ComparisonID::kGreater, ComparisonID::kGreaterOrEqual}) {
      for (const double lit : {-60.0, 12.5, 12.55, 64.875, 1000.0}) {
        QueryContext ctx;
        Predicate pred;
        pred.conjuncts.push_back({1, cmp, TypedLiteral::Double(lit)});
        pred.conjuncts.push_back({0, ComparisonID::kLess, TypedLiteral::Int(50)});   // a second term: the range is a filter for it
        const auto pred_id = ctx.addPredicate(pred);
        const auto dest_a = ctx.addInsertDestination(&out_a, &storage);
        const auto dest_b = ctx.addInsertDestination(&out_b, &storage);
        SelectOperator on_sorted(0, sorted, false, out_a, dest_a, pred_id, std::vector<attribute_id>{0, 1}, true);
        SelectOperator on_plain(0, plain, false, out_b, dest_b, pred_id, std::vector<attribute_id>{0, 1}, true);
        fetchAndExecuteWorkOrders(&on_sorted, &ctx, &storage);
        fetchAndExecuteWorkOrders(&on_plain, &ctx, &storage);
        auto collect = [&](QueryContext::insert_destination_id d) {
          std::vector<std::pair<std::int32_t, double>> v;
          for (block_id id : ctx.getInsertDestination(d)->getTouchedBlocks()) {
            BlockReference blk = storage.getBlock(id);
            const std::size_t k = static_cast<std::size_t>(blk->numTuples());
            std::vector<std::int32_t> ca(k);
            std::vector<double> cb(k);
            blk->copyAttributeToHost(0, ca.data());
            blk->copyAttributeToHost(1, cb.data());
            for (std::size_t i = 0; i < k; ++i) v.emplace_back(ca[i], cb[i]);
          }
          return v;
        };
        const auto got = collect(dest_a), want = collect(dest_b);
        EXPECT_TRUE(got == want);
        std::size_t expected = 0;
        for (const Row &r : by_b) {
          bool m = false;
          switch (cmp) {
            case ComparisonID::kEqual: m = r.b == lit; break;
            case ComparisonID::kNotEqual: m = r.b != lit; break;
            case ComparisonID::kLess: m = r.b < lit; break;
            case ComparisonID::kLessOrEqual: m = r.b <= lit; break;
            case ComparisonID::kGreater: m = r.b > lit; break;
            default: m = r.b >= lit; break;
          }
          if (m && r.a < 50) ++expected;
        }
        EXPECT_EQ(got.size(), expected);
      }
    }
  }
  return finish("sort_operator_test");
}
