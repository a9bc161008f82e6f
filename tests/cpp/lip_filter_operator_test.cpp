// LIP filters through the operator layer (utility/lip_filter/, QueryContext.hpp:338-395, deployLIPFilters
// RelationalOperator.hpp:294-297).  Data of query_optimizer/tests/execution_generator/LIP.test:20-29: R(x, y) holds the
// even numbers 0..100000, S(z) the multiples of 3; the semi join R.x = S.z leaves the multiples of 6, among them
// exactly {0, 30000, 60000, 90000} with x % 10000 = 0 (:39-75).  BuildHash builds an exact bit-vector filter on S.z;
// Select, Aggregation and HashJoin work orders probe it on R.x.  GPU work orders.
#include <algorithm>
#include <numeric>

#include "test_util.hpp"

using namespace quickstep;

namespace {
constexpr std::int32_t kLimit = 100000;
constexpr std::int64_t kBlockRows = 8192;

struct Fixture {
  CatalogRelation r{1, "r"}, s{2, "s"};
  StorageManager storage;
  Fixture() {
    r.addAttribute("x", Type::Int());
    r.addAttribute("y", Type::Int());
    s.addAttribute("z", Type::Int());
    load(&r, 2, 2);
    load(&s, 3, 1);
  }
  void load(CatalogRelation *rel, int step, int ncols) {
    std::vector<std::int32_t> v;
    for (std::int32_t i = 0; i <= kLimit; i += step) v.push_back(i);
    for (std::size_t at = 0; at < v.size(); at += kBlockRows) {
      const std::int64_t n = static_cast<std::int64_t>(std::min<std::size_t>(kBlockRows, v.size() - at));
      std::vector<const void *> cols(ncols, v.data() + at);
      storage.loadBlock(rel, cols, n);
    }
  }
};

std::vector<std::int32_t> readInts(QueryContext &ctx, QueryContext::insert_destination_id dest, StorageManager &storage) {
  std::vector<std::int32_t> out;
  for (block_id b : ctx.getInsertDestination(dest)->getTouchedBlocks()) {
    BlockReference blk = storage.getBlock(b);
    const std::size_t at = out.size();
    out.resize(at + static_cast<std::size_t>(blk->numTuples()));
    blk->copyAttributeToHost(0, out.data() + at);
  }
  std::sort(out.begin(), out.end());
  return out;
}

void checkMultiplesOfSix(const std::vector<std::int32_t> &x) {
  EXPECT_EQ(x.size(), static_cast<std::size_t>(kLimit / 6 + 1));
  std::vector<std::int32_t> round;
  for (std::size_t i = 0; i < x.size(); ++i) {
    EXPECT_EQ(x[i], static_cast<std::int32_t>(6 * i));
    if (x[i] % 10000 == 0) round.push_back(x[i]);
  }
  EXPECT_EQ(round.size(), static_cast<std::size_t>(4));   // LIP.test:70-75: 0, 30000, 60000, 90000
  if (round.size() == 4) { EXPECT_EQ(round[1], 30000); EXPECT_EQ(round[2], 60000); EXPECT_EQ(round[3], 90000); }
}
}  // namespace

int main() {
  if (qsx_device_count() < 1) {
    std::fprintf(stderr, "lip_filter_operator_test needs an MI355X: %s\n", qsx_status_string(QSX_ERR_NO_DEVICE));
    return 2;
  }
  for (const int variant : {0, 1, 2, 3}) {
    const bool use_foreman = (variant & 1) != 0;
    const std::size_t blocks_per_work_order = (variant & 2) != 0 ? 4 : 1;   // 4: work orders over runs of blocks
    for (const qsx_lip_kind_t kind : {QSX_LIP_BITVECTOR_EXACT, QSX_LIP_SINGLE_IDENTITY_HASH}) {
      Fixture f;
      CatalogRelation selected(3, "selected"), semi(4, "semi"), sums(5, "sums");
      selected.addAttribute("x", Type::Int());
      semi.addAttribute("x", Type::Int());
      sums.addAttribute("sum_x", Type::Long());
      sums.addAttribute("count", Type::Long());
      QueryContext ctx;
      // exact: one bit per value of [min, max] of S.z; identity hash: 8 bits per build row, here a multiple of the
      // key range so that it is exact too (SingleIdentityHashFilter.hpp:156-169: bit = value % cardinality)
      const auto filter = kind == QSX_LIP_BITVECTOR_EXACT ? ctx.addLIPFilter(kind, kLimit + 1, 0)
                                                          : ctx.addLIPFilter(kind, 2 * (kLimit + 1));
      QueryContext::LIPFilterDeployment build_dep, probe_dep;
      build_dep.build_entries.push_back({filter, 0});   // S.z
      probe_dep.probe_entries.push_back({filter, 0});   // R.x
      const auto build_dep_id = ctx.addLIPDeployment(build_dep);
      const auto probe_dep_id = ctx.addLIPDeployment(probe_dep);
      const auto table = ctx.addJoinHashTable(kInt, kLimit / 3 + 1);
      const auto sel_dest = ctx.addInsertDestination(&selected, &f.storage);
      const auto semi_dest = ctx.addInsertDestination(&semi, &f.storage);
      const auto sum_dest = ctx.addInsertDestination(&sums, &f.storage);
      AggregationStateSpec spec;
      spec.input_relation = &f.r;
      spec.aggregates = {{AggregationID::kSum, 0}, {AggregationID::kCount, kInvalidAttributeID}};
      spec.strategy = QSX_AGG_SINGLE_STATE;
      const auto state = ctx.addAggregationState(spec);
      const auto semi_selection = ctx.addScalarGroup({0});
      const std::vector<bool> on_build = {false};

      auto *builder = new BuildHashOperator(0, f.s, true, {0}, false, 1, table);
      builder->deployLIPFilters(build_dep_id);
      auto *select = new SelectOperator(0, f.r, false, selected, sel_dest, QueryContext::kInvalidPredicateId,
                                        std::vector<attribute_id>{0}, true);
      select->deployLIPFilters(probe_dep_id);
      auto *aggregate = new AggregationOperator(0, f.r, true, state);
      aggregate->deployLIPFilters(probe_dep_id);
      auto *finalize = new FinalizeAggregationOperator(0, state, 1, false, 1, sums, sum_dest);
      auto *prober = new HashJoinOperator(0, f.s, f.r, true, {0}, false, 1, false, semi, semi_dest, table,
                                          QueryContext::kInvalidPredicateId, semi_selection, &on_build,
                                          HashJoinOperator::JoinType::kLeftSemiJoin);
      prober->deployLIPFilters(probe_dep_id);
      builder->setBlocksPerWorkOrder(blocks_per_work_order);
      select->setBlocksPerWorkOrder(blocks_per_work_order);
      aggregate->setBlocksPerWorkOrder(blocks_per_work_order);
      prober->setBlocksPerWorkOrder(blocks_per_work_order);
      std::vector<std::unique_ptr<RelationalOperator>> owned;
      if (use_foreman) {
        QueryPlan plan;
        const auto b = plan.addRelationalOperator(builder);
        const auto s = plan.addRelationalOperator(select);
        const auto a = plan.addRelationalOperator(aggregate);
        const auto z = plan.addRelationalOperator(finalize);
        const auto p = plan.addRelationalOperator(prober);
        plan.addDirectDependency(s, b, true);   // the filter must be complete before anything probes it
        plan.addDirectDependency(a, b, true);
        plan.addDirectDependency(p, b, true);
        plan.addDirectDependency(z, a, true);
        ForemanSingleNode foreman(&plan, &ctx, &f.storage, 4);
        foreman.run();
      } else {
        for (RelationalOperator *op : {static_cast<RelationalOperator *>(builder), static_cast<RelationalOperator *>(select),
                                       static_cast<RelationalOperator *>(aggregate), static_cast<RelationalOperator *>(finalize),
                                       static_cast<RelationalOperator *>(prober)}) {
          owned.emplace_back(op);
          fetchAndExecuteWorkOrders(op, &ctx, &f.storage);
        }
      }
      checkMultiplesOfSix(readInts(ctx, sel_dest, f.storage));
      checkMultiplesOfSix(readInts(ctx, semi_dest, f.storage));
      // SUM(x), COUNT(*) over the filtered tuples
      const std::int64_t k = kLimit / 6;
      std::int64_t sum = 0, count = 0;
      for (block_id b : ctx.getInsertDestination(sum_dest)->getTouchedBlocks()) {
        BlockReference blk = f.storage.getBlock(b);
        EXPECT_EQ(blk->numTuples(), static_cast<std::int64_t>(1));
        blk->copyAttributeToHost(0, &sum);
        blk->copyAttributeToHost(1, &count);
      }
      EXPECT_EQ(sum, 6 * k * (k + 1) / 2);
      EXPECT_EQ(count, k + 1);
    }
  }
  return finish("lip_filter_operator_test");
}
