// query_optimizer/tests/execution_generator/Partition.test through the operator layer: the test table of
// TestDatabaseLoader.cpp:118-170 (x = 0..24: int_col = (-1)^x x, NULL when x % 10 == 0; double_col = (-1)^x x sqrt(x);
// char_col = "<int_col> <sqrt(x)>"), dim_4_hash_partitions / dim_2_hash_partitions / fact as Partition.test:18-43 fills them,
// and the known answers of
//   :45-73    the membership listing of a relation PARTITION BY HASH(id) PARTITIONS 4 — here produced by a SelectOperator
//             with has_repartition = true into a PartitionAwareInsertDestination (storage/InsertDestination.hpp:490-660);
//   :75-92    the partitioned hash join (both sides 4-way partitioned on the key);
//   :94-101   the join with only the probe side partitioned (broadcast build, BuildHashOperator.hpp:99,146-152);
//   :112-133  the REPARTITIONED hash join: dim_2_hash_partitions is repartitioned 4 ways on its way into the build
//             (Select --streaming, partition ids carried by the data-pipeline edge--> BuildHash), then joined per partition;
//   :135-162  partitioned aggregation (COUNT(*) = 22; GROUP BY id over the rows with id > 0).
// Also: has_repartition with a plain destination (and the reverse) is an error, never ignored; NULL bits and a CHAR(20)
// attribute follow their tuples through the scatter.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>

#include "test_util.hpp"

using namespace quickstep;

namespace {
struct TestRow {
  std::int32_t int_col;
  bool int_null;
  double double_col;
  char char_col[20];
};
std::vector<TestRow> testTable() {
  std::vector<TestRow> rows;
  for (int x = 0; x < 25; ++x) {
    TestRow r;
    r.int_col = (x % 2 == 0 ? 1 : -1) * x;
    r.int_null = x % 10 == 0;
    r.double_col = (x % 2 == 0 ? 1 : -1) * x * std::sqrt(static_cast<double>(x));
    std::memset(r.char_col, 0, sizeof(r.char_col));
    std::snprintf(r.char_col, sizeof(r.char_col), "%d %f", r.int_col, std::sqrt(static_cast<double>(x)));
    rows.push_back(r);
  }
  return rows;
}
std::size_t pid(std::int32_t id, std::size_t parts) {   // HashPartitionSchemeHeader::getPartitionId, P a power of two
  return static_cast<std::size_t>(static_cast<std::uint32_t>(id)) & (parts - 1);
}

// (id INT NULL, char_col CHAR(20)) hash-partitioned `parts` ways on id; rows with int_col > 0 or int_col < 0
void loadDim(StorageManager *storage, CatalogRelation *rel, std::size_t parts) {
  rel->addAttribute("id", Type::Int().getNullableVersion());
  rel->addAttribute("char_col", Type::Char(20));
  if (parts > 1) rel->setPartitionScheme(parts, 0);
  for (std::size_t p = 0; p < parts; ++p) {
    std::vector<std::int32_t> id;
    std::vector<char> text;
    for (const TestRow &r : testTable()) {
      if (r.int_null || r.int_col == 0 || (parts > 1 && pid(r.int_col, parts) != p)) continue;
      id.push_back(r.int_col);
      text.insert(text.end(), r.char_col, r.char_col + 20);
    }
    storage->loadBlock(rel, {id.data(), text.data()}, static_cast<std::int64_t>(id.size()), p);
  }
}
// fact (id INT NULL, score DOUBLE NULL) PARTITION BY HASH(id) PARTITIONS 4: the rows with int_col % 2 = 0 (NULLs drop out)
void loadFact(StorageManager *storage, CatalogRelation *rel) {
  rel->addAttribute("id", Type::Int().getNullableVersion());
  rel->addAttribute("score", Type::Double().getNullableVersion());
  rel->setPartitionScheme(4, 0);
  for (std::size_t p = 0; p < 4; ++p) {
    std::vector<std::int32_t> id;
    std::vector<double> score;
    for (const TestRow &r : testTable()) {
      if (r.int_null || r.int_col % 2 != 0 || pid(r.int_col, 4) != p) continue;
      id.push_back(r.int_col);
      score.push_back(r.double_col);
    }
    storage->loadBlock(rel, {id.data(), score.data()}, static_cast<std::int64_t>(id.size()), p);
  }
}

// the answer of every join of Partition.test:75-133: (fact.id, char_col of the matching dim row)
std::map<std::int32_t, std::string> expectedJoin() {
  std::map<std::int32_t, std::string> want;
  for (const TestRow &r : testTable()) {
    if (!r.int_null && r.int_col != 0 && r.int_col % 2 == 0) want[r.int_col] = std::string(r.char_col, strnlen(r.char_col, 20));
  }
  return want;
}

std::map<std::int32_t, std::string> collectJoin(QueryContext &ctx, QueryContext::insert_destination_id dest, StorageManager &storage,
                                                std::size_t *rows) {
  std::map<std::int32_t, std::string> got;
  *rows = 0;
  for (block_id b : ctx.getInsertDestination(dest)->getTouchedBlocks()) {
    BlockReference blk = storage.getBlock(b);
    const std::size_t k = static_cast<std::size_t>(blk->numTuples());
    if (k == 0) continue;
    std::vector<std::int32_t> id(k);
    std::vector<char> text(k * 20);
    blk->copyAttributeToHost(0, id.data());
    blk->copyAttributeToHost(1, text.data());
    for (std::size_t i = 0; i < k; ++i, ++*rows) got[id[i]] = std::string(&text[i * 20], strnlen(&text[i * 20], 20));
  }
  return got;
}

// build_parts: partitions of the stored dim relation (4: partitioned join; 1: broadcast build); repartition: dim_2 goes
// through a repartitioning Select first
void runJoin(std::size_t dim_parts, bool repartition, bool use_foreman) {
  StorageManager storage;
  CatalogRelation dim(1, "dim"), fact(2, "fact"), dim4(3, "dim_repartitioned"), out(4, "out");
  loadDim(&storage, &dim, dim_parts);
  loadFact(&storage, &fact);
  out.addAttribute("id", Type::Int().getNullableVersion());
  out.addAttribute("char_col", Type::Char(20));
  QueryContext ctx;
  const auto table = ctx.addJoinHashTable(kInt, 32, 4);
  const auto d_out = ctx.addInsertDestination(&out, &storage);
  const auto selection = ctx.addScalarGroup({0, 1});         // fact.id (probe), dim.char_col (build)
  const std::vector<bool> on_build{false, true};
  QueryPlan plan;
  const CatalogRelation *build_rel = &dim;
  std::size_t i_select = 0;
  if (repartition) {
    dim4.addAttribute("id", Type::Int().getNullableVersion());
    dim4.addAttribute("char_col", Type::Char(20));
    dim4.setPartitionScheme(4, 0);
    const auto d_dim4 = ctx.addPartitionAwareInsertDestination(&dim4, &storage);
    i_select = plan.addRelationalOperator(new SelectOperator(0, dim, /*has_repartition=*/true, dim4, d_dim4, QueryContext::kInvalidPredicateId,
                                                             std::vector<attribute_id>{0, 1}, true));
    build_rel = &dim4;
  }
  const auto i_build = plan.addRelationalOperator(new BuildHashOperator(0, *build_rel, !repartition, {0}, true, 4, table));
  const auto i_join = plan.addRelationalOperator(new HashJoinOperator(0, *build_rel, fact, true, {0}, true, 4, false, out, d_out, table,
                                                                      QueryContext::kInvalidPredicateId, selection, &on_build,
                                                                      HashJoinOperator::JoinType::kInnerJoin));
  if (repartition) plan.addDirectDependency(i_build, i_select, false);
  plan.addDirectDependency(i_join, i_build, true);
  if (use_foreman || repartition) {
    ForemanSingleNode foreman(&plan, &ctx, &storage, 3);
    foreman.run();
  } else {
    for (std::size_t i = 0; i < plan.size(); ++i) fetchAndExecuteWorkOrders(plan.getOperator(i), &ctx, &storage);
  }
  std::size_t rows = 0;
  const auto got = collectJoin(ctx, d_out, storage, &rows);
  const auto want = expectedJoin();
  EXPECT_EQ(rows, static_cast<std::size_t>(10));
  EXPECT_EQ(got.size(), want.size());
  EXPECT_TRUE(got == want);
  // the output of a join over partitioned inputs WITHOUT repartition keeps the probe partition of every block
  for (const InsertDestination::TouchedBlock &t : ctx.getInsertDestination(d_out)->getTouchedBlocksWithPartitions()) {
    BlockReference blk = storage.getBlock(t.id);
    std::vector<std::int32_t> id(static_cast<std::size_t>(blk->numTuples()));
    if (id.empty()) continue;
    blk->copyAttributeToHost(0, id.data());
    for (std::int32_t v : id) EXPECT_EQ(pid(v, 4), t.partition);
  }
}

// Partition.test:45-73: SELECT * of the 4-way partitioned relation lists partition after partition
void runMembership() {
  StorageManager storage;
  CatalogRelation src(1, "dim_unpartitioned"), dst(2, "dim_4_hash_partitions");
  loadDim(&storage, &src, 1);
  dst.addAttribute("id", Type::Int().getNullableVersion());
  dst.addAttribute("char_col", Type::Char(20));
  dst.setPartitionScheme(4, 0);
  QueryContext ctx;
  const auto dest = ctx.addPartitionAwareInsertDestination(&dst, &storage);
  SelectOperator op(0, src, /*has_repartition=*/true, dst, dest, QueryContext::kInvalidPredicateId, std::vector<attribute_id>{0, 1}, true);
  fetchAndExecuteWorkOrders(&op, &ctx, &storage);
  const std::vector<std::vector<std::int32_t>> listing = {{4, 8, 12, 16, 24}, {-3, -7, -11, -15, -19, -23}, {2, 6, 14, 18, 22}, {-1, -5, -9, -13, -17, -21}};
  std::size_t total = 0;
  for (std::size_t p = 0; p < 4; ++p) {
    std::vector<std::int32_t> got;
    for (block_id b : dst.getBlocksInPartition(p)) {
      BlockReference blk = storage.getBlock(b);
      const std::size_t k = static_cast<std::size_t>(blk->numTuples());
      std::vector<std::int32_t> id(k);
      std::vector<char> text(k * 20);
      blk->copyAttributeToHost(0, id.data());
      blk->copyAttributeToHost(1, text.data());
      for (std::size_t i = 0; i < k; ++i) {
        got.push_back(id[i]);
        EXPECT_EQ(std::atoi(&text[i * 20]), id[i]);          // the CHAR(20) attribute went with its tuple ("<id> <sqrt>")
      }
    }
    total += got.size();
    EXPECT_TRUE(got == listing[p]);                          // the scatter is stable: input order within a partition
  }
  EXPECT_EQ(total, static_cast<std::size_t>(22));
}

// The repartitioning Select over runs of blocks: with attributes K9 moves by itself (1 / 2 / 4 / 8 bytes, not nullable) the
// scatter reads the stored blocks where they lie (InsertDestination::insertRunRepartitioned, qsx_partition_scatter_blocks) — the
// partitions hold what the block-at-a-time work orders leave: every tuple once, in input order, projected and reordered
// (the partition attribute is output attribute 1, input attribute 2).  A relation with an attribute the scatter cannot move
// (CHAR(20)) or NULLs keeps the copy-then-scatter path: runMembership / runNullsThroughRepartition with runs below.
void runRepartitionOverRuns(std::size_t parts, std::size_t blocks_per_work_order) {
  StorageManager storage;
  CatalogRelation src(1, "lineitem_like"), dst(2, "by_orderkey");
  src.addAttribute("payload", Type::Long());
  src.addAttribute("unused", Type::Double());
  src.addAttribute("orderkey", Type::Int());
  dst.addAttribute("payload", Type::Long());
  dst.addAttribute("orderkey", Type::Int());
  dst.setPartitionScheme(parts, 1);
  const std::vector<std::int64_t> block_rows = {30000, 30000, 30000, 0, 30000, 30000, 30000, 30000, 1234, 1};
  std::vector<std::vector<std::pair<std::int64_t, std::int32_t>>> want(parts);
  std::uint64_t state = 88172645463325252ull;
  std::int64_t serial = 0;
  for (const std::int64_t rows : block_rows) {
    std::vector<std::int64_t> payload(rows);
    std::vector<double> unused(rows, 1.5);
    std::vector<std::int32_t> key(rows);
    for (std::int64_t i = 0; i < rows; ++i) {
      state ^= state << 13; state ^= state >> 7; state ^= state << 17;
      key[i] = static_cast<std::int32_t>(state % 1000003) - 500;           // (negative keys: the zero-extended bit pattern)
      payload[i] = serial++;
      const std::uint64_t h = static_cast<std::uint32_t>(key[i]);
      const std::size_t p = (parts & (parts - 1)) == 0 ? h & (parts - 1) : (h >= parts ? h % parts : h);
      want[p].emplace_back(payload[i], key[i]);
    }
    storage.loadBlock(&src, {payload.data(), unused.data(), key.data()}, rows, 0);
  }
  QueryContext ctx;
  const auto dest = ctx.addPartitionAwareInsertDestination(&dst, &storage);
  SelectOperator op(0, src, /*has_repartition=*/true, dst, dest, QueryContext::kInvalidPredicateId, std::vector<attribute_id>{0, 2}, true);
  op.setBlocksPerWorkOrder(blocks_per_work_order);
  fetchAndExecuteWorkOrders(&op, &ctx, &storage);
  for (std::size_t p = 0; p < parts; ++p) {
    std::vector<std::pair<std::int64_t, std::int32_t>> got;
    for (block_id b : dst.getBlocksInPartition(p)) {
      BlockReference blk = storage.getBlock(b);
      const std::size_t k = static_cast<std::size_t>(blk->numTuples());
      std::vector<std::int64_t> payload(k);
      std::vector<std::int32_t> key(k);
      blk->copyAttributeToHost(0, payload.data());
      blk->copyAttributeToHost(1, key.data());
      for (std::size_t i = 0; i < k; ++i) got.emplace_back(payload[i], key[i]);
    }
    EXPECT_TRUE(got == want[p]);
  }
}

// NULL bits follow their tuples through the scatter; an empty work-order output leaves no block behind
void runNullsThroughRepartition() {
  StorageManager storage;
  CatalogRelation src(1, "src"), dst(2, "dst");
  for (CatalogRelation *r : {&src, &dst}) {
    r->addAttribute("id", Type::Long());
    r->addAttribute("v", Type::Double().getNullableVersion());
  }
  dst.setPartitionScheme(8, 0);
  const std::int64_t n = 10007;
  std::vector<std::int64_t> id(n);
  std::vector<double> v(n);
  std::vector<std::uint64_t> nulls(static_cast<std::size_t>((n + 63) / 64) + 1, 0);
  for (std::int64_t i = 0; i < n; ++i) {
    id[i] = (i * 2654435761ll) % 100003 - 50000;
    v[i] = static_cast<double>(i);
    if (i % 7 == 3) nulls[i >> 6] |= 1ull << (63 - (i & 63));
  }
  const std::vector<const std::uint64_t *> null_bitmaps = {nullptr, nulls.data()};
  storage.loadBlock(&src, {id.data(), v.data()}, n, 0, nullptr, &null_bitmaps);
  storage.loadBlock(&src, {id.data(), v.data()}, 0);          // an empty block
  QueryContext ctx;
  const auto dest = ctx.addPartitionAwareInsertDestination(&dst, &storage);
  SelectOperator op(0, src, true, dst, dest, QueryContext::kInvalidPredicateId, std::vector<attribute_id>{0, 1}, true);
  fetchAndExecuteWorkOrders(&op, &ctx, &storage);
  std::int64_t total = 0;
  for (std::size_t p = 0; p < 8; ++p) {
    for (block_id b : dst.getBlocksInPartition(p)) {
      BlockReference blk = storage.getBlock(b);
      const std::size_t k = static_cast<std::size_t>(blk->numTuples());
      EXPECT_TRUE(k > 0);
      std::vector<std::int64_t> got_id(k);
      std::vector<double> got_v(k);
      std::vector<std::uint64_t> got_nulls((k + 63) / 64);
      blk->copyAttributeToHost(0, got_id.data());
      blk->copyAttributeToHost(1, got_v.data());
      blk->copyNullBitmapToHost(1, got_nulls.data());
      for (std::size_t i = 0; i < k; ++i, ++total) {
        const std::int64_t row = static_cast<std::int64_t>(got_v[i]);       // v = source row number
        EXPECT_EQ(static_cast<std::size_t>(static_cast<std::uint64_t>(got_id[i]) & 7u), p);
        EXPECT_EQ(got_id[i], id[row]);
        EXPECT_EQ(((got_nulls[i >> 6] >> (63 - (i & 63))) & 1u) != 0, row % 7 == 3);
      }
    }
  }
  EXPECT_EQ(total, n);
}

// Partition.test:135-162: COUNT(*) over the partitioned relation (one state per partition, every partition finalized);
// GROUP BY id WHERE id > 0 — the partition attribute is the group-by key, so the partitions' results are disjoint
void runPartitionedAggregation() {
  StorageManager storage;
  CatalogRelation dim(1, "dim_4_hash_partitions"), out_count(2, "count"), out_groups(3, "groups");
  loadDim(&storage, &dim, 4);
  out_count.addAttribute("count", Type::Long());
  out_groups.addAttribute("id", Type::Int().getNullableVersion());
  out_groups.addAttribute("count", Type::Long());
  QueryContext ctx;
  AggregationStateSpec count_spec;
  count_spec.input_relation = &dim;
  count_spec.aggregates = {AggregateSpec(AggregationID::kCount, kInvalidAttributeID)};
  count_spec.strategy = QSX_AGG_GENERIC;
  const auto count_state = ctx.addAggregationState(count_spec, 4);
  Predicate positive;
  positive.conjuncts.push_back(ComparisonPredicate(0, ComparisonID::kGreater, TypedLiteral::Int(0)));
  AggregationStateSpec group_spec;
  group_spec.input_relation = &dim;
  group_spec.group_by = {0};
  group_spec.aggregates = {AggregateSpec(AggregationID::kCount, kInvalidAttributeID)};
  group_spec.predicate = &positive;
  group_spec.strategy = QSX_AGG_GENERIC;
  const auto group_state = ctx.addAggregationState(group_spec, 4);
  const auto d_count = ctx.addInsertDestination(&out_count, &storage), d_groups = ctx.addInsertDestination(&out_groups, &storage);
  AggregationOperator agg_count(0, dim, true, count_state, 4), agg_groups(0, dim, true, group_state, 4);
  FinalizeAggregationOperator fin_count(0, count_state, 4, false, 1, out_count, d_count), fin_groups(0, group_state, 4, false, 1, out_groups, d_groups);
  for (RelationalOperator *op : std::initializer_list<RelationalOperator *>{&agg_count, &agg_groups, &fin_count, &fin_groups}) {
    fetchAndExecuteWorkOrders(op, &ctx, &storage);
  }
  std::int64_t total = 0;
  for (block_id b : ctx.getInsertDestination(d_count)->getTouchedBlocks()) {
    BlockReference blk = storage.getBlock(b);
    std::vector<std::int64_t> c(static_cast<std::size_t>(blk->numTuples()));
    if (c.empty()) continue;
    blk->copyAttributeToHost(0, c.data());
    for (std::int64_t x : c) total += x;       // (the reference adds the partitions' counts in a final aggregation)
  }
  EXPECT_EQ(total, 22);
  std::map<std::int32_t, std::int64_t> groups;
  for (block_id b : ctx.getInsertDestination(d_groups)->getTouchedBlocks()) {
    BlockReference blk = storage.getBlock(b);
    const std::size_t k = static_cast<std::size_t>(blk->numTuples());
    if (k == 0) continue;
    std::vector<std::int32_t> id(k);
    std::vector<std::int64_t> c(k);
    blk->copyAttributeToHost(0, id.data());
    blk->copyAttributeToHost(1, c.data());
    for (std::size_t i = 0; i < k; ++i) groups[id[i]] += c[i];
  }
  const std::set<std::int32_t> want = {4, 8, 12, 16, 24, 2, 6, 14, 18, 22};
  EXPECT_EQ(groups.size(), want.size());
  for (const auto &kv : groups) {
    EXPECT_TRUE(want.count(kv.first) == 1);
    EXPECT_EQ(kv.second, 1);
  }
}

void runMismatchedRepartitionIsAnError() {
  StorageManager storage;
  CatalogRelation src(1, "src"), plain(2, "plain"), parted(3, "parted");
  loadDim(&storage, &src, 1);
  for (CatalogRelation *r : {&plain, &parted}) {
    r->addAttribute("id", Type::Int().getNullableVersion());
    r->addAttribute("char_col", Type::Char(20));
  }
  parted.setPartitionScheme(4, 0);
  QueryContext ctx;
  const auto d_plain = ctx.addInsertDestination(&plain, &storage);
  const auto d_parted = ctx.addPartitionAwareInsertDestination(&parted, &storage);
  for (const bool has_repartition : {true, false}) {
    SelectOperator op(0, src, has_repartition, has_repartition ? plain : parted, has_repartition ? d_plain : d_parted,
                      QueryContext::kInvalidPredicateId, std::vector<attribute_id>{0, 1}, true);
    bool threw = false;
    try {
      fetchAndExecuteWorkOrders(&op, &ctx, &storage);
    } catch (const ExecutionError &e) {
      threw = e.status() == QSX_ERR_INVALID_ARGUMENT;
    }
    EXPECT_TRUE(threw);
  }
  EXPECT_EQ(plain.getBlocksSnapshot().size(), static_cast<std::size_t>(0));
  bool threw = false;
  try {
    ctx.addPartitionAwareInsertDestination(&plain, &storage);     // no partition scheme to be aware of
  } catch (const ExecutionError &) {
    threw = true;
  }
  EXPECT_TRUE(threw);
}
}  // namespace

int main() {
  if (qsx_device_count() < 1) {
    std::fprintf(stderr, "partition_operator_test needs an MI355X: %s\n", qsx_status_string(QSX_ERR_NO_DEVICE));
    return 2;
  }
  runMembership();
  runNullsThroughRepartition();
  for (const std::size_t parts : {std::size_t(1), std::size_t(4), std::size_t(5), std::size_t(8), std::size_t(64)}) {
    for (const std::size_t blocks : {std::size_t(1), std::size_t(3), std::size_t(100)}) runRepartitionOverRuns(parts, blocks);
  }
  for (const bool use_foreman : {false, true}) {
    runJoin(4, false, use_foreman);    // partitioned hash join
    runJoin(1, false, use_foreman);    // broadcast build
  }
  runJoin(2, true, true);              // repartitioned hash join
  runPartitionedAggregation();
  runMismatchedRepartitionIsAnError();
  return finish("partition_operator_test");
}
