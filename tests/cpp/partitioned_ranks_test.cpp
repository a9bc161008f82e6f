// partitioned_ranks_test — sharding FROM THE C++ OPERATOR LAYER: every rank is a process of its own with its own
// StorageManager, QueryContext and ForemanSingleNode, all ranks run the same plan, rank r owns the partitions p with
// p % world == r, and tuples of foreign partitions leave through PartitionExchangeOperator (qsx_exchange_counts +
// qsx_alltoallv), partial aggregation states meet in ExchangeAggregationStatesOperator (qsx_agg_allgather_merge /
// qsx_agg_reduce_scatter).  The ranks share the box's one GPU; libqsx.so is bound to the tests' loopback transport
// (QSX_RCCL_LIBRARY = libloopback_rccl.so) — the pattern of the reference's distributed test runner, several Shiftboss
// "nodes" with their own StorageManagers replaying the same .test files
// (query_optimizer/tests/DistributedExecutionGeneratorTestRunner.cpp:72-150).
//
// The parent (no arguments) starts world = 2 and world = 3 rank processes BEFORE it touches the GPU, then runs every
// scenario in ONE process without any exchange and compares: the union of the ranks' result rows must be exactly the
// single-process operators' rows.  Scenarios:
//   partitioned_join     Partition.test:75-92     both sides 4-way partitioned on the key, stored per owner
//   broadcast_join       Partition.test:94-101    build side unpartitioned (dealt over the ranks) -> broadcast exchange
//   repartitioned_join   Partition.test:112-133   dim_2_hash_partitions repartitioned 4 ways -> exchange -> build
//   c4                   BASELINE config 4 shape  orders ⋈ lineitem, both sides repartitioned on orderkey and shuffled,
//                                                 output (key, o_payload, l_payload)
//   partitioned_agg      Partition.test:135-162   COUNT(*), GROUP BY the partition key
//   merged_agg           group-by keys that are NOT the partition key: hash state (all-gather + merge) and
//                        CollisionFreeVector state (reduce-scatter), every rank finalizes its slice
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <thread>

#include "test_util.hpp"

extern char **environ;

using namespace quickstep;

namespace {

// ---- who am I --------------------------------------------------------------------------------------------------------
struct Ranks {
  int world = 1, rank = 0;
  RankGroup *group = nullptr;      // nullptr: the single-process reference run
  bool owns(std::size_t part) const { return part % static_cast<std::size_t>(world) == static_cast<std::size_t>(rank); }
};
typedef std::vector<std::string> Lines;

// ---- Partition.test's tables -------------------------------------------------------------------------------------------
struct TestRow {
  std::int32_t int_col;
  bool int_null;
  double double_col;
  char char_col[20];
};
std::vector<TestRow> testTable() {   // TestDatabaseLoader.cpp:118-170
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
std::size_t pid(std::int64_t id, std::size_t parts) {   // HashPartitionSchemeHeader::getPartitionId on the zero-extended value
  const std::uint64_t h = static_cast<std::uint32_t>(static_cast<std::int32_t>(id));
  return (parts & (parts - 1)) == 0 ? static_cast<std::size_t>(h & (parts - 1)) : static_cast<std::size_t>(h >= parts ? h % parts : h);
}

void addDimAttributes(CatalogRelation *rel) {
  rel->addAttribute("id", Type::Int().getNullableVersion());
  rel->addAttribute("char_col", Type::Char(20));
}
// (id INT NULL, char_col CHAR(20)) hash-partitioned `parts` ways on id (parts = 1: unpartitioned); rows with int_col != 0,
// non-NULL.  A rank loads the partitions it owns; an unpartitioned relation is dealt row by row over the ranks.
void loadDim(StorageManager *storage, CatalogRelation *rel, std::size_t parts, const Ranks &ranks) {
  addDimAttributes(rel);
  if (parts > 1) rel->setPartitionScheme(parts, 0);
  for (std::size_t p = 0; p < parts; ++p) {
    if (parts > 1 && !ranks.owns(p)) continue;
    std::vector<std::int32_t> id;
    std::vector<char> text;
    int dealt = 0;
    for (const TestRow &r : testTable()) {
      if (r.int_null || r.int_col == 0 || (parts > 1 && pid(r.int_col, parts) != p)) continue;
      if (parts == 1 && (dealt++ % ranks.world) != ranks.rank) continue;
      id.push_back(r.int_col);
      text.insert(text.end(), r.char_col, r.char_col + 20);
    }
    storage->loadBlock(rel, {id.data(), text.data()}, static_cast<std::int64_t>(id.size()), p);
  }
}
// fact (id INT NULL, score DOUBLE NULL) PARTITION BY HASH(id) PARTITIONS 4: the rows with int_col % 2 = 0
void loadFact(StorageManager *storage, CatalogRelation *rel, const Ranks &ranks) {
  rel->addAttribute("id", Type::Int().getNullableVersion());
  rel->addAttribute("score", Type::Double().getNullableVersion());
  rel->setPartitionScheme(4, 0);
  for (std::size_t p = 0; p < 4; ++p) {
    if (!ranks.owns(p)) continue;
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

// ---- result rows as text -----------------------------------------------------------------------------------------------
void collect(const std::string &tag, QueryContext &ctx, QueryContext::insert_destination_id dest, StorageManager &storage, Lines *out,
             std::vector<partition_id> *partitions = nullptr) {
  for (const InsertDestination::TouchedBlock &t : ctx.getInsertDestination(dest)->getTouchedBlocksWithPartitions()) {
    BlockReference blk = storage.getBlock(t.id);
    const CatalogRelation &rel = blk->getRelation();
    const std::size_t k = static_cast<std::size_t>(blk->numTuples());
    if (k == 0) continue;
    std::vector<std::vector<char>> cols(rel.size());
    std::vector<std::vector<std::uint64_t>> nulls(rel.size());
    for (std::size_t a = 0; a < rel.size(); ++a) {
      cols[a].resize(k * static_cast<std::size_t>(rel.getAttributeType(static_cast<attribute_id>(a)).width));
      blk->copyAttributeToHost(static_cast<attribute_id>(a), cols[a].data());
      nulls[a].assign((k + 63) / 64, 0);
      blk->copyNullBitmapToHost(static_cast<attribute_id>(a), nulls[a].data());
    }
    for (std::size_t i = 0; i < k; ++i) {
      std::ostringstream line;
      line << tag;
      for (std::size_t a = 0; a < rel.size(); ++a) {
        const Type &t2 = rel.getAttributeType(static_cast<attribute_id>(a));
        line << '|';
        if ((nulls[a][i >> 6] >> (63 - (i & 63))) & 1u) {
          line << "NULL";
          continue;
        }
        const char *v = cols[a].data() + i * static_cast<std::size_t>(t2.width);
        char buf[64];
        switch (t2.id) {
          case kInt: { std::int32_t x; std::memcpy(&x, v, 4); line << x; break; }
          case kLong: { std::int64_t x; std::memcpy(&x, v, 8); line << x; break; }
          case kDouble: { double x; std::memcpy(&x, v, 8); std::snprintf(buf, sizeof(buf), "%.17g", x); line << buf; break; }
          case kFloat: { float x; std::memcpy(&x, v, 4); std::snprintf(buf, sizeof(buf), "%.9g", x); line << buf; break; }
          default: line << std::string(v, strnlen(v, static_cast<std::size_t>(t2.width))); break;
        }
      }
      out->push_back(line.str());
      if (partitions != nullptr) partitions->push_back(t.partition);
    }
  }
}

// ---- joins of Partition.test ---------------------------------------------------------------------------------------------
enum class DimKind { kPartitioned4, kUnpartitionedBroadcast, kPartitioned2Repartitioned };
void runTestTableJoin(const std::string &tag, DimKind kind, const Ranks &ranks, Lines *out) {
  StorageManager storage;
  CatalogRelation dim(1, "dim"), fact(2, "fact"), dim_scattered(3, "dim_scattered"), dim_arrived(4, "dim_arrived"), joined(5, "out");
  loadDim(&storage, &dim, kind == DimKind::kPartitioned4 ? 4 : (kind == DimKind::kPartitioned2Repartitioned ? 2 : 1), ranks);
  loadFact(&storage, &fact, ranks);
  joined.addAttribute("id", Type::Int().getNullableVersion());
  joined.addAttribute("char_col", Type::Char(20));
  QueryContext ctx;
  const auto table = ctx.addJoinHashTable(kInt, 32, 4);
  const auto d_out = ctx.addInsertDestination(&joined, &storage);
  const auto selection = ctx.addScalarGroup({0, 1});         // fact.id (probe), dim.char_col (build)
  const std::vector<bool> on_build{false, true};
  QueryPlan plan;
  const CatalogRelation *build_rel = &dim;
  bool build_stored = true;
  std::vector<std::pair<std::size_t, std::size_t>> streaming;   // (consumer, producer)
  std::size_t last = 0;
  bool have_last = false;
  if (kind == DimKind::kPartitioned2Repartitioned) {
    addDimAttributes(&dim_scattered);
    dim_scattered.setPartitionScheme(4, 0);
    const auto d_scattered = ctx.addPartitionAwareInsertDestination(&dim_scattered, &storage);
    last = plan.addRelationalOperator(new SelectOperator(0, dim, /*has_repartition=*/true, dim_scattered, d_scattered, QueryContext::kInvalidPredicateId,
                                                         std::vector<attribute_id>{0, 1}, true));
    have_last = true;
    build_rel = &dim_scattered;
    build_stored = false;
    if (ranks.group != nullptr) {
      addDimAttributes(&dim_arrived);
      dim_arrived.setPartitionScheme(4, 0);
      const auto d_arrived = ctx.addInsertDestination(&dim_arrived, &storage);
      const std::size_t x = plan.addRelationalOperator(new PartitionExchangeOperator(0, dim_scattered, false, dim_arrived, d_arrived, ranks.group));
      streaming.emplace_back(x, last);
      last = x;
      build_rel = &dim_arrived;
    }
  } else if (kind == DimKind::kUnpartitionedBroadcast && ranks.group != nullptr) {
    addDimAttributes(&dim_arrived);
    const auto d_arrived = ctx.addInsertDestination(&dim_arrived, &storage);
    last = plan.addRelationalOperator(new PartitionExchangeOperator(0, dim, true, dim_arrived, d_arrived, ranks.group, /*broadcast=*/true));
    have_last = true;
    build_rel = &dim_arrived;
    build_stored = false;
  }
  const auto i_build = plan.addRelationalOperator(new BuildHashOperator(0, *build_rel, build_stored, {0}, true, 4, table));
  const auto i_join = plan.addRelationalOperator(new HashJoinOperator(0, *build_rel, fact, true, {0}, true, 4, false, joined, d_out, table,
                                                                      QueryContext::kInvalidPredicateId, selection, &on_build,
                                                                      HashJoinOperator::JoinType::kInnerJoin));
  for (const auto &e : streaming) plan.addDirectDependency(e.first, e.second, false);
  if (have_last) plan.addDirectDependency(i_build, last, false);
  plan.addDirectDependency(i_join, i_build, true);
  ForemanSingleNode foreman(&plan, &ctx, &storage, 3);
  foreman.run();
  std::vector<partition_id> parts;
  const std::size_t before = out->size();
  collect(tag, ctx, d_out, storage, out, &parts);
  // a joined tuple is produced by the rank that owns the partition of its key, and its block carries that partition
  for (std::size_t i = before; i < out->size(); ++i) {
    const int id = std::atoi((*out)[i].c_str() + tag.size() + 1);
    EXPECT_EQ(pid(id, 4), parts[i - before]);
    EXPECT_TRUE(ranks.owns(pid(id, 4)));
  }
}

// ---- BASELINE config 4's shape -----------------------------------------------------------------------------------------
struct C4Slice {
  std::vector<std::int32_t> o_key, l_key;
  std::vector<std::int64_t> o_pay, l_pay;
};
C4Slice c4Slice(int slice, int orders_per_slice) {   // the share one rank starts with: a contiguous key range, shuffled
  C4Slice s;
  const std::int32_t first = slice * orders_per_slice + 1;
  std::uint64_t x = 88172645463325252ull + static_cast<std::uint64_t>(slice) * 7919;
  auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (int i = 0; i < orders_per_slice; ++i) s.o_key.push_back(first + i);
  for (int i = orders_per_slice - 1; i > 0; --i) std::swap(s.o_key[static_cast<std::size_t>(i)], s.o_key[next() % static_cast<std::uint64_t>(i + 1)]);
  for (std::int32_t k : s.o_key) s.o_pay.push_back(3ll * k + 1);
  for (int i = 0; i < orders_per_slice; ++i) {   // lineitem clustered on the key, 1-7 lines per order
    const int lines = 1 + static_cast<int>(next() % 7);
    for (int l = 0; l < lines; ++l) {
      s.l_key.push_back(first + i);
      s.l_pay.push_back(5ll * (first + i) + l);
    }
  }
  return s;
}
template <typename K, typename V>
void loadInBlocks(StorageManager *storage, CatalogRelation *rel, const std::vector<K> &a, const std::vector<V> &b, int blocks) {
  const std::size_t n = a.size(), step = (n + static_cast<std::size_t>(blocks) - 1) / static_cast<std::size_t>(blocks);
  for (std::size_t at = 0; at < n; at += step) {
    const std::size_t k = std::min(step, n - at);
    storage->loadBlock(rel, {a.data() + at, b.data() + at}, static_cast<std::int64_t>(k));
  }
  storage->loadBlock(rel, {a.data(), b.data()}, 0);   // and an empty block
}
void runC4(const Ranks &ranks, int slices, int orders_per_slice, std::size_t parts, Lines *out) {
  StorageManager storage;
  CatalogRelation orders(1, "orders"), lineitem(2, "lineitem"), o_scattered(3, "o_scattered"), l_scattered(4, "l_scattered"), o_arrived(5, "o_arrived"),
      l_arrived(6, "l_arrived"), joined(7, "out");
  for (CatalogRelation *r : {&orders, &lineitem, &o_scattered, &l_scattered, &o_arrived, &l_arrived}) {
    r->addAttribute("key", Type::Int());
    r->addAttribute("payload", Type::Long());
  }
  for (CatalogRelation *r : {&o_scattered, &l_scattered, &o_arrived, &l_arrived}) r->setPartitionScheme(parts, 0);
  joined.addAttribute("key", Type::Int());
  joined.addAttribute("o_payload", Type::Long());
  joined.addAttribute("l_payload", Type::Long());
  // a rank starts with its own slice; the single-process run with all of them
  std::int64_t lines_here = 0;
  for (int s = 0; s < slices; ++s) {
    if (ranks.group != nullptr && s != ranks.rank) continue;
    const C4Slice slice = c4Slice(s, orders_per_slice);
    loadInBlocks(&storage, &orders, slice.o_key, slice.o_pay, 3);
    loadInBlocks(&storage, &lineitem, slice.l_key, slice.l_pay, 5);
    lines_here += static_cast<std::int64_t>(slice.l_key.size());
  }
  QueryContext ctx;
  const auto table = ctx.addJoinHashTable(kInt, orders_per_slice * 2, parts);
  const auto d_o = ctx.addPartitionAwareInsertDestination(&o_scattered, &storage), d_l = ctx.addPartitionAwareInsertDestination(&l_scattered, &storage);
  const auto d_out = ctx.addInsertDestination(&joined, &storage);
  const auto selection = ctx.addScalarGroup({0, 1, 1});       // probe key, build payload, probe payload
  const std::vector<bool> on_build{false, true, false};
  QueryPlan plan;
  const auto sel_o = plan.addRelationalOperator(new SelectOperator(0, orders, true, o_scattered, d_o, QueryContext::kInvalidPredicateId,
                                                                   std::vector<attribute_id>{0, 1}, true));
  const CatalogRelation *build_rel = &o_scattered, *probe_rel = &l_scattered;
  std::size_t into_build = sel_o, x_o = 0;
  PartitionExchangeOperator *exchange_o = nullptr, *exchange_l = nullptr;
  if (ranks.group != nullptr) {
    const auto d_xo = ctx.addInsertDestination(&o_arrived, &storage);
    exchange_o = new PartitionExchangeOperator(0, o_scattered, false, o_arrived, d_xo, ranks.group);
    x_o = plan.addRelationalOperator(exchange_o);
    plan.addDirectDependency(x_o, sel_o, false);
    into_build = x_o;
    build_rel = &o_arrived;
  }
  const auto i_build = plan.addRelationalOperator(new BuildHashOperator(0, *build_rel, false, {0}, false, parts, table));
  plan.addDirectDependency(i_build, into_build, false);
  const auto sel_l = plan.addRelationalOperator(new SelectOperator(0, lineitem, true, l_scattered, d_l, QueryContext::kInvalidPredicateId,
                                                                   std::vector<attribute_id>{0, 1}, true));
  std::size_t into_join = sel_l;
  if (ranks.group != nullptr) {
    const auto d_xl = ctx.addInsertDestination(&l_arrived, &storage);
    exchange_l = new PartitionExchangeOperator(0, l_scattered, false, l_arrived, d_xl, ranks.group);
    const auto x_l = plan.addRelationalOperator(exchange_l);
    plan.addDirectDependency(x_l, sel_l, false);
    into_join = x_l;
    probe_rel = &l_arrived;
  }
  HashJoinOperator *join = new HashJoinOperator(0, *build_rel, *probe_rel, false, {0}, false, parts, false, joined, d_out, table,
                                                QueryContext::kInvalidPredicateId, selection, &on_build, HashJoinOperator::JoinType::kInnerJoin);
  const auto i_join = plan.addRelationalOperator(join);
  plan.addDirectDependency(i_join, into_join, false);
  plan.addDirectDependency(i_join, i_build, true);
  ForemanSingleNode foreman(&plan, &ctx, &storage, 4);
  foreman.run();
  const std::size_t before = out->size();
  std::vector<partition_id> block_parts;
  collect("c4", ctx, d_out, storage, out, &block_parts);
  for (std::size_t i = before; i < out->size(); ++i) {
    long long key = 0, o_pay = 0, l_pay = 0;
    EXPECT_EQ(std::sscanf((*out)[i].c_str(), "c4|%lld|%lld|%lld", &key, &o_pay, &l_pay), 3);
    EXPECT_EQ(o_pay, 3 * key + 1);                                   // the join condition, from the payloads alone
    EXPECT_TRUE(l_pay - 5 * key >= 0 && l_pay - 5 * key < 7);
    EXPECT_EQ(pid(key, parts), block_parts[i - before]);
    EXPECT_TRUE(ranks.owns(pid(key, parts)));                        // produced where its partition lives
  }
  if (ranks.group != nullptr) {
    // something really left this rank: (world - 1) / world of a uniformly hashed relation
    EXPECT_TRUE(exchange_o->bytesSentToPeers() > static_cast<std::uint64_t>(orders_per_slice) * 12 / 4);
    EXPECT_TRUE(exchange_l->bytesSentToPeers() > static_cast<std::uint64_t>(lines_here) * 12 / 4);
  }
}

// ---- aggregations ---------------------------------------------------------------------------------------------------------
void runPartitionedAggregation(const Ranks &ranks, Lines *out) {   // Partition.test:135-162
  StorageManager storage;
  CatalogRelation dim(1, "dim_4_hash_partitions"), out_count(2, "count"), out_groups(3, "groups");
  loadDim(&storage, &dim, 4, ranks);
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
  QueryPlan plan;
  const auto a1 = plan.addRelationalOperator(new AggregationOperator(0, dim, true, count_state, 4));
  const auto a2 = plan.addRelationalOperator(new AggregationOperator(0, dim, true, group_state, 4));
  const auto f1 = plan.addRelationalOperator(new FinalizeAggregationOperator(0, count_state, 4, false, 1, out_count, d_count));
  const auto f2 = plan.addRelationalOperator(new FinalizeAggregationOperator(0, group_state, 4, false, 1, out_groups, d_groups));
  plan.addDirectDependency(f1, a1, true);
  plan.addDirectDependency(f2, a2, true);
  ForemanSingleNode foreman(&plan, &ctx, &storage, 3);
  foreman.run();
  // COUNT(*): one count per partition state; the reference adds them in a final aggregation — the parent does
  Lines counts;
  collect("partitioned_count", ctx, d_count, storage, &counts);
  long long total = 0;
  for (const std::string &l : counts) total += std::atoll(l.c_str() + std::strlen("partitioned_count|"));
  out->push_back("partitioned_count_partial|" + std::to_string(total));
  collect("partitioned_groups", ctx, d_groups, storage, out);
}

// lineitem-like relation (flag INT in [0, 5), key INT in [0, entries), quantity LONG, price DOUBLE multiple of 1/64) — a rank's
// slice, or all slices; group-by keys that are not partitioned: the ranks' states are merged before finalize
void runMergedAggregation(const Ranks &ranks, int slices, Lines *out) {
  const int rows_per_slice = 30'000, entries = 1'003;
  StorageManager storage;
  CatalogRelation rel(1, "items"), by_flag(2, "by_flag"), by_key(3, "by_key");
  rel.addAttribute("flag", Type::Int());
  rel.addAttribute("key", Type::Int());
  rel.addAttribute("quantity", Type::Long());
  rel.addAttribute("price", Type::Double());
  for (int s = 0; s < slices; ++s) {
    if (ranks.group != nullptr && s != ranks.rank) continue;
    std::vector<std::int32_t> flag, key;
    std::vector<std::int64_t> qty;
    std::vector<double> price;
    std::uint64_t x = 1234567 + static_cast<std::uint64_t>(s) * 104729;
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (int i = 0; i < rows_per_slice; ++i) {
      flag.push_back(static_cast<std::int32_t>(next() % 5));
      std::int32_t k = static_cast<std::int32_t>(next() % entries);
      if (k % 3 == s % 3) k = 0;                       // holes that differ by slice
      key.push_back(k);
      qty.push_back(static_cast<std::int64_t>(next() % 50) - 10);
      price.push_back(static_cast<double>(static_cast<std::int64_t>(next() % 8192) - 4096) / 64.0);
    }
    const std::size_t half = flag.size() / 2;
    storage.loadBlock(&rel, {flag.data(), key.data(), qty.data(), price.data()}, static_cast<std::int64_t>(half));
    storage.loadBlock(&rel, {flag.data() + half, key.data() + half, qty.data() + half, price.data() + half}, static_cast<std::int64_t>(flag.size() - half));
  }
  by_flag.addAttribute("flag", Type::Int());
  by_flag.addAttribute("sum_qty", Type::Long().getNullableVersion());
  by_flag.addAttribute("count", Type::Long());
  by_flag.addAttribute("min_price", Type::Double().getNullableVersion());
  by_key.addAttribute("key", Type::Int());
  by_key.addAttribute("sum_price", Type::Double().getNullableVersion());
  by_key.addAttribute("max_qty", Type::Long().getNullableVersion());
  QueryContext ctx;
  AggregationStateSpec flag_spec;
  flag_spec.input_relation = &rel;
  flag_spec.group_by = {0};
  flag_spec.aggregates = {AggregateSpec(AggregationID::kSum, 2), AggregateSpec(AggregationID::kCount, kInvalidAttributeID), AggregateSpec(AggregationID::kMin, 3)};
  flag_spec.strategy = QSX_AGG_GENERIC;
  const auto flag_state = ctx.addAggregationState(flag_spec, 1);
  AggregationStateSpec key_spec;
  key_spec.input_relation = &rel;
  key_spec.group_by = {1};
  key_spec.aggregates = {AggregateSpec(AggregationID::kSum, 3), AggregateSpec(AggregationID::kMax, 2)};
  key_spec.strategy = QSX_AGG_COLLISION_FREE;
  key_spec.collision_free_num_entries = entries;
  const auto key_state = ctx.addAggregationState(key_spec, 1);
  const auto d_flag = ctx.addInsertDestination(&by_flag, &storage), d_key = ctx.addInsertDestination(&by_key, &storage);
  const std::size_t world = static_cast<std::size_t>(ranks.world);
  QueryPlan plan;
  const auto a1 = plan.addRelationalOperator(new AggregationOperator(0, rel, true, flag_state, 1));
  const auto a2 = plan.addRelationalOperator(new AggregationOperator(0, rel, true, key_state, 1));
  std::size_t before_f1 = a1, before_f2 = a2;
  if (ranks.group != nullptr) {
    before_f1 = plan.addRelationalOperator(new ExchangeAggregationStatesOperator(0, flag_state, 1, ranks.group));
    before_f2 = plan.addRelationalOperator(new ExchangeAggregationStatesOperator(0, key_state, 1, ranks.group));
    plan.addDirectDependency(before_f1, a1, true);
    plan.addDirectDependency(before_f2, a2, true);
  }
  FinalizeAggregationOperator *fin_flag = new FinalizeAggregationOperator(0, flag_state, 1, false, world, by_flag, d_flag);
  FinalizeAggregationOperator *fin_key = new FinalizeAggregationOperator(0, key_state, 1, false, world, by_key, d_key);
  if (ranks.group != nullptr) {
    fin_flag->setRankSlice(static_cast<std::size_t>(ranks.rank));
    fin_key->setRankSlice(static_cast<std::size_t>(ranks.rank));
  }
  const auto f1 = plan.addRelationalOperator(fin_flag), f2 = plan.addRelationalOperator(fin_key);
  plan.addDirectDependency(f1, before_f1, true);
  plan.addDirectDependency(f2, before_f2, true);
  ForemanSingleNode foreman(&plan, &ctx, &storage, 3);
  foreman.run();
  collect("merged_by_flag", ctx, d_flag, storage, out);
  const std::size_t before = out->size();
  collect("merged_by_key", ctx, d_key, storage, out);
  if (ranks.group != nullptr) {   // a CollisionFreeVector slice: only keys of this rank's range
    const long long length = (entries + ranks.world - 1) / ranks.world;
    for (std::size_t i = before; i < out->size(); ++i) {
      const long long k = std::atoll((*out)[i].c_str() + std::strlen("merged_by_key|"));
      EXPECT_TRUE(k >= ranks.rank * length && k < (ranks.rank + 1) * length);
    }
  }
}

// One rank loses a block between plan construction and the exchange (a rank-local failure between collectives
// used to leave every other rank inside the all-to-all for ever).  The round must end with an error on EVERY rank, promptly
// — the failing rank with its own error, the healthy ones with QSX_ERR_COMM naming the rank that failed
// (RankGroup::agreeOn -> qsx_comm_agree) — and the communicator must still be in step afterwards.
void runFailingExchange(const Ranks &ranks) {
  StorageManager storage;
  CatalogRelation scattered(1, "scattered"), arrived(2, "arrived");
  addDimAttributes(&scattered);
  scattered.setPartitionScheme(4, 0);
  addDimAttributes(&arrived);
  arrived.setPartitionScheme(4, 0);
  std::vector<block_id> mine;
  for (std::size_t p = 0; p < 4; ++p) {       // every rank holds a piece of EVERY partition: the exchange really moves tuples
    std::vector<std::int32_t> id;
    std::vector<char> text;
    int dealt = 0;
    for (const TestRow &r : testTable()) {
      if (r.int_null || r.int_col == 0 || pid(r.int_col, 4) != p) continue;
      if ((dealt++ % ranks.world) != ranks.rank) continue;
      id.push_back(r.int_col);
      text.insert(text.end(), r.char_col, r.char_col + 20);
    }
    // (the test table has fewer rows per partition than eight ranks: one row of the rank's own, so that every rank — the one
    // that is about to lose a block included — holds something)
    if (pid(1000 + ranks.rank, 4) == p) {
      char own[20] = {};
      std::snprintf(own, sizeof(own), "rank %d", ranks.rank);
      id.push_back(1000 + ranks.rank);
      text.insert(text.end(), own, own + 20);
    }
    if (!id.empty()) mine.push_back(storage.loadBlock(&scattered, {id.data(), text.data()}, static_cast<std::int64_t>(id.size()), p));
  }
  QueryContext ctx;
  const auto d_arrived = ctx.addInsertDestination(&arrived, &storage);
  QueryPlan plan;
  plan.addRelationalOperator(new PartitionExchangeOperator(0, scattered, true, arrived, d_arrived, ranks.group));
  const int failing = ranks.world - 1;
  if (ranks.rank == failing) {
    EXPECT_TRUE(!mine.empty());
    if (!mine.empty()) storage.deleteBlockOrBlobFile(mine.front());
  }
  bool threw = false;
  std::string what;
  const auto t0 = std::chrono::steady_clock::now();
  try {
    ForemanSingleNode foreman(&plan, &ctx, &storage, 2);
    foreman.run();
  } catch (const std::exception &e) {
    threw = true;
    what = e.what();
  }
  const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  EXPECT_TRUE(threw);
  EXPECT_TRUE(seconds < 30.0);
  if (ranks.rank == failing) {
    EXPECT_TRUE(what.find("unknown block") != std::string::npos);
  } else {
    EXPECT_TRUE(what.find("rank " + std::to_string(failing) + " failed") != std::string::npos);
  }
  std::printf("rank %d: the failed round ended after %.2f s with '%s'\n", ranks.rank, seconds, what.c_str());
  bool in_step = true;
  try {
    ranks.group->agreeOn([]() {}, "after the failed round");
  } catch (const std::exception &e) {
    in_step = false;
    std::fprintf(stderr, "rank %d: the communicator is out of step after the failed round: %s\n", ranks.rank, e.what());
  }
  EXPECT_TRUE(in_step);
}

void runAll(const Ranks &ranks, int slices, Lines *out) {
  runTestTableJoin("partitioned_join", DimKind::kPartitioned4, ranks, out);
  runTestTableJoin("broadcast_join", DimKind::kUnpartitionedBroadcast, ranks, out);
  runTestTableJoin("repartitioned_join", DimKind::kPartitioned2Repartitioned, ranks, out);
  // C4: P = the partition count of the machine this is for (world 8: pid = h & 7, catalog/PartitionSchemeHeader.hpp:207-214, one
  // exchange round) and a scheme with more partitions than ranks (two rounds at world 2, 3 and 4; at world 8: 16 partitions)
  const std::size_t parts = slices <= 4 ? 4 : 8;
  runC4(ranks, slices, slices <= 4 ? 20'000 : 6'000, parts, out);
  if (slices >= 4) runC4(ranks, slices, 3'000, 2 * static_cast<std::size_t>(slices), out);
  runPartitionedAggregation(ranks, out);
  runMergedAggregation(ranks, slices, out);
}

// ---- processes -----------------------------------------------------------------------------------------------------------
std::string selfPath() {
  char buf[4096];
  const ssize_t n = readlink("/proc/self/exe", buf, sizeof(buf) - 1);
  return n > 0 ? std::string(buf, static_cast<std::size_t>(n)) : std::string();
}

int rankMain(int world, int rank, const std::string &dir) {
  if (qsx_device_count() < 1) return 2;
  const std::string id_path = dir + "/id_w" + std::to_string(world);
  std::vector<unsigned char> id(QSX_COMM_ID_BYTES);
  if (rank == 0) {
    id = RankGroup::MakeUniqueId();
    std::ofstream(id_path + ".tmp", std::ios::binary).write(reinterpret_cast<const char *>(id.data()), static_cast<std::streamsize>(id.size()));
    std::rename((id_path + ".tmp").c_str(), id_path.c_str());
  } else {
    bool got = false;
    for (int tries = 0; tries < 1200 && !got; ++tries) {
      std::ifstream f(id_path, std::ios::binary);
      if (f && f.read(reinterpret_cast<char *>(id.data()), static_cast<std::streamsize>(id.size()))) got = true;
      else std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    if (!got) {
      std::fprintf(stderr, "rank %d: no communicator id from rank 0\n", rank);
      return 3;
    }
  }
  RankGroup group(world, rank, id.data());
  Ranks ranks;
  ranks.world = world;
  ranks.rank = rank;
  ranks.group = &group;
  Lines lines, lines_again;
  runAll(ranks, world, &lines);
  runFailingExchange(ranks);
  runAll(ranks, world, &lines_again);      // ... and every plan still runs on the same communicator afterwards
  {
    Lines a = lines, b = lines_again;            // (row order depends on which Worker returns its block first)
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    EXPECT_TRUE(a == b);
  }
  std::ofstream out(dir + "/w" + std::to_string(world) + "_r" + std::to_string(rank) + ".txt");
  for (const std::string &l : lines) out << l << "\n";
  out.close();
  std::printf("rank %d of %d: %zu result rows, %d failures\n", rank, world, lines.size(), g_failures);
  return g_failures == 0 ? 0 : 1;
}

bool startRanks(int world, const std::string &dir, const std::string &loopback) {
  const std::string self = selfPath();
  std::vector<std::string> env_text;
  for (char **e = environ; *e != nullptr; ++e) {
    if (std::strncmp(*e, "QSX_RCCL_LIBRARY=", 17) != 0) env_text.push_back(*e);
  }
  env_text.push_back("QSX_RCCL_LIBRARY=" + loopback);
  env_text.push_back("QSX_ALLOW_TEST_TRANSPORT=1");
  std::vector<char *> envp;
  for (std::string &s : env_text) envp.push_back(&s[0]);
  envp.push_back(nullptr);
  std::vector<pid_t> children;
  for (int r = 0; r < world; ++r) {
    const std::string w = std::to_string(world), rk = std::to_string(r);
    const char *argv[] = {self.c_str(), "--rank", w.c_str(), rk.c_str(), dir.c_str(), nullptr};
    pid_t pid = 0;
    if (posix_spawn(&pid, self.c_str(), nullptr, nullptr, const_cast<char *const *>(argv), envp.data()) != 0) return false;
    children.push_back(pid);
  }
  bool ok = true;
  for (pid_t pid : children) {
    int status = 0;
    if (waitpid(pid, &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) ok = false;
  }
  return ok;
}

Lines readLines(const std::string &path) {
  Lines lines;
  std::ifstream f(path);
  for (std::string l; std::getline(f, l);) lines.push_back(l);
  return lines;
}

// count lines: the ranks' partial COUNT(*)s add up to the single process's (which also reports a sum over its partitions)
Lines normalised(Lines lines) {
  long long count = 0;
  Lines out;
  for (const std::string &l : lines) {
    if (l.compare(0, 26, "partitioned_count_partial|") == 0) count += std::atoll(l.c_str() + 26);
    else out.push_back(l);
  }
  out.push_back("partitioned_count|" + std::to_string(count));
  std::sort(out.begin(), out.end());
  return out;
}
}  // namespace

int main(int argc, char **argv) {
  if (argc == 5 && std::strcmp(argv[1], "--rank") == 0) return rankMain(std::atoi(argv[2]), std::atoi(argv[3]), argv[4]);
  // the parent: rank processes first (this process has not touched the GPU yet), then the single-process reference
  char dir_template[] = "/tmp/qsx_ranks_XXXXXX";
  const char *dir = mkdtemp(dir_template);
  if (dir == nullptr) return 3;
  const std::string self = selfPath();
  const std::string loopback = self.substr(0, self.rfind('/')) + "/libloopback_rccl.so";
  // 2 and 3 (not a power of two: h % P), 4, and 8 = the node this is built for (QSX_RANKS_TEST_WORLDS="2,3": a subset)
  std::vector<int> worlds = {2, 3, 4, 8};
  if (const char *e = std::getenv("QSX_RANKS_TEST_WORLDS")) {
    worlds.clear();
    for (const char *p = e; *p != 0;) {
      char *end = nullptr;
      const long w = std::strtol(p, &end, 10);
      if (end == p) break;
      if (w >= 2 && w <= 16) worlds.push_back(static_cast<int>(w));
      p = *end == ',' ? end + 1 : end;
    }
  }
  std::map<int, bool> ran;
  for (int world : worlds) ran[world] = startRanks(world, dir, loopback);
  if (qsx_device_count() < 1) {
    std::fprintf(stderr, "partitioned_ranks_test needs an MI355X: %s\n", qsx_status_string(QSX_ERR_NO_DEVICE));
    return 2;
  }
  for (int world : worlds) {
    EXPECT_TRUE(ran[world]);
    Ranks alone;
    Lines reference;
    runAll(alone, world, &reference);     // all `world` slices in one process, no exchange operator in any plan
    Lines got;
    for (int r = 0; r < world; ++r) {
      const Lines part = readLines(std::string(dir) + "/w" + std::to_string(world) + "_r" + std::to_string(r) + ".txt");
      got.insert(got.end(), part.begin(), part.end());
    }
    const Lines want = normalised(reference), have = normalised(got);
    EXPECT_EQ(have.size(), want.size());
    EXPECT_TRUE(have == want);
    if (have != want) {
      std::size_t shown = 0;
      for (std::size_t i = 0; i < std::min(have.size(), want.size()) && shown < 10; ++i) {
        if (have[i] != want[i]) {
          std::fprintf(stderr, "world %d: row %zu: ranks '%s' vs single process '%s'\n", world, i, have[i].c_str(), want[i].c_str());
          ++shown;
        }
      }
    }
    std::size_t c4_rows = 0, join_rows = 0;
    for (const std::string &l : have) {
      c4_rows += l.compare(0, 3, "c4|") == 0;
      join_rows += l.compare(0, 17, "partitioned_join|") == 0;
    }
    EXPECT_EQ(join_rows, static_cast<std::size_t>(10));            // Partition.test:75-92: ten joined rows
    EXPECT_TRUE(c4_rows > static_cast<std::size_t>(world) * (world <= 4 ? 20'000 : 6'000));   // one output row per lineitem row
    std::printf("world %d: %zu result rows over the ranks = the single-process operators' (%zu joined C4 rows)\n", world, have.size(), c4_rows);
  }
  return finish("partitioned_ranks_test");
}
