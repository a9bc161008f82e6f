// NULLs through the GPU work orders.  Known answers of the reference:
//   * query_optimizer/tests/execution_generator/Join.test:136-196 — two LEFT JOIN chains over a / b / c / d; in the
//     second one the padded rows of a join are the probe side of the next (NULL join keys match nothing);
//   * query_optimizer/tests/execution_generator/Select.test:582-623 over the 25-row test table of
//     query_optimizer/tests/TestDatabaseLoader.cpp:118-170 (int_col, double_col NULL where x % 10 == 0): GROUP BY on a
//     nullable key, COUNT / SUM / AVG / MIN / MAX skipping NULL arguments;
// plus semi / anti joins and a selection over NULLs, checked against what the reference's loops do row by row
// (HashTable.hpp:2158-2160 NULL keys are not looked up; LiteralComparators-inl.hpp:330-370 a comparison with NULL is
// not true; HashAntiJoinWorkOrder keeps what was never matched, HashJoinOperator.cpp:860-877).
#include <algorithm>
#include <cmath>
#include <map>

#include "test_util.hpp"

using namespace quickstep;

namespace {

struct Column {
  std::vector<std::int64_t> values;
  std::vector<bool> is_null;
};

std::vector<std::uint64_t> Bitmap(const std::vector<bool> &bits) {
  std::vector<std::uint64_t> words((bits.size() + 63) / 64 + 1, 0);
  for (std::size_t i = 0; i < bits.size(); ++i) {
    if (bits[i]) words[i >> 6] |= 1ull << (63 - (i & 63));
  }
  return words;
}

// All rows of attribute `a` (LONG) of a relation, with null flags, in block order.
Column ReadLong(const CatalogRelation &rel, StorageManager &storage, attribute_id a) {
  Column c;
  for (block_id b : rel.getBlocksSnapshot()) {
    BlockReference blk = storage.getBlock(b);
    const std::size_t at = c.values.size(), k = static_cast<std::size_t>(blk->numTuples());
    c.values.resize(at + k);
    blk->copyAttributeToHost(a, c.values.data() + at);
    std::vector<std::uint64_t> nulls((k + 63) / 64 + 1, 0);
    blk->copyNullBitmapToHost(a, nulls.data());
    for (std::size_t i = 0; i < k; ++i) c.is_null.push_back((nulls[i >> 6] >> (63 - (i & 63))) & 1u);
  }
  return c;
}

// One LEFT OUTER / INNER / SEMI / ANTI hash join of `probe` with `build` on probe.key = build.key; the output relation
// (attributes: all of probe, then `build_selection` of build, nullable for the outer join) is registered in `out`.
void Join(StorageManager *storage, const CatalogRelation &build, attribute_id build_key, const CatalogRelation &probe,
          attribute_id probe_key, const std::vector<attribute_id> &build_selection, HashJoinOperator::JoinType type,
          CatalogRelation *out, bool use_foreman) {
  QueryContext ctx;
  std::vector<attribute_id> selection;
  std::vector<bool> on_build;
  const bool probe_only = type == HashJoinOperator::JoinType::kLeftSemiJoin || type == HashJoinOperator::JoinType::kLeftAntiJoin;
  for (std::size_t a = 0; a < probe.size(); ++a) {
    out->addAttribute("p" + std::to_string(a), probe.getAttributeType(static_cast<attribute_id>(a)));
    selection.push_back(static_cast<attribute_id>(a));
    on_build.push_back(false);
  }
  for (attribute_id a : probe_only ? std::vector<attribute_id>() : build_selection) {
    out->addAttribute("b" + std::to_string(a), build.getAttributeType(a).getNullableVersion());
    selection.push_back(a);
    on_build.push_back(true);
  }
  const auto table = ctx.addJoinHashTable(kLong, 64);
  const auto dest = ctx.addInsertDestination(out, storage);
  const auto sel = ctx.addScalarGroup(selection);
  auto *builder = new BuildHashOperator(0, build, true, {build_key}, build.getAttributeType(build_key).nullable, 1, table);
  auto *prober = new HashJoinOperator(0, build, probe, true, {probe_key}, probe.getAttributeType(probe_key).nullable, 1, false, *out,
                                      dest, table, QueryContext::kInvalidPredicateId, sel, &on_build, type);
  auto *cleaner = new DestroyHashOperator(0, 1, table);
  if (use_foreman) {
    QueryPlan plan;
    const auto bi = plan.addRelationalOperator(builder);
    const auto pi = plan.addRelationalOperator(prober);
    const auto ci = plan.addRelationalOperator(cleaner);
    plan.addDirectDependency(pi, bi, true);
    plan.addDirectDependency(ci, pi, true);
    ForemanSingleNode foreman(&plan, &ctx, storage, 3);
    foreman.run();
  } else {
    std::unique_ptr<RelationalOperator> b(builder), p(prober), c(cleaner);
    fetchAndExecuteWorkOrders(b.get(), &ctx, storage);
    fetchAndExecuteWorkOrders(p.get(), &ctx, storage);
    fetchAndExecuteWorkOrders(c.get(), &ctx, storage);
  }
}

// Join.test:19-57: a(w, x, y) = (i, 10 i, 100 i); b = (w, x + (w/2)%2) of the even w; c = (x, y + (x/3)%3 - 1) of the x
// divisible by 3; d = (y, w) [z = 'C<w>' carried as w].  DOUBLE columns hold whole numbers: carried as LONG.
struct JoinTestTables {
  StorageManager storage;
  CatalogRelation a{1, "a"}, b{2, "b"}, c{3, "c"}, d{4, "d"};
  JoinTestTables() {
    for (const char *n : {"w", "x", "y"}) a.addAttribute(n, Type::Long());
    for (const char *n : {"w", "x"}) b.addAttribute(n, Type::Long());
    for (const char *n : {"x", "y"}) c.addAttribute(n, Type::Long());
    for (const char *n : {"y", "z_w"}) d.addAttribute(n, Type::Long());
    std::vector<std::int64_t> aw, ax, ay, bw, bx, cx, cy;
    for (std::int64_t i = 0; i < 20; ++i) {
      aw.push_back(i); ax.push_back(10 * i); ay.push_back(100 * i);
      if (i % 2 == 0) { bw.push_back(i); bx.push_back(10 * i + (i / 2) % 2); }
      if ((10 * i) % 3 == 0) { cx.push_back(10 * i); cy.push_back(100 * i + (10 * i / 3) % 3 - 1); }
    }
    // a in blocks of 7 rows: several probe work orders
    for (std::size_t at = 0; at < aw.size(); at += 7) {
      const std::int64_t k = static_cast<std::int64_t>(std::min<std::size_t>(7, aw.size() - at));
      storage.loadBlock(&a, {aw.data() + at, ax.data() + at, ay.data() + at}, k);
    }
    storage.loadBlock(&b, {bw.data(), bx.data()}, static_cast<std::int64_t>(bw.size()));
    storage.loadBlock(&c, {cx.data(), cy.data()}, static_cast<std::int64_t>(cx.size()));
    storage.loadBlock(&d, {ay.data(), aw.data()}, static_cast<std::int64_t>(ay.size()));
  }
};

constexpr std::int64_t kNull = -999999;   // how the expectations below spell NULL

void ExpectColumn(const Column &got, const Column &row_key, const std::vector<std::int64_t> &expected_by_w, const char *what) {
  EXPECT_EQ(got.values.size(), expected_by_w.size());
  for (std::size_t i = 0; i < got.values.size() && i < row_key.values.size(); ++i) {
    const std::int64_t w = row_key.values[i];
    const std::int64_t want = expected_by_w.at(static_cast<std::size_t>(w));
    if (want == kNull) {
      if (!got.is_null[i]) { std::fprintf(stderr, "%s: row w=%lld should be NULL\n", what, static_cast<long long>(w)); ++g_failures; }
    } else {
      if (got.is_null[i] || got.values[i] != want) {
        std::fprintf(stderr, "%s: row w=%lld: got %s%lld, want %lld\n", what, static_cast<long long>(w), got.is_null[i] ? "NULL/" : "",
                     static_cast<long long>(got.values[i]), static_cast<long long>(want));
        ++g_failures;
      }
    }
  }
}

void TestLeftJoinChains(bool use_foreman) {
  using JT = HashJoinOperator::JoinType;
  const std::int64_t N = kNull;
  const std::vector<std::int64_t> b_x = {0, N, 21, N, 40, N, 61, N, 80, N, 101, N, 120, N, 141, N, 160, N, 181, N};
  {  // Join.test:136-165: every join probes with an attribute of a
    JoinTestTables t;
    CatalogRelation ab(10, "ab"), abc(11, "abc"), abcd(12, "abcd");
    Join(&t.storage, t.b, 0, t.a, 0, {1}, JT::kLeftOuterJoin, &ab, use_foreman);        // a.w = b.w      -> (w, x, y, b.x)
    Join(&t.storage, t.c, 0, ab, 1, {1}, JT::kLeftOuterJoin, &abc, use_foreman);        // a.x = c.x      -> (..., c.y)
    Join(&t.storage, t.d, 0, abc, 2, {1}, JT::kLeftOuterJoin, &abcd, use_foreman);      // a.y = d.y      -> (..., d.z)
    const Column w = ReadLong(abcd, t.storage, 0);
    EXPECT_EQ(w.values.size(), static_cast<std::size_t>(20));
    ExpectColumn(ReadLong(abcd, t.storage, 3), w, b_x, "chain 1 b.x");
    ExpectColumn(ReadLong(abcd, t.storage, 4), w,
                 {-1, N, N, 300, N, N, 601, N, N, 899, N, N, 1200, N, N, 1501, N, N, 1799, N}, "chain 1 c.y");
    std::vector<std::int64_t> every_w(20);
    for (std::int64_t i = 0; i < 20; ++i) every_w[i] = i;
    ExpectColumn(ReadLong(abcd, t.storage, 5), w, every_w, "chain 1 d.z");
  }
  {  // Join.test:167-196: ... LEFT JOIN c ON b.x = c.x LEFT JOIN d ON c.y = d.y — NULL probe keys
    JoinTestTables t;
    CatalogRelation ab(10, "ab"), abc(11, "abc"), abcd(12, "abcd");
    Join(&t.storage, t.b, 0, t.a, 0, {1}, JT::kLeftOuterJoin, &ab, use_foreman);        // (w, x, y, b.x?)
    Join(&t.storage, t.c, 0, ab, 3, {1}, JT::kLeftOuterJoin, &abc, use_foreman);        // b.x = c.x  -> (..., c.y?)
    Join(&t.storage, t.d, 0, abc, 4, {1}, JT::kLeftOuterJoin, &abcd, use_foreman);      // c.y = d.y  -> (..., d.z?)
    const Column w = ReadLong(abcd, t.storage, 0);
    EXPECT_EQ(w.values.size(), static_cast<std::size_t>(20));
    ExpectColumn(ReadLong(abcd, t.storage, 3), w, b_x, "chain 2 b.x");
    ExpectColumn(ReadLong(abcd, t.storage, 4), w, {-1, N, N, N, N, N, N, N, N, N, N, N, 1200, N, N, N, N, N, N, N}, "chain 2 c.y");
    ExpectColumn(ReadLong(abcd, t.storage, 5), w, {N, N, N, N, N, N, N, N, N, N, N, N, 12, N, N, N, N, N, N, N}, "chain 2 d.z");
    // The same probe side through the other join types.  ab.b_x is NULL for odd w; c.x holds 0, 30, .., 180:
    // matches for w = 0 (b.x 0) and w = 12 (b.x 120).
    CatalogRelation inner(20, "inner"), semi(21, "semi"), anti(22, "anti");
    Join(&t.storage, t.c, 0, ab, 3, {1}, JT::kInnerJoin, &inner, use_foreman);
    Join(&t.storage, t.c, 0, ab, 3, {1}, JT::kLeftSemiJoin, &semi, use_foreman);
    Join(&t.storage, t.c, 0, ab, 3, {1}, JT::kLeftAntiJoin, &anti, use_foreman);
    Column iw = ReadLong(inner, t.storage, 0), sw = ReadLong(semi, t.storage, 0), aw = ReadLong(anti, t.storage, 0);
    std::sort(iw.values.begin(), iw.values.end());
    std::sort(sw.values.begin(), sw.values.end());
    std::sort(aw.values.begin(), aw.values.end());
    EXPECT_TRUE(iw.values == std::vector<std::int64_t>({0, 12}));
    EXPECT_TRUE(sw.values == std::vector<std::int64_t>({0, 12}));
    std::vector<std::int64_t> rest;   // the anti join keeps the tuples with a NULL key: they were never matched
    for (std::int64_t i = 0; i < 20; ++i) if (i != 0 && i != 12) rest.push_back(i);
    EXPECT_TRUE(aw.values == rest);
    // the projected nullable attribute keeps its null bits through the semi / anti projection
    const Column anti_bx = ReadLong(anti, t.storage, 3), anti_w = ReadLong(anti, t.storage, 0);
    for (std::size_t i = 0; i < anti_w.values.size(); ++i) EXPECT_EQ(static_cast<int>(anti_bx.is_null[i]), static_cast<int>(anti_w.values[i] % 2 == 1));
    // a build side with NULL keys: those tuples are not inserted (HashTable.hpp:1409-1418): a JOIN ab ON a.x = ab.b_x
    CatalogRelation rev(23, "rev");
    Join(&t.storage, ab, 3, t.a, 1, {0}, JT::kInnerJoin, &rev, use_foreman);
    Column rw = ReadLong(rev, t.storage, 0);
    std::sort(rw.values.begin(), rw.values.end());
    EXPECT_TRUE(rw.values == std::vector<std::int64_t>({0, 4, 8, 12, 16}));   // b.x = 0, 40, 80, 120, 160 are multiples of 10
  }
}

// TestDatabaseLoader.cpp:118-170: x = 0..24, int = (-1)^x x (NULL when x % 10 == 0), long = x^2, float = sqrt(x),
// double = (-1)^x x sqrt(x) (NULL when x % 10 == 0).  The stripes hold garbage under the NULLs.
struct TestTable {
  StorageManager storage;
  CatalogRelation test{1, "test"};
  TestTable() {
    test.addAttribute("int_col", Type::Int().getNullableVersion());
    test.addAttribute("long_col", Type::Long());
    test.addAttribute("float_col", Type::Float());
    test.addAttribute("double_col", Type::Double().getNullableVersion());
    std::vector<std::int32_t> ints;
    std::vector<std::int64_t> longs;
    std::vector<float> floats;
    std::vector<double> doubles;
    std::vector<bool> nulls;
    for (int x = 0; x < 25; ++x) {
      const bool is_null = x % 10 == 0;
      const int sign = x % 2 == 0 ? 1 : -1;
      ints.push_back(is_null ? 1 << 30 : sign * x);
      longs.push_back(static_cast<std::int64_t>(x) * x);
      floats.push_back(std::sqrt(static_cast<float>(x)));
      doubles.push_back(is_null ? 1e300 : sign * x * std::sqrt(static_cast<double>(x)));
      nulls.push_back(is_null);
    }
    // two blocks (13 + 12 rows): the second block's bitmap starts at its own row 0
    for (std::size_t at : {std::size_t(0), std::size_t(13)}) {
      const std::size_t k = at == 0 ? 13 : 12;
      const std::vector<std::uint64_t> bits = Bitmap(std::vector<bool>(nulls.begin() + at, nulls.begin() + at + k));
      const std::vector<const std::uint64_t *> bitmaps = {bits.data(), nullptr, nullptr, bits.data()};
      storage.loadBlock(&test, {ints.data() + at, longs.data() + at, floats.data() + at, doubles.data() + at},
                        static_cast<std::int64_t>(k), 0, nullptr, &bitmaps);
    }
  }
};

void RunAggregation(TestTable *t, const AggregationStateSpec &spec, CatalogRelation *result, bool use_foreman) {
  QueryContext ctx;
  const auto state = ctx.addAggregationState(spec);
  const auto dest = ctx.addInsertDestination(result, &t->storage);
  auto *agg = new AggregationOperator(0, t->test, true, state, 1);
  auto *fin = new FinalizeAggregationOperator(0, state, 1, false, 1, *result, dest);
  if (use_foreman) {
    QueryPlan plan;
    const auto ai = plan.addRelationalOperator(agg);
    const auto fi = plan.addRelationalOperator(fin);
    plan.addDirectDependency(fi, ai, true);
    ForemanSingleNode foreman(&plan, &ctx, &t->storage, 3);
    foreman.run();
  } else {
    std::unique_ptr<RelationalOperator> a(agg), f(fin);
    fetchAndExecuteWorkOrders(a.get(), &ctx, &t->storage);
    fetchAndExecuteWorkOrders(f.get(), &ctx, &t->storage);
  }
}

template <typename T>
std::vector<T> ReadAll(const CatalogRelation &rel, StorageManager &storage, attribute_id a, std::vector<bool> *is_null = nullptr) {
  std::vector<T> v;
  for (block_id b : rel.getBlocksSnapshot()) {
    BlockReference blk = storage.getBlock(b);
    const std::size_t at = v.size(), k = static_cast<std::size_t>(blk->numTuples());
    v.resize(at + k);
    blk->copyAttributeToHost(a, v.data() + at);
    if (is_null != nullptr) {
      std::vector<std::uint64_t> nulls((k + 63) / 64 + 1, 0);
      blk->copyNullBitmapToHost(a, nulls.data());
      for (std::size_t i = 0; i < k; ++i) is_null->push_back((nulls[i >> 6] >> (63 - (i & 63))) & 1u);
    }
  }
  return v;
}

void TestAggregatesOverNulls(bool use_foreman) {
  {  // Select.test:609-623 (the expressions + 0 / + 100 / + 1 are applied to the results here: the host layer aggregates
     // attributes): COUNT(*) = 25, SUM(int_col) / 25 = 0, AVG(int_col) * 25 = -20.4545..., MAX(double_col) + 100 =
     // 217.5755..., MIN(float_col) + 1 = 1; COUNT(int_col) = 22
    TestTable t;
    AggregationStateSpec spec;
    spec.input_relation = &t.test;
    spec.aggregates = {{AggregationID::kCount, kInvalidAttributeID}, {AggregationID::kSum, 0}, {AggregationID::kAvg, 0},
                       {AggregationID::kMax, 3}, {AggregationID::kMin, 2}, {AggregationID::kCount, 0}};
    CatalogRelation result(2, "result");
    result.addAttribute("count_star", Type::Long());
    result.addAttribute("sum_int", Type::Long().getNullableVersion());
    result.addAttribute("avg_int", Type::Double().getNullableVersion());
    result.addAttribute("max_double", Type::Double().getNullableVersion());
    result.addAttribute("min_float", Type::Float().getNullableVersion());
    result.addAttribute("count_int", Type::Long());
    RunAggregation(&t, spec, &result, use_foreman);
    std::vector<bool> sum_null, avg_null;
    const auto count_star = ReadAll<std::int64_t>(result, t.storage, 0);
    const auto sum_int = ReadAll<std::int64_t>(result, t.storage, 1, &sum_null);
    const auto avg_int = ReadAll<double>(result, t.storage, 2, &avg_null);
    const auto max_double = ReadAll<double>(result, t.storage, 3);
    const auto min_float = ReadAll<float>(result, t.storage, 4);
    const auto count_int = ReadAll<std::int64_t>(result, t.storage, 5);
    EXPECT_EQ(count_star.size(), static_cast<std::size_t>(1));
    EXPECT_EQ(count_star.at(0), 25);
    EXPECT_EQ(count_int.at(0), 22);
    EXPECT_EQ(sum_int.at(0) / count_star.at(0), 0);
    EXPECT_EQ(sum_int.at(0), -18);
    EXPECT_NEAR(avg_int.at(0) * count_star.at(0), -20.454545454545457, 1e-12);
    EXPECT_NEAR(max_double.at(0) + 100, 217.57550765359252, 1e-10);
    EXPECT_NEAR(min_float.at(0) + 1, 1.0, 1e-6);
    EXPECT_TRUE(!sum_null.at(0) && !avg_null.at(0));
  }
  {  // Select.test:582-607: SELECT int_col FROM test GROUP BY int_col — 22 groups, none for NULL; with COUNT(*) = 1 each
     // and (not in the reference's listing, same rule) SUM(double_col) per group
    TestTable t;
    AggregationStateSpec spec;
    spec.input_relation = &t.test;
    spec.group_by = {0};
    spec.aggregates = {{AggregationID::kCount, kInvalidAttributeID}, {AggregationID::kSum, 3}};
    spec.strategy = QSX_AGG_GENERIC;
    spec.estimated_num_groups = 32;
    CatalogRelation result(2, "result");
    result.addAttribute("int_col", Type::Int());
    result.addAttribute("count", Type::Long());
    result.addAttribute("sum_double", Type::Double().getNullableVersion());
    RunAggregation(&t, spec, &result, use_foreman);
    auto keys = ReadAll<std::int32_t>(result, t.storage, 0);
    const auto counts = ReadAll<std::int64_t>(result, t.storage, 1);
    std::sort(keys.begin(), keys.end());
    const std::vector<std::int32_t> want = {-23, -21, -19, -17, -15, -13, -11, -9, -7, -5, -3, -1, 2, 4, 6, 8, 12, 14, 16, 18, 22, 24};
    EXPECT_TRUE(keys == want);
    for (std::int64_t c : counts) EXPECT_EQ(c, 1);
  }
  {  // GROUP BY long_col / 100 is an expression; the same NULL rule with the key long_col % 2 (parity of x):
     // SUM(int_col) of the even x = 2 + 4 + ... + 24 minus the NULL rows 10, 20 = 126, of the odd x = -(1 + 3 + ... + 23) = -144;
     // and a predicate on the nullable attribute: int_col > 0 is not true for NULL
    TestTable t;
    Predicate positive;
    positive.conjuncts.emplace_back(0, ComparisonID::kGreater, TypedLiteral::Int(0));
    AggregationStateSpec spec;
    spec.input_relation = &t.test;
    spec.aggregates = {{AggregationID::kCount, kInvalidAttributeID}, {AggregationID::kSum, 0}};
    spec.predicate = &positive;
    CatalogRelation result(2, "result");
    result.addAttribute("count", Type::Long());
    result.addAttribute("sum", Type::Long().getNullableVersion());
    RunAggregation(&t, spec, &result, use_foreman);
    EXPECT_EQ(ReadAll<std::int64_t>(result, t.storage, 0).at(0), 10);     // x = 2, 4, 6, 8, 12, 14, 16, 18, 22, 24
    EXPECT_EQ(ReadAll<std::int64_t>(result, t.storage, 1).at(0), 126);
  }
}

void TestSelectOverNulls() {
  // SELECT int_col, long_col FROM test WHERE double_col < 0: NULL double_col is not < 0 although the stripe holds garbage;
  // then WHERE long_col >= 100 keeps rows 10..24 whose int_col null bits (rows 10, 20) travel with the values
  TestTable t;
  for (int variant = 0; variant < 2; ++variant) {
    QueryContext ctx;
    Predicate p;
    if (variant == 0) p.conjuncts.emplace_back(3, ComparisonID::kLess, TypedLiteral::Double(0.0));
    else p.conjuncts.emplace_back(1, ComparisonID::kGreaterOrEqual, TypedLiteral::Long(100));
    const auto pred = ctx.addPredicate(p);
    CatalogRelation out(5 + variant, "out");
    out.addAttribute("int_col", Type::Int().getNullableVersion());
    out.addAttribute("long_col", Type::Long());
    const auto dest = ctx.addInsertDestination(&out, &t.storage);
    SelectOperator select(0, t.test, false, out, dest, pred, std::vector<attribute_id>{0, 1}, true);
    fetchAndExecuteWorkOrders(&select, &ctx, &t.storage);
    std::vector<bool> is_null;
    const auto ints = ReadAll<std::int32_t>(out, t.storage, 0, &is_null);
    const auto longs = ReadAll<std::int64_t>(out, t.storage, 1);
    if (variant == 0) {
      EXPECT_EQ(ints.size(), static_cast<std::size_t>(12));   // the odd x
      for (std::size_t i = 0; i < ints.size(); ++i) EXPECT_TRUE(!is_null[i] && ints[i] < 0 && longs[i] == static_cast<std::int64_t>(ints[i]) * ints[i]);
    } else {
      EXPECT_EQ(ints.size(), static_cast<std::size_t>(15));
      for (std::size_t i = 0; i < ints.size(); ++i) {
        const std::int64_t x = static_cast<std::int64_t>(std::llround(std::sqrt(static_cast<double>(longs[i]))));
        EXPECT_EQ(static_cast<int>(is_null[i]), static_cast<int>(x % 10 == 0));
        if (!is_null[i]) EXPECT_EQ(ints[i], static_cast<std::int32_t>(x % 2 == 0 ? x : -x));
      }
    }
  }
}

}  // namespace

int main() {
  if (qsx_device_count() < 1) {
    std::fprintf(stderr, "nullable_operator_test: %s\n", qsx_status_string(QSX_ERR_NO_DEVICE));
    return 2;
  }
  for (bool use_foreman : {false, true}) {
    TestLeftJoinChains(use_foreman);
    TestAggregatesOverNulls(use_foreman);
  }
  TestSelectOverNulls();
  return finish("nullable_operator_test");
}
