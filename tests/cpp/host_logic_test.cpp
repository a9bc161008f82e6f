// Host logic of the operator layer that needs no device: CompressedBlockBuilder's per-attribute choice, the rewrite of
// comparisons into code comparisons (CompressedStoreUtil.cpp:51-140, 425-616) checked row by row against the comparison on
// the decoded values, partition-scheme bookkeeping, work-order container order.  Runs in the CPU test suite.
#include <algorithm>
#include <cstring>

#include "test_util.hpp"

using namespace quickstep;

namespace {
std::uint64_t g_rng = 88172645463325252ull;
std::uint64_t rnd() { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17; return g_rng; }

template <typename T>
bool Compare(T a, ComparisonID c, T b) {
  switch (c) {
    case ComparisonID::kEqual: return a == b;
    case ComparisonID::kNotEqual: return a != b;
    case ComparisonID::kLess: return a < b;
    case ComparisonID::kLessOrEqual: return a <= b;
    case ComparisonID::kGreater: return a > b;
    default: return a >= b;
  }
}

std::uint32_t CodeAt(const CompressedAttribute &c, const std::vector<unsigned char> &codes, std::size_t i) {
  switch (c.code_width) {
    case 1: return codes[i];
    case 2: return reinterpret_cast<const std::uint16_t *>(codes.data())[i];
    default: return reinterpret_cast<const std::uint32_t *>(codes.data())[i];
  }
}

bool CodeMatches(const PredicateTransformResult &r, std::uint32_t code) {
  switch (r.type) {
    case PredicateTransformResult::kAll: return true;
    case PredicateTransformResult::kNone: return false;
    case PredicateTransformResult::kRangeComparison: return code >= r.first_literal && code < r.second_literal;
    default:
      switch (r.comp) {
        case QSX_CODE_EQ: return code == r.first_literal;
        case QSX_CODE_NE: return code != r.first_literal;
        case QSX_CODE_LT: return code < r.first_literal;
        default: return code >= r.first_literal;
      }
  }
}

template <typename T>
void CheckColumn(const char *name, TypeID type, const std::vector<T> &values, CompressedAttribute::Kind want_kind, int want_width,
                 const std::vector<T> &literals, TypedLiteral (*make)(T)) {
  CompressedAttribute c;
  std::vector<unsigned char> codes;
  CompressValues(type, values.data(), static_cast<std::int64_t>(values.size()), &c, &codes);
  if (c.kind != want_kind || (want_kind != CompressedAttribute::kUncompressed && c.code_width != want_width)) {
    std::fprintf(stderr, "%s: kind %d width %d, expected %d / %d\n", name, c.kind, c.code_width, want_kind, want_width);
    ++g_failures;
    return;
  }
  if (c.kind == CompressedAttribute::kUncompressed) return;
  const T *dict = reinterpret_cast<const T *>(c.dictionary_host.data());
  for (std::size_t i = 0; i < values.size(); ++i) {   // decode round trip
    const std::uint32_t code = CodeAt(c, codes, i);
    const T decoded = c.kind == CompressedAttribute::kDictionary ? dict[code] : static_cast<T>(code);
    if (decoded != values[i]) { std::fprintf(stderr, "%s: decode mismatch at %zu\n", name, i); ++g_failures; return; }
  }
  for (const T lit : literals) {
    for (const ComparisonID comp : {ComparisonID::kEqual, ComparisonID::kNotEqual, ComparisonID::kLess, ComparisonID::kLessOrEqual,
                                    ComparisonID::kGreater, ComparisonID::kGreaterOrEqual}) {
      const PredicateTransformResult r = TransformPredicateOnCompressedAttribute(c, type, comp, make(lit));
      for (std::size_t i = 0; i < values.size(); ++i) {
        if (CodeMatches(r, CodeAt(c, codes, i)) != Compare(values[i], comp, lit)) {
          std::fprintf(stderr, "%s: comparison %d with literal %g differs at row %zu\n", name, static_cast<int>(comp),
                       static_cast<double>(lit), i);
          ++g_failures;
          return;
        }
      }
    }
  }
}
}  // namespace

int main() {
  // ---- compressed attributes -----------------------------------------------------------------------------------------
  {
    std::vector<std::int32_t> small, dict, wide;
    std::vector<std::int64_t> longs;
    std::vector<double> disc;
    std::vector<float> halves;
    for (int i = 0; i < 5000; ++i) {
      small.push_back(static_cast<std::int32_t>(rnd() % 200));
      const std::int32_t choices[] = {-7, -1, 3, 900, 1000000};
      dict.push_back(choices[rnd() % 5]);
      wide.push_back(static_cast<std::int32_t>(rnd()));
      longs.push_back(static_cast<std::int64_t>(rnd() % 60000));
      disc.push_back(static_cast<double>(rnd() % 11) / 100.0);
      halves.push_back(static_cast<float>(rnd() % 700) * 0.5f);
    }
    CheckColumn<std::int32_t>("int truncated to 1 byte", kInt, small, CompressedAttribute::kTruncated, 1,
                              {-1, 0, 1, 57, 199, 200, 255, 256, 100000}, TypedLiteral::Int);
    CheckColumn<std::int32_t>("int dictionary", kInt, dict, CompressedAttribute::kDictionary, 1,
                              {-8, -7, -2, 3, 4, 900, 1000000, 1000001}, TypedLiteral::Int);
    CheckColumn<std::int32_t>("int incompressible", kInt, wide, CompressedAttribute::kUncompressed, 4, {}, TypedLiteral::Int);
    CheckColumn<std::int64_t>("long truncated to 2 bytes", kLong, longs, CompressedAttribute::kTruncated, 2,
                              {-5, 0, 30000, 59999, 65535, 65536, 1ll << 40}, TypedLiteral::Long);
    CheckColumn<double>("double dictionary", kDouble, disc, CompressedAttribute::kDictionary, 1,
                        {-0.5, 0.0, 0.045, 0.05, 0.1, 0.2}, TypedLiteral::Double);
    CheckColumn<float>("float dictionary, 2-byte codes", kFloat, halves, CompressedAttribute::kDictionary, 2,
                       {-1.0f, 0.0f, 0.25f, 100.5f, 349.5f, 1000.0f}, TypedLiteral::Float);
  }
  // ---- partition scheme bookkeeping -----------------------------------------------------------------------------------
  {
    CatalogRelation rel(1, "r");
    rel.addAttribute("k", Type::Long());
    EXPECT_TRUE(!rel.hasPartitionScheme());
    EXPECT_EQ(rel.getNumPartitions(), static_cast<std::size_t>(1));
    rel.setPartitionScheme(4, 0);
    for (block_id b = 1; b <= 10; ++b) rel.addBlockToPartition(b, static_cast<partition_id>(b % 4));
    EXPECT_EQ(rel.getBlocksSnapshot().size(), static_cast<std::size_t>(10));
    EXPECT_EQ(rel.getBlocksInPartition(1).size(), static_cast<std::size_t>(3));   // blocks 1, 5, 9
    EXPECT_EQ(rel.getBlocksInPartition(0).size(), static_cast<std::size_t>(2));   // blocks 4, 8
    EXPECT_EQ(rel.getBlocksInPartition(1)[1], static_cast<block_id>(5));
  }
  // ---- work-order container: FIFO per operator -----------------------------------------------------------------------
  {
    struct Tagged : WorkOrder {
      int tag;
      explicit Tagged(int t) : WorkOrder(0), tag(t) {}
      void execute() override {}
    };
    WorkOrdersContainer container(2);
    for (int t = 0; t < 5; ++t) container.addNormalWorkOrder(new Tagged(t), static_cast<std::size_t>(t % 2));
    EXPECT_EQ(container.getNumNormalWorkOrders(0), static_cast<std::size_t>(3));
    int expected = 0;
    while (container.hasNormalWorkOrder(0)) {
      std::unique_ptr<WorkOrder> wo(container.getNormalWorkOrder(0));
      EXPECT_EQ(static_cast<Tagged *>(wo.get())->tag, expected);
      expected += 2;
    }
    EXPECT_TRUE(container.hasNormalWorkOrder(1));
  }
  return finish("host_logic_test");
}
