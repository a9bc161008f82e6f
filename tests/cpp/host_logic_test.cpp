// Host logic of the operator layer that needs no device: CompressedBlockBuilder's per-attribute choice, the rewrite of
// comparisons into code comparisons (CompressedStoreUtil.cpp:51-140, 425-616) checked row by row against the comparison on
// the decoded values, partition-scheme bookkeeping, work-order container order.  Runs in the CPU test suite.
#include <algorithm>
#include <cstring>

#include "block_image_util.hpp"
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

// Reference block images (host memory mode: no device needed): header parsing, max_tuples, stripe and null bitmap offsets.
void CheckBlockImages() {
  UseHostMemoryForBlocks(true);
  {
    CatalogRelation rel(40, "plain");
    rel.addAttribute("k", Type::Int());
    rel.addAttribute("p", Type::Double());
    rel.addAttribute("c", Type::Char(7));
    const std::int64_t n = 1000;
    std::vector<std::int32_t> k(n);
    std::vector<double> pcol(n);
    std::vector<char> c(n * 7);
    for (std::int64_t i = 0; i < n; ++i) { k[i] = static_cast<std::int32_t>(i * 3); pcol[i] = i * 0.5; std::snprintf(&c[i * 7], 7, "r%lld", static_cast<long long>(i)); }
    std::int64_t max_tuples = 0;
    std::vector<unsigned char> image = block_image::Build(rel, {k.data(), pcol.data(), c.data()}, {{}, {}, {}}, n, 2u << 20, 1, &max_tuples);
    const ReferenceBlockLayout layout = ParseReferenceBlockImage(rel, image.data(), 4096, image.size());
    EXPECT_EQ(layout.num_tuples, n);
    EXPECT_EQ(layout.max_tuples, max_tuples);
    EXPECT_EQ(layout.sort_attribute, 1);
    // 2 MB - 4 - header - 8 bytes over 19-byte rows
    EXPECT_EQ(layout.max_tuples, static_cast<std::int64_t>((layout.tuple_store_size - 8) / 19));
    EXPECT_EQ(layout.stripe_offset[1] - layout.stripe_offset[0], static_cast<std::size_t>(max_tuples) * 4);
    EXPECT_EQ(layout.stripe_offset[2] - layout.stripe_offset[1], static_cast<std::size_t>(max_tuples) * 8);
    EXPECT_TRUE(layout.null_bitmap_offset[0] == static_cast<std::size_t>(-1));
    StorageManager storage;
    const block_id id = storage.adoptBlockImage(&rel, image.data(), image.size());
    BlockReference blk = storage.getBlock(id);
    EXPECT_EQ(blk->numTuples(), n);
    EXPECT_EQ(blk->sortColumn(), 1);
    EXPECT_TRUE(blk->stripe(0) == image.data() + layout.stripe_offset[0]);      // in place: no copy
    std::vector<double> got(n);
    blk->copyAttributeToHost(1, got.data());
    EXPECT_TRUE(got == pcol);
  }
  {
    CatalogRelation rel(41, "nullable");
    rel.addAttribute("k", Type::Long());
    rel.addAttribute("q", Type::Int().getNullableVersion());
    rel.addAttribute("v", Type::Double().getNullableVersion());
    const std::int64_t n = 12345;
    std::vector<std::int64_t> k(n);
    std::vector<std::int32_t> q(n);
    std::vector<double> v(n);
    std::vector<bool> qn(n), vn(n);
    for (std::int64_t i = 0; i < n; ++i) { k[i] = i; q[i] = static_cast<std::int32_t>(i % 50); v[i] = i * 0.25; qn[i] = i % 17 == 0; vn[i] = i % 5 == 2; }
    std::int64_t max_tuples = 0;
    std::vector<unsigned char> image = block_image::Build(rel, {k.data(), q.data(), v.data()}, {{}, qn, vn}, n, 4u << 20, -1, &max_tuples);
    const ReferenceBlockLayout layout = ParseReferenceBlockImage(rel, image.data(), 8192, image.size());
    EXPECT_EQ(layout.max_tuples, max_tuples);
    EXPECT_EQ(layout.sort_attribute, kInvalidAttributeID);
    const std::size_t bitmap_bytes = static_cast<std::size_t>((max_tuples + 63) / 64 * 8);
    EXPECT_EQ(layout.null_bitmap_offset[1], layout.tuple_store_offset + 8);
    EXPECT_EQ(layout.null_bitmap_offset[2], layout.tuple_store_offset + 8 + bitmap_bytes);
    EXPECT_EQ(layout.stripe_offset[0], layout.tuple_store_offset + 8 + 2 * bitmap_bytes);
    // the two-step max_tuples of the reference: the bitmaps' rounding to whole words must still fit
    EXPECT_TRUE(8 + 2 * bitmap_bytes + static_cast<std::size_t>(max_tuples) * 20 <= layout.tuple_store_size);
    EXPECT_TRUE(8 + 2 * bitmap_bytes + static_cast<std::size_t>(max_tuples + 1) * 20 > layout.tuple_store_size - 16);
    StorageManager storage;
    BlockReference blk = storage.getBlock(storage.adoptBlockImage(&rel, image.data(), image.size()));
    std::vector<std::uint64_t> nulls(static_cast<std::size_t>((n + 63) / 64));
    blk->copyNullBitmapToHost(2, nulls.data());
    bool same = true;
    for (std::int64_t i = 0; i < n; ++i) same = same && (((nulls[i >> 6] >> (63 - (i & 63))) & 1u) != 0) == vn[i];
    EXPECT_TRUE(same);
    // malformed images are refused (StorageBlock.cpp:108-131), other tuple stores reported as unsupported
    auto refused = [&](std::vector<unsigned char> bad, std::size_t bytes, int want_status) {
      try {
        (void)ParseReferenceBlockImage(rel, bad.data(), std::min<std::size_t>(bytes, 8192), bytes);
      } catch (const ExecutionError &e) {
        return e.status() == want_status;
      }
      return false;
    };
    std::vector<unsigned char> bad = image;
    const std::int32_t negative = -5;
    std::memcpy(bad.data(), &negative, 4);
    EXPECT_TRUE(refused(bad, bad.size(), QSX_ERR_INVALID_ARGUMENT));
    EXPECT_TRUE(refused(image, 100, QSX_ERR_INVALID_ARGUMENT));                          // sub-block sizes exceed the block
    bad = image;
    const std::int32_t too_many = static_cast<std::int32_t>(max_tuples + 1);
    std::memcpy(bad.data() + layout.tuple_store_offset, &too_many, 4);
    EXPECT_TRUE(refused(bad, bad.size(), QSX_ERR_INVALID_ARGUMENT));
    std::vector<unsigned char> split_row = block_image::Header(2, 1000, -1, /*SPLIT_ROW_STORE=*/3);
    std::vector<unsigned char> other(4096, 0);
    const std::int32_t len = static_cast<std::int32_t>(split_row.size());
    std::memcpy(other.data(), &len, 4);
    std::memcpy(other.data() + 4, split_row.data(), split_row.size());
    EXPECT_TRUE(refused(other, other.size(), QSX_ERR_UNSUPPORTED));
    // a CompressedColumnStore image of the same relation: k truncated to 2 bytes and the sort column, q dictionary-coded in
    // one byte with a NULL code, v as values with a null bitmap (CompressedColumnStoreTupleStorageSubBlock.cpp:755-798)
    {
      for (std::int64_t i = 0; i < n; ++i) k[i] = i / 4;     // (fits 2 bytes, ascending)
      block_image::Coding truncated, dictionary, values;
      truncated.kind = block_image::Coding::kTruncated;
      truncated.code_width = 2;
      dictionary.kind = block_image::Coding::kDictionary;
      dictionary.code_width = 1;
      std::int64_t cmax = 0;
      const std::vector<unsigned char> cimage = block_image::BuildCompressed(rel, {k.data(), q.data(), v.data()}, {{}, qn, vn}, n, 2u << 20, 0,
                                                                             {truncated, dictionary, values}, &cmax);
      const ReferenceBlockLayout cl = ParseReferenceBlockImage(rel, cimage.data(), 8192, cimage.size());
      EXPECT_TRUE(cl.compressed);
      EXPECT_EQ(cl.num_tuples, n);
      EXPECT_EQ(cl.max_tuples, cmax);
      EXPECT_EQ(cl.sort_attribute, 0);
      EXPECT_EQ(cl.attribute_size[0], static_cast<std::size_t>(2));
      EXPECT_EQ(cl.attribute_size[1], static_cast<std::size_t>(1));
      EXPECT_EQ(cl.attribute_size[2], static_cast<std::size_t>(8));
      EXPECT_TRUE(cl.dictionary_offset[0] == static_cast<std::size_t>(-1) && cl.dictionary_offset[2] == static_cast<std::size_t>(-1));
      EXPECT_EQ(cl.dictionary_bytes[1], static_cast<std::size_t>(8 + 50 * 4));          // {num_codes, null_code} + 50 INT values
      std::uint32_t head[2];
      std::memcpy(head, cimage.data() + cl.dictionary_offset[1], 8);
      EXPECT_EQ(head[0], 50u);
      EXPECT_EQ(head[1], 50u);                                                             // NULL = the code num_codes
      EXPECT_EQ(cl.null_bitmap_bits, static_cast<std::size_t>(n));
      EXPECT_TRUE(cl.null_bitmap_offset[1] == static_cast<std::size_t>(-1));              // q's NULLs are codes, not bits
      EXPECT_EQ(cl.null_bitmap_offset[2], cl.dictionary_offset[1] + cl.dictionary_bytes[1]);
      EXPECT_EQ(cl.stripe_offset[0], cl.null_bitmap_offset[2] + static_cast<std::size_t>((n + 63) / 64 * 8));
      EXPECT_EQ(cl.stripe_offset[1], cl.stripe_offset[0] + static_cast<std::size_t>(cmax) * 2);
      EXPECT_EQ(cl.stripe_offset[2], cl.stripe_offset[1] + static_cast<std::size_t>(cmax));
      EXPECT_TRUE(cl.stripe_offset[2] + static_cast<std::size_t>(cmax) * 8 <= cimage.size());
      std::uint16_t key_at_1000 = 0;
      std::memcpy(&key_at_1000, cimage.data() + cl.stripe_offset[0] + 2000, 2);
      EXPECT_EQ(key_at_1000, 250);
      std::vector<unsigned char> cbad = cimage;
      const std::int32_t absurd = 1 << 30;
      std::memcpy(cbad.data() + cl.tuple_store_offset + 4, &absurd, 4);                    // info size beyond the block
      EXPECT_TRUE(refused(cbad, cbad.size(), QSX_ERR_INVALID_ARGUMENT));
    }
  }
  UseHostMemoryForBlocks(false);
}

int main() {
  CheckBlockImages();
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
  // ---- DATE and CHAR(n) attributes: dictionaries under the type's own order ---------------------------------------------
  // (TPC-H: l_shipdate / o_orderdate DATE, l_shipmode CHAR(10), l_shipinstruct CHAR(25) of the compressed column stores,
  // benchmarks/tpch/create.sql:73-121).  Every comparison on codes must select the rows the comparison on values selects.
  {
    const int n = 6000;
    std::vector<DateLit> dates;
    for (int i = 0; i < n; ++i) {
      DateLit d = DateLit::Create(1992 + static_cast<int>(rnd() % 7), static_cast<std::uint8_t>(1 + rnd() % 12), static_cast<std::uint8_t>(1 + rnd() % 28));
      d.unused[0] = static_cast<std::uint8_t>(rnd());   // garbage in the padding bytes: not part of the value
      d.unused[1] = static_cast<std::uint8_t>(rnd());
      dates.push_back(d);
    }
    CompressedAttribute c;
    std::vector<unsigned char> codes;
    CompressValues(kDate, dates.data(), n, &c, &codes);
    EXPECT_TRUE(c.kind == CompressedAttribute::kDictionary && c.code_width == 2 && c.value_width == 8);
    const DateLit *dict = reinterpret_cast<const DateLit *>(c.dictionary_host.data());
    for (int i = 0; i < n; ++i) EXPECT_TRUE(dict[CodeAt(c, codes, i)] == dates[i] && dict[CodeAt(c, codes, i)].unused[0] == 0);
    const int probes[][3] = {{1991, 12, 31}, {1992, 1, 1}, {1995, 3, 15}, {1998, 9, 2}, {1998, 12, 28}, {1999, 1, 1}, {-18017, 4, 13}, {99999, 12, 31}};
    for (const auto &p : probes) {
      const DateLit lit = DateLit::Create(p[0], static_cast<std::uint8_t>(p[1]), static_cast<std::uint8_t>(p[2]));
      for (const ComparisonID comp : {ComparisonID::kEqual, ComparisonID::kNotEqual, ComparisonID::kLess, ComparisonID::kLessOrEqual,
                                      ComparisonID::kGreater, ComparisonID::kGreaterOrEqual}) {
        const PredicateTransformResult r = TransformPredicateOnCompressedAttribute(c, kDate, comp, TypedLiteral::Date(p[0], p[1], p[2]));
        for (int i = 0; i < n; ++i) {
          const int order = dates[i] < lit ? -1 : (lit < dates[i] ? 1 : 0);
          if (CodeMatches(r, CodeAt(c, codes, i)) != Compare(order, comp, 0)) {
            std::fprintf(stderr, "date dictionary: comparison %d with %d-%d-%d differs at row %d\n", static_cast<int>(comp), p[0], p[1], p[2], i);
            ++g_failures;
            break;
          }
        }
      }
    }
    // CHAR(10): five words, one of them filling the field exactly, bytes behind a terminator are junk
    const char *words[] = {"BUILDING", "AUTOMOBILE", "MACHINERY", "HOUSEHOLD", "FURNITURE"};
    std::vector<char> column(static_cast<std::size_t>(n) * 10, 0);
    for (int i = 0; i < n; ++i) {
      const char *w = words[rnd() % 5];
      std::strncpy(&column[static_cast<std::size_t>(i) * 10], w, 10);
      if (std::strlen(w) < 9 && rnd() % 4 == 0) column[static_cast<std::size_t>(i) * 10 + 9] = 'x';
    }
    CompressedAttribute cc;
    CompressValues(kChar, column.data(), n, &cc, &codes, 10);
    EXPECT_TRUE(cc.kind == CompressedAttribute::kDictionary && cc.code_width == 1 && cc.num_codes == 5 && cc.value_width == 10);
    for (const char *lit : {"BUILDING", "BUILD", "BUILDINGS", "AUTOMOBILE", "AUTOMOBILES", "", "ZZZ", "HOUSEHOLD", "MACHINERZ"}) {
      for (const ComparisonID comp : {ComparisonID::kEqual, ComparisonID::kNotEqual, ComparisonID::kLess, ComparisonID::kLessOrEqual,
                                      ComparisonID::kGreater, ComparisonID::kGreaterOrEqual}) {
        const PredicateTransformResult r = TransformPredicateOnCompressedAttribute(cc, kChar, comp, TypedLiteral::Char(lit));
        for (int i = 0; i < n; ++i) {
          // the field as a C string (ends at its terminator or at 10 bytes) against the literal
          const std::string value(&column[static_cast<std::size_t>(i) * 10], strnlen(&column[static_cast<std::size_t>(i) * 10], 10));
          const int order = value.compare(lit);
          if (CodeMatches(r, CodeAt(cc, codes, i)) != Compare(order < 0 ? -1 : (order > 0 ? 1 : 0), comp, 0)) {
            std::fprintf(stderr, "char dictionary: comparison %d with '%s' differs at row %d ('%s')\n", static_cast<int>(comp), lit, i, value.c_str());
            ++g_failures;
            break;
          }
        }
      }
    }
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
