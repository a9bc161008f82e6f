// storage.cpp — catalog, storage blocks, attribute compression, reference block images (see quickstep_gpu.hpp; what the files share: quickstep_gpu_internal.hpp)
#include "quickstep_gpu_internal.hpp"

namespace quickstep {

// ---------------------------------------------------------------------------
// catalog + storage
// ---------------------------------------------------------------------------
attribute_id CatalogRelation::addAttribute(const std::string &name, Type type) {
  names_.push_back(name);
  types_.push_back(type);
  return static_cast<attribute_id>(types_.size() - 1);
}
attribute_id CatalogRelation::getAttributeByName(const std::string &name) const {
  for (std::size_t i = 0; i < names_.size(); ++i) {
    if (names_[i] == name) return static_cast<attribute_id>(i);
  }
  return kInvalidAttributeID;
}
void CatalogRelation::addBlock(block_id b) {
  std::lock_guard<std::mutex> lock(mutex_);
  blocks_.push_back(b);
  if (num_partitions_ > 0) partition_blocks_.at(0).push_back(b);
}
std::vector<block_id> CatalogRelation::getBlocksSnapshot() const {
  std::lock_guard<std::mutex> lock(mutex_);
  return blocks_;
}
void CatalogRelation::setPartitionScheme(std::size_t num_partitions, attribute_id partition_attribute) {
  std::lock_guard<std::mutex> lock(mutex_);
  num_partitions_ = num_partitions;
  partition_attribute_ = partition_attribute;
  partition_blocks_.assign(num_partitions, {});
}
void CatalogRelation::addBlockToPartition(block_id b, partition_id part) {
  std::lock_guard<std::mutex> lock(mutex_);
  blocks_.push_back(b);
  if (num_partitions_ > 0) partition_blocks_.at(part).push_back(b);
}
std::vector<block_id> CatalogRelation::getBlocksInPartition(partition_id part) const {
  std::lock_guard<std::mutex> lock(mutex_);
  if (num_partitions_ == 0) return part == 0 ? blocks_ : std::vector<block_id>();
  return partition_blocks_.at(part);
}

namespace {
bool g_host_memory = false;  // CPU plumbing mode (BASELINE config 1): blocks live in host memory
}
void UseHostMemoryForBlocks(bool on) { g_host_memory = on; }

StorageBlock::StorageBlock(const CatalogRelation &relation, std::int64_t capacity, std::int64_t first_row, bool one_allocation)
    : relation_(relation), capacity_(capacity), num_tuples_(0), first_row_(first_row) {
  if (one_allocation && !g_host_memory && std::getenv("QSX_HOST_BLOCK_SLAB_OFF") == nullptr) {
    // an output block: all stripes and null bitmaps in one allocation (a device allocation of a few MB costs ~170 us
    // whatever its size, and a block has one per attribute otherwise)
    auto round_up = [](std::size_t v) { return (v + 255) / 256 * 256; };
    std::size_t total = 0, null_bytes = 0;
    for (std::size_t a = 0; a < relation.size(); ++a) {
      const Type &t = relation.getAttributeType(static_cast<attribute_id>(a));
      total += round_up(static_cast<std::size_t>(capacity) * t.width + 8);
      if (t.nullable) null_bytes += round_up(static_cast<std::size_t>((capacity + 63) / 64) * 8 + 8);
    }
    slab_bytes_ = total + null_bytes;
    slab_ = TakePooled(slab_bytes_, &slab_granted_);
    char *at = static_cast<char *>(slab_);
    char *nulls_at = at + total;
    if (null_bytes != 0) CheckStatus(qsx_memset_device(nulls_at, 0, null_bytes, CurrentStream()), "qsx_memset_device(null bitmaps)");
    for (std::size_t a = 0; a < relation.size(); ++a) {
      const Type &t = relation.getAttributeType(static_cast<attribute_id>(a));
      stripes_.push_back(at);
      at += round_up(static_cast<std::size_t>(capacity) * t.width + 8);
      void *nulls = nullptr;
      if (t.nullable) {
        nulls = nulls_at;
        nulls_at += round_up(static_cast<std::size_t>((capacity + 63) / 64) * 8 + 8);
      }
      null_bitmaps_.push_back(nulls);
    }
    return;
  }
  for (std::size_t a = 0; a < relation.size(); ++a) {
    void *p = nullptr;
    const std::size_t bytes = static_cast<std::size_t>(capacity) * relation.getAttributeType(static_cast<attribute_id>(a)).width;
    if (g_host_memory) {
      p = std::malloc(bytes ? bytes : 8);
    } else {
      CheckStatus(qsx_device_alloc(bytes ? bytes : 8, &p), "qsx_device_alloc(stripe)");
    }
    stripes_.push_back(p);
    void *nulls = nullptr;
    if (relation.getAttributeType(static_cast<attribute_id>(a)).nullable) {
      const std::size_t nbytes = static_cast<std::size_t>((capacity + 63) / 64) * 8 + 8;
      if (g_host_memory) {
        nulls = std::calloc(nbytes, 1);
      } else {
        CheckStatus(qsx_device_alloc(nbytes, &nulls), "qsx_device_alloc(null bitmap)");
        CheckStatus(qsx_memset_device(nulls, 0, nbytes, CurrentStream()), "qsx_memset_device(null bitmap)");
      }
    }
    null_bitmaps_.push_back(nulls);
  }
}
StorageBlock::StorageBlock(std::shared_ptr<StorageBlock> parent, std::int64_t first_tuple, std::int64_t num_tuples)
    : relation_(parent->relation_), capacity_(num_tuples), num_tuples_(num_tuples), first_row_(0), view_parent_(std::move(parent)) {
  for (std::size_t a = 0; a < relation_.size(); ++a) {
    const Type &t = relation_.getAttributeType(static_cast<attribute_id>(a));
    stripes_.push_back(static_cast<char *>(view_parent_->stripe(static_cast<attribute_id>(a))) + first_tuple * t.width);
    void *nulls = nullptr;
    if (t.nullable) {
      const std::size_t nbytes = static_cast<std::size_t>((num_tuples + 63) / 64) * 8 + 8;
      if (g_host_memory) {
        nulls = std::calloc(nbytes, 1);
      } else {
        CheckStatus(qsx_device_alloc(nbytes, &nulls), "qsx_device_alloc(null bitmap)");
        CheckStatus(qsx_memset_device(nulls, 0, nbytes, CurrentStream()), "qsx_memset_device(null bitmap)");
      }
    }
    null_bitmaps_.push_back(nulls);
  }
}
StorageBlock::StorageBlock(const CatalogRelation &relation, std::int64_t num_tuples, const std::vector<void *> &stripes,
                           const std::vector<void *> &null_bitmaps)
    : relation_(relation), capacity_(num_tuples), num_tuples_(num_tuples), first_row_(0), external_memory_(true),
      stripes_(stripes), null_bitmaps_(null_bitmaps) {}

StorageBlock::~StorageBlock() {
  if (external_memory_) {
    // pointers into the adopted image are the caller's; what the block allocated afterwards (a decoded stripe, the null
    // bitmap made from a dictionary's NULL code) is freed below like any other block's
    auto foreign = [&](const void *p) {
      if (external_bytes_ == 0) return true;
      const char *c = static_cast<const char *>(p);
      return c >= external_base_ && c < external_base_ + external_bytes_;
    };
    for (void *&p : stripes_) if (p != nullptr && foreign(p)) p = nullptr;
    for (void *&p : null_bitmaps_) if (p != nullptr && foreign(p)) p = nullptr;
    for (CompressedAttribute &c : compressed_) {
      if (c.codes != nullptr && foreign(c.codes)) c.codes = nullptr;
      if (c.dictionary != nullptr && foreign(c.dictionary)) c.dictionary = nullptr;
    }
  }
  // A block that dies while an exception unwinds a work order may still be the target of kernels that work order has
  // queued: they must finish before its memory goes back to a pool another Worker takes from (the regular path has
  // synchronised its stream before the last reference goes).
  if (std::uncaught_exceptions() > 0 && !g_host_memory) (void)qsx_stream_synchronize(CurrentStream());
  if (view_parent_ != nullptr) {
    for (void *&p : stripes_) p = nullptr;   // the parent's
  }
  if (slab_ != nullptr) {
    // (a stripe outside the slab was materialised later, stripe(): freed on its own below)
    auto in_slab = [&](void *p) { return p >= slab_ && p < static_cast<char *>(slab_) + slab_bytes_; };
    for (void *&p : stripes_) if (in_slab(p)) p = nullptr;
    for (void *&p : null_bitmaps_) if (in_slab(p)) p = nullptr;
    GivePooled(slab_, slab_granted_);
  }
  for (void *p : stripes_) {
    if (g_host_memory) std::free(p); else qsx_device_free(p);
  }
  for (void *p : null_bitmaps_) {
    if (p == nullptr) continue;
    if (g_host_memory) std::free(p); else qsx_device_free(p);
  }
  for (CompressedAttribute &c : compressed_) {
    qsx_device_free(c.codes);
    qsx_device_free(c.dictionary);
  }
}
void StorageBlock::copyNullBitmapToHost(attribute_id a, std::uint64_t *dst) const {
  const std::size_t bytes = static_cast<std::size_t>((num_tuples_ + 63) / 64) * 8;
  if (null_bitmaps_.at(a) == nullptr) {
    std::memset(dst, 0, bytes);
  } else if (g_host_memory) {
    std::memcpy(dst, null_bitmaps_.at(a), bytes);
  } else {
    CheckStatus(qsx_copy_to_host(dst, null_bitmaps_.at(a), bytes, CurrentStream()), "qsx_copy_to_host(null bitmap)");
  }
}
void StorageBlock::copyAttributeToHost(attribute_id a, void *dst) const {
  const std::size_t bytes = static_cast<std::size_t>(num_tuples_) * relation_.getAttributeType(a).width;
  if (g_host_memory) {
    std::memcpy(dst, stripes_.at(a), bytes);
  } else {
    CheckStatus(qsx_copy_to_host(dst, stripe(a), bytes, CurrentStream()), "qsx_copy_to_host");
  }
}

void StorageBlock::adoptCompressedAttribute(attribute_id a, CompressedAttribute attribute) {
  if (compressed_.empty()) compressed_.resize(relation_.size());
  compressed_.at(a) = std::move(attribute);
  stripes_.at(a) = nullptr;   // decoded on first use (stripe())
}

void *StorageBlock::stripe(attribute_id a) const {
  if (compressed_.empty() || compressed_.at(a).kind == CompressedAttribute::kUncompressed) return stripes_.at(a);
  std::lock_guard<std::mutex> lock(decode_mutex_);
  if (stripes_.at(a) == nullptr) {
    const CompressedAttribute &c = compressed_.at(a);
    const int width = relation_.getAttributeType(a).width;
    void *values = nullptr;
    CheckStatus(qsx_device_alloc(static_cast<std::size_t>(capacity_ ? capacity_ : 1) * width, &values), "qsx_device_alloc(decoded stripe)");
    CheckStatus(qsx_decode_codes(c.code_width, c.codes, num_tuples_, c.dictionary, width, values, CurrentStream()), "qsx_decode_codes");
    CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");   // other workers' streams read it next
    stripes_.at(a) = values;
  }
  return stripes_.at(a);
}

namespace {
// CompressedBlockBuilder's per-attribute decision for fixed-width numeric attributes without NULLs
// (storage/CompressedBlockBuilder.cpp:508-566 truncated width, :590-650 truncation vs dictionary;
// compression/CompressionDictionaryBuilder.{hpp:87-110,cpp:128-134} code length and dictionary size).
template <typename T>
void BuildCompressedAttribute(TypeID type, const T *values, std::int64_t n, CompressedAttribute *out,
                              std::vector<unsigned char> *codes_host) {
  std::vector<T> dict(values, values + n);
  std::sort(dict.begin(), dict.end());
  dict.erase(std::unique(dict.begin(), dict.end()), dict.end());
  unsigned code_bits = 0;
  for (std::size_t num_values = 1; num_values <= dict.size(); ++num_values) {
    if (code_bits == 0 || num_values == (1ull << code_bits) + 1) ++code_bits;
  }
  const std::size_t dict_code_bytes = code_bits < 9 ? 1 : (code_bits < 17 ? 2 : 4);
  const std::size_t dictionary_bytes = 2 * sizeof(std::uint32_t) + dict.size() * sizeof(T) + static_cast<std::size_t>(n) * dict_code_bytes;
  std::size_t truncated_width = sizeof(T);
  if ((type == kInt || type == kLong) && n > 0) {
    bool negative = false;
    std::int64_t mx = 0;
    for (std::int64_t i = 0; i < n; ++i) {
      const std::int64_t v = static_cast<std::int64_t>(values[i]);
      negative = negative || v < 0;
      mx = std::max(mx, v);
    }
    if (!negative && !(type == kLong && mx == 0xFFFFFFFFll)) {
      unsigned needed_bits = 0;
      while (needed_bits < 64 && (static_cast<std::uint64_t>(mx) >> needed_bits) != 0) ++needed_bits;
      if (needed_bits < 9) truncated_width = 1;
      else if (needed_bits < 17) truncated_width = 2;
      else if (needed_bits < 33) truncated_width = 4;
    }
  }
  auto put_code = [&](std::int64_t i, std::uint64_t code, std::size_t width) {
    switch (width) {
      case 1: (*codes_host)[i] = static_cast<unsigned char>(code); break;
      case 2: reinterpret_cast<std::uint16_t *>(codes_host->data())[i] = static_cast<std::uint16_t>(code); break;
      default: reinterpret_cast<std::uint32_t *>(codes_host->data())[i] = static_cast<std::uint32_t>(code); break;
    }
  };
  if (static_cast<std::size_t>(n) * truncated_width < dictionary_bytes) {
    if (truncated_width == sizeof(T)) {
      out->kind = CompressedAttribute::kUncompressed;
      return;
    }
    out->kind = CompressedAttribute::kTruncated;
    out->code_width = static_cast<int>(truncated_width);
    codes_host->assign(static_cast<std::size_t>(n) * truncated_width, 0);
    for (std::int64_t i = 0; i < n; ++i) put_code(i, static_cast<std::uint64_t>(static_cast<std::int64_t>(values[i])), truncated_width);
    return;
  }
  out->kind = CompressedAttribute::kDictionary;
  out->code_width = static_cast<int>(dict_code_bytes);
  out->num_codes = static_cast<std::uint32_t>(dict.size());
  out->dictionary_host.assign(reinterpret_cast<const unsigned char *>(dict.data()),
                              reinterpret_cast<const unsigned char *>(dict.data() + dict.size()));
  codes_host->assign(static_cast<std::size_t>(n) * dict_code_bytes, 0);
  for (std::int64_t i = 0; i < n; ++i) {
    put_code(i, static_cast<std::uint64_t>(std::lower_bound(dict.begin(), dict.end(), values[i]) - dict.begin()), dict_code_bytes);
  }
}

template <typename T>
PredicateTransformResult TransformT(const CompressedAttribute &attr, ComparisonID comparison, T lit) {
  PredicateTransformResult r;   // kNone
  auto basic = [&](qsx_code_cmp_t comp, std::uint32_t code) {
    r.type = PredicateTransformResult::kBasicComparison;
    r.comp = comp;
    r.first_literal = code;
  };
  constexpr std::uint32_t kMax = 0xFFFFFFFFu;
  std::pair<std::uint32_t, std::uint32_t> range(0, 0);
  if (attr.kind == CompressedAttribute::kDictionary) {
    const T *dict = reinterpret_cast<const T *>(attr.dictionary_host.data());
    const T *end = dict + attr.num_codes;
    const std::uint32_t lower = static_cast<std::uint32_t>(std::lower_bound(dict, end, lit) - dict);
    const std::uint32_t upper = static_cast<std::uint32_t>(std::upper_bound(dict, end, lit) - dict);
    if (comparison == ComparisonID::kEqual) {           // TransformEqualPredicateOnCompressedAttribute (:425-470)
      if (lower != upper) basic(QSX_CODE_EQ, lower);
      return r;
    }
    if (comparison == ComparisonID::kNotEqual) {        // TransformNotEqualPredicate... (:472-535), no null code
      if (lower == upper) r.type = PredicateTransformResult::kAll;
      else basic(QSX_CODE_NE, lower);
      return r;
    }
    switch (comparison) {                               // getLimitCodesForComparisonTyped (CompressionDictionary.cpp:276-305)
      case ComparisonID::kLess: range = {0, lower}; break;
      case ComparisonID::kLessOrEqual: range = {0, upper}; break;
      case ComparisonID::kGreater: range = {upper, attr.num_codes}; break;
      default: range = {lower, attr.num_codes}; break;
    }
    if (range.first >= range.second) return r;
    if (range.second == attr.num_codes) range.second = kMax;
  } else {
    // truncated attribute (:144-236 TruncationHelper, :266-420 always-true / always-false)
    const std::int64_t max_truncated = attr.code_width == 4 ? 0xFFFFFFFFll : (1ll << (8 * attr.code_width)) - 1;
    const double as_double = static_cast<double>(lit);
    const bool long_exact = std::is_integral<T>::value || as_double == static_cast<double>(static_cast<std::int64_t>(as_double));
    const std::int64_t as_long = static_cast<std::int64_t>(lit);
    const bool in_range = as_long >= 0 && as_long <= max_truncated;
    if (comparison == ComparisonID::kEqual) {
      if (long_exact && in_range) basic(QSX_CODE_EQ, static_cast<std::uint32_t>(as_long));
      return r;
    }
    if (comparison == ComparisonID::kNotEqual) {
      if (!long_exact || !in_range) r.type = PredicateTransformResult::kAll;
      else basic(QSX_CODE_NE, static_cast<std::uint32_t>(as_long));
      return r;
    }
    const bool lower_side = comparison == ComparisonID::kLess || comparison == ComparisonID::kGreaterOrEqual;
    const std::int64_t eff = long_exact ? as_long
                                        : static_cast<std::int64_t>(lower_side ? std::ceil(as_double) : std::floor(as_double));
    bool always_true = false, always_false = false;
    switch (comparison) {
      case ComparisonID::kLess: always_true = eff > max_truncated; always_false = eff <= 0; break;
      case ComparisonID::kLessOrEqual: always_true = eff >= max_truncated; always_false = eff < 0; break;
      case ComparisonID::kGreater: always_true = eff < 0; always_false = eff >= max_truncated; break;
      default: always_true = eff <= 0; always_false = eff > max_truncated; break;
    }
    if (always_true) { r.type = PredicateTransformResult::kAll; return r; }
    if (always_false) return r;
    switch (comparison) {
      case ComparisonID::kLess: range = {0, static_cast<std::uint32_t>(eff)}; break;
      case ComparisonID::kLessOrEqual: range = {0, static_cast<std::uint32_t>(eff + 1)}; break;
      case ComparisonID::kGreater: range = {static_cast<std::uint32_t>(eff + 1), kMax}; break;
      default: range = {static_cast<std::uint32_t>(eff), kMax}; break;
    }
  }
  if (range.first == 0) {                                // :590-612
    if (range.second == kMax) r.type = PredicateTransformResult::kAll;
    else basic(QSX_CODE_LT, range.second);
  } else if (range.second == kMax) {
    basic(QSX_CODE_GE, range.first);
  } else {
    r.type = PredicateTransformResult::kRangeComparison;
    r.comp = QSX_CODE_RANGE;
    r.first_literal = range.first;
    r.second_literal = range.second;
  }
  return r;
}

// ---- DATE and CHAR(n) attributes: dictionaries of fixed-width byte values under the type's own order -------------------
// (CompressionDictionaryBuilder keeps its values in a set ordered by the type's less comparison,
// compression/CompressionDictionaryBuilder.cpp:40-90; there is no truncation for these types.)
// strcmpHelper of the reference (types/operations/comparisons/AsciiStringComparators.hpp:218-251)
int CompareAsciiStrings(const char *left, std::size_t left_length, const char *right, std::size_t right_length) {
  if (right_length > left_length) {
    const int res = std::strncmp(left, right, left_length);
    if (res) return res;
    return strnlen(right, right_length) > left_length ? -1 : res;
  } else if (left_length > right_length) {
    const int res = std::strncmp(left, right, right_length);
    if (res) return res;
    return strnlen(left, left_length) > right_length ? 1 : res;
  }
  return std::strncmp(left, right, left_length);
}
// <0, 0, >0: value (value_width bytes) against the literal
int CompareBytesWithLiteral(TypeID type, const unsigned char *value, int value_width, const void *literal, std::size_t literal_length) {
  if (type == kDate) {
    DateLit a, b;
    std::memcpy(&a, value, 8);
    std::memcpy(&b, literal, 8);
    return a < b ? -1 : (b < a ? 1 : 0);
  }
  return CompareAsciiStrings(reinterpret_cast<const char *>(value), static_cast<std::size_t>(value_width),
                             static_cast<const char *>(literal), literal_length);
}

void BuildByteDictionary(TypeID type, int width, const unsigned char *values, std::int64_t n, CompressedAttribute *out,
                         std::vector<unsigned char> *codes_host) {
  auto less = [&](const unsigned char *a, const unsigned char *b) { return CompareBytesWithLiteral(type, a, width, b, width) < 0; };
  std::vector<const unsigned char *> dict(static_cast<std::size_t>(n));
  for (std::int64_t i = 0; i < n; ++i) dict[static_cast<std::size_t>(i)] = values + i * width;
  std::sort(dict.begin(), dict.end(), less);
  dict.erase(std::unique(dict.begin(), dict.end(), [&](const unsigned char *a, const unsigned char *b) { return !less(a, b) && !less(b, a); }),
             dict.end());
  unsigned code_bits = 0;
  for (std::size_t num_values = 1; num_values <= dict.size(); ++num_values) {
    if (code_bits == 0 || num_values == (1ull << code_bits) + 1) ++code_bits;
  }
  const std::size_t code_bytes = code_bits < 9 ? 1 : (code_bits < 17 ? 2 : 4);
  const std::size_t dictionary_bytes = 2 * sizeof(std::uint32_t) + dict.size() * width + static_cast<std::size_t>(n) * code_bytes;
  if (static_cast<std::size_t>(n) * width < dictionary_bytes) {   // CompressedBlockBuilder.cpp:590-650: compress only if it is smaller
    out->kind = CompressedAttribute::kUncompressed;
    return;
  }
  out->kind = CompressedAttribute::kDictionary;
  out->code_width = static_cast<int>(code_bytes);
  out->num_codes = static_cast<std::uint32_t>(dict.size());
  out->value_width = width;
  out->dictionary_host.assign(dict.size() * width, 0);
  for (std::size_t e = 0; e < dict.size(); ++e) {
    // a DATE entry keeps year, month, day only; a CHAR entry ends at its terminator (what follows is not part of the value)
    const std::size_t keep = type == kDate ? 6 : strnlen(reinterpret_cast<const char *>(dict[e]), static_cast<std::size_t>(width));
    std::memcpy(out->dictionary_host.data() + e * width, dict[e], keep);
  }
  codes_host->assign(static_cast<std::size_t>(n) * code_bytes, 0);
  for (std::int64_t i = 0; i < n; ++i) {
    const std::uint32_t code = static_cast<std::uint32_t>(std::lower_bound(dict.begin(), dict.end(), values + i * width, less) - dict.begin());
    switch (code_bytes) {
      case 1: (*codes_host)[static_cast<std::size_t>(i)] = static_cast<unsigned char>(code); break;
      case 2: reinterpret_cast<std::uint16_t *>(codes_host->data())[i] = static_cast<std::uint16_t>(code); break;
      default: reinterpret_cast<std::uint32_t *>(codes_host->data())[i] = code; break;
    }
  }
}

// The dictionary branch of TransformT for byte dictionaries (same rules: CompressedStoreUtil.cpp:425-616,
// CompressionDictionary.cpp:276-305).
PredicateTransformResult TransformBytes(const CompressedAttribute &attr, TypeID type, ComparisonID comparison, const void *literal,
                                        std::size_t literal_length) {
  PredicateTransformResult r;
  const int width = attr.value_width;
  std::uint32_t lower = 0, upper = attr.num_codes;   // first entry >= literal, first entry > literal
  {
    std::uint32_t lo = 0, hi = attr.num_codes;
    while (lo < hi) {
      const std::uint32_t mid = lo + (hi - lo) / 2;
      if (CompareBytesWithLiteral(type, attr.dictionary_host.data() + static_cast<std::size_t>(mid) * width, width, literal, literal_length) < 0) lo = mid + 1;
      else hi = mid;
    }
    lower = lo;
    hi = attr.num_codes;
    while (lo < hi) {
      const std::uint32_t mid = lo + (hi - lo) / 2;
      if (CompareBytesWithLiteral(type, attr.dictionary_host.data() + static_cast<std::size_t>(mid) * width, width, literal, literal_length) <= 0) lo = mid + 1;
      else hi = mid;
    }
    upper = lo;
  }
  auto basic = [&](qsx_code_cmp_t comp, std::uint32_t code) {
    r.type = PredicateTransformResult::kBasicComparison;
    r.comp = comp;
    r.first_literal = code;
  };
  constexpr std::uint32_t kMax = 0xFFFFFFFFu;
  if (comparison == ComparisonID::kEqual) {
    if (lower != upper) basic(QSX_CODE_EQ, lower);
    return r;
  }
  if (comparison == ComparisonID::kNotEqual) {
    if (lower == upper) r.type = PredicateTransformResult::kAll;
    else basic(QSX_CODE_NE, lower);
    return r;
  }
  std::pair<std::uint32_t, std::uint32_t> range(0, 0);
  switch (comparison) {
    case ComparisonID::kLess: range = {0, lower}; break;
    case ComparisonID::kLessOrEqual: range = {0, upper}; break;
    case ComparisonID::kGreater: range = {upper, attr.num_codes}; break;
    default: range = {lower, attr.num_codes}; break;
  }
  if (range.first >= range.second) return r;
  if (range.second == attr.num_codes) range.second = kMax;
  if (range.first == 0) {
    if (range.second == kMax) r.type = PredicateTransformResult::kAll;
    else basic(QSX_CODE_LT, range.second);
  } else if (range.second == kMax) {
    basic(QSX_CODE_GE, range.first);
  } else {
    r.type = PredicateTransformResult::kRangeComparison;
    r.comp = QSX_CODE_RANGE;
    r.first_literal = range.first;
    r.second_literal = range.second;
  }
  return r;
}
}  // namespace

PredicateTransformResult TransformPredicateOnCompressedAttribute(const CompressedAttribute &attribute, TypeID type,
                                                                 ComparisonID comparison, const TypedLiteral &literal) {
  switch (type) {
    case kInt: return TransformT<std::int32_t>(attribute, comparison, literal.v.i32);
    case kLong: return TransformT<std::int64_t>(attribute, comparison, literal.v.i64);
    case kFloat: return TransformT<float>(attribute, comparison, literal.v.f32);
    case kDouble: return TransformT<double>(attribute, comparison, literal.v.f64);
    case kDate: return TransformBytes(attribute, type, comparison, &literal.v.i64, 8);
    case kChar: return TransformBytes(attribute, type, comparison, literal.text.data(), literal.text.size());
    default: throw ExecutionError("compressed attributes: INT / LONG / FLOAT / DOUBLE / DATE / CHAR(n)", QSX_ERR_UNSUPPORTED);
  }
}

void CompressValues(TypeID type, const void *values, std::int64_t n, CompressedAttribute *out,
                    std::vector<unsigned char> *codes_host, int value_width) {
  codes_host->clear();
  switch (type) {
    case kDate: BuildByteDictionary(type, 8, static_cast<const unsigned char *>(values), n, out, codes_host); break;
    case kChar:
      if (value_width > 0) BuildByteDictionary(type, value_width, static_cast<const unsigned char *>(values), n, out, codes_host);
      else out->kind = CompressedAttribute::kUncompressed;
      break;
    case kInt: BuildCompressedAttribute(type, static_cast<const std::int32_t *>(values), n, out, codes_host); break;
    case kLong: BuildCompressedAttribute(type, static_cast<const std::int64_t *>(values), n, out, codes_host); break;
    case kFloat: BuildCompressedAttribute(type, static_cast<const float *>(values), n, out, codes_host); break;
    case kDouble: BuildCompressedAttribute(type, static_cast<const double *>(values), n, out, codes_host); break;
    default: out->kind = CompressedAttribute::kUncompressed; break;
  }
}

void StorageBlock::compressAttribute(attribute_id a, const void *host_values) {
  if (g_host_memory) return;   // CPU plumbing mode keeps plain stripes
  const Type &t = relation_.getAttributeType(a);
  if (compressed_.empty()) compressed_.resize(relation_.size());
  CompressedAttribute &c = compressed_.at(a);
  std::vector<unsigned char> codes_host;
  CompressValues(t.id, host_values, num_tuples_, &c, &codes_host, t.width);
  if (c.kind == CompressedAttribute::kUncompressed) return;
  CheckStatus(qsx_device_alloc(codes_host.size() + 8, &c.codes), "qsx_device_alloc(codes)");
  CheckStatus(qsx_copy_to_device(c.codes, codes_host.data(), codes_host.size(), nullptr), "qsx_copy_to_device(codes)");
  if (c.kind == CompressedAttribute::kDictionary) {
    CheckStatus(qsx_device_alloc(c.dictionary_host.size() + 8, &c.dictionary), "qsx_device_alloc(dictionary)");
    CheckStatus(qsx_copy_to_device(c.dictionary, c.dictionary_host.data(), c.dictionary_host.size(), nullptr), "qsx_copy_to_device(dictionary)");
  }
  CheckStatus(qsx_stream_synchronize(nullptr), "qsx_stream_synchronize");
  if (slab_ == nullptr) qsx_device_free(stripes_.at(a));      // the values are gone until somebody asks for them (stripe())
  stripes_.at(a) = nullptr;
}

block_id StorageManager::createBlock(CatalogRelation *relation, std::int64_t capacity) {
  std::lock_guard<std::mutex> lock(mutex_);
  const block_id id = next_id_++;
  // first_row is fixed when the block is registered with its final size (returnBlock / loadBlock)
  blocks_[id] = std::make_shared<StorageBlock>(*relation, capacity, 0, /*one_allocation=*/true);
  return id;
}

block_id StorageManager::loadBlock(CatalogRelation *relation, const std::vector<const void *> &host_columns,
                                   std::int64_t num_tuples, partition_id part, const std::vector<bool> *compress,
                                   const std::vector<const std::uint64_t *> *null_bitmaps) {
  BlockReference block;
  block_id id;
  {
    std::lock_guard<std::mutex> lock(mutex_);
    id = next_id_++;
    std::int64_t &rows = rows_in_relation_[relation->getID()];
    block = std::make_shared<StorageBlock>(*relation, num_tuples, rows);
    rows += num_tuples;
    blocks_[id] = block;
  }
  for (std::size_t a = 0; a < relation->size(); ++a) {
    const std::size_t bytes = static_cast<std::size_t>(num_tuples) * relation->getAttributeType(static_cast<attribute_id>(a)).width;
    if (g_host_memory) {
      std::memcpy(block->stripe(static_cast<attribute_id>(a)), host_columns.at(a), bytes);
    } else {
      CheckStatus(qsx_copy_to_device(block->stripe(static_cast<attribute_id>(a)), host_columns.at(a), bytes, nullptr),
                  "qsx_copy_to_device");
    }
  }
  for (std::size_t a = 0; null_bitmaps != nullptr && a < relation->size() && a < null_bitmaps->size(); ++a) {
    if ((*null_bitmaps)[a] == nullptr) continue;
    std::uint64_t *dst = block->nullBitmap(static_cast<attribute_id>(a));
    if (dst == nullptr) throw ExecutionError("loadBlock: null bitmap given for a non-nullable attribute", QSX_ERR_INVALID_ARGUMENT);
    const std::size_t bytes = static_cast<std::size_t>((num_tuples + 63) / 64) * 8;
    if (g_host_memory) {
      std::memcpy(dst, (*null_bitmaps)[a], bytes);
    } else {
      CheckStatus(qsx_copy_to_device(dst, (*null_bitmaps)[a], bytes, nullptr), "qsx_copy_to_device(null bitmap)");
    }
  }
  if (!g_host_memory) CheckStatus(qsx_stream_synchronize(nullptr), "qsx_stream_synchronize");
  block->setNumTuples(num_tuples);
  if (compress != nullptr) {
    for (std::size_t a = 0; a < relation->size() && a < compress->size(); ++a) {
      if ((*compress)[a]) block->compressAttribute(static_cast<attribute_id>(a), host_columns.at(a));
    }
  }
  relation->addBlockToPartition(id, part);
  return id;
}

BlockReference StorageManager::getBlock(block_id id) const {
  std::lock_guard<std::mutex> lock(mutex_);
  auto it = blocks_.find(id);
  if (it == blocks_.end()) throw std::out_of_range("StorageManager::getBlock: unknown block");
  return it->second;
}

std::int64_t StorageManager::reserveRows(relation_id relation, std::int64_t num_tuples) {
  std::lock_guard<std::mutex> lock(mutex_);
  std::int64_t &rows = rows_in_relation_[relation];
  const std::int64_t first = rows;
  rows += num_tuples;
  return first;
}

// ---- reference block images ------------------------------------------------------------------------------------------------
namespace {
// protobuf wire format, as far as a StorageBlockHeader needs it (varints, fixed64 / fixed32, length-delimited fields)
struct WireReader {
  const unsigned char *at, *end;
  bool ok = true;
  std::uint64_t varint() {
    std::uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      if (at >= end) { ok = false; return 0; }
      const unsigned char b = *at++;
      v |= static_cast<std::uint64_t>(b & 0x7F) << shift;
      if ((b & 0x80) == 0) return v;
    }
    ok = false;
    return 0;
  }
  std::uint64_t fixed(int bytes) {
    if (end - at < bytes) { ok = false; return 0; }
    std::uint64_t v = 0;
    std::memcpy(&v, at, static_cast<std::size_t>(bytes));
    at += bytes;
    return v;
  }
  WireReader sub() {   // a length-delimited field
    const std::uint64_t len = varint();
    if (!ok || static_cast<std::uint64_t>(end - at) < len) { ok = false; return WireReader{at, at}; }
    WireReader r{at, at + len};
    at += len;
    return r;
  }
  void skip(int wire_type) {
    switch (wire_type) {
      case 0: (void)varint(); break;
      case 1: (void)fixed(8); break;
      case 2: (void)sub(); break;
      case 5: (void)fixed(4); break;
      default: ok = false;
    }
  }
};
[[noreturn]] void Malformed(const char *what) {
  throw ExecutionError(std::string("malformed block image: ") + what, QSX_ERR_INVALID_ARGUMENT);
}
}  // namespace

ReferenceBlockLayout ParseReferenceBlockImage(const CatalogRelation &relation, const void *prefix, std::size_t prefix_bytes,
                                              std::size_t image_bytes) {
  const unsigned char *bytes = static_cast<const unsigned char *>(prefix);
  if (prefix_bytes < sizeof(std::int32_t) || prefix_bytes > image_bytes) Malformed("shorter than its length word");
  std::int32_t header_length = 0;
  std::memcpy(&header_length, bytes, sizeof(header_length));
  if (header_length <= 0 || static_cast<std::size_t>(header_length) + sizeof(std::int32_t) > image_bytes) Malformed("header length");   // StorageBlock.cpp:112-117
  if (static_cast<std::size_t>(header_length) + sizeof(std::int32_t) + 8 > prefix_bytes) Malformed("the prefix handed in does not cover the block header");
  // StorageBlockHeader { layout = 1 (StorageBlockLayoutDescription { num_slots = 1; tuple_store_description = 2 {
  //   sub_block_type = 1; [sort_attribute_id = 64] }; index_description = 3 }); fixed64 tuple_store_size = 2; ... }
  WireReader header{bytes + sizeof(std::int32_t), bytes + sizeof(std::int32_t) + header_length};
  ReferenceBlockLayout out;
  bool have_layout = false, have_size = false;
  std::uint64_t sub_block_type = ~0ull;
  while (header.ok && header.at < header.end) {
    const std::uint64_t tag = header.varint();
    const int field = static_cast<int>(tag >> 3), wire = static_cast<int>(tag & 7);
    if (field == 1 && wire == 2) {
      WireReader layout = header.sub();
      have_layout = true;
      while (layout.ok && layout.at < layout.end) {
        const std::uint64_t ltag = layout.varint();
        if ((ltag >> 3) == 2 && (ltag & 7) == 2) {
          WireReader store = layout.sub();
          while (store.ok && store.at < store.end) {
            const std::uint64_t stag = store.varint();
            if ((stag >> 3) == 1 && (stag & 7) == 0) sub_block_type = store.varint();
            // sort_attribute_id: extension 64 of a basic column store, 128 of a compressed one (StorageBlockLayout.proto:38-58);
            // 129 (compressed_attribute_id, repeated) only says what the builder was ASKED to compress — what it did is in
            // the sub-block's own CompressedBlockInfo
            else if (((stag >> 3) == 64 || (stag >> 3) == 128) && (stag & 7) == 0) out.sort_attribute = static_cast<attribute_id>(static_cast<std::int32_t>(store.varint()));
            else store.skip(static_cast<int>(stag & 7));
          }
          if (!store.ok) Malformed("tuple store description");
        } else {
          layout.skip(static_cast<int>(ltag & 7));
        }
      }
      if (!layout.ok) Malformed("layout description");
    } else if (field == 2 && wire == 1) {
      out.tuple_store_size = static_cast<std::size_t>(header.fixed(8));
      have_size = true;
    } else {
      header.skip(wire);
    }
  }
  if (!header.ok || !have_layout || !have_size || sub_block_type == ~0ull) Malformed("block header");   // !IsInitialized()
  // TupleStorageSubBlockDescription: BASIC_COLUMN_STORE = 0, COMPRESSED_COLUMN_STORE = 2 (the row stores 1 and 3 have no stripes)
  if (sub_block_type != 0 && sub_block_type != 2) {
    throw ExecutionError("block image: the tuple store is a row store (only column stores are adopted in place)", QSX_ERR_UNSUPPORTED);
  }
  out.tuple_store_offset = sizeof(std::int32_t) + static_cast<std::size_t>(header_length);
  if (out.tuple_store_offset + out.tuple_store_size > image_bytes) Malformed("sub-block sizes exceed the block");   // :141-143
  if (out.tuple_store_size < 8) Malformed("tuple store smaller than its header");   // BlockMemoryTooSmall
  if (out.sort_attribute != kInvalidAttributeID && (out.sort_attribute < 0 || static_cast<std::size_t>(out.sort_attribute) >= relation.size())) {
    Malformed("sort attribute");
  }
  if (sub_block_type == 2) {
    // CompressedTupleStorageSubBlock::initializeCommon (storage/CompressedTupleStorageSubBlock.cpp:281-342) +
    // CompressedColumnStoreTupleStorageSubBlock::initialize (.cpp:755-798)
    out.compressed = true;
    const std::size_t store_end = out.tuple_store_offset + out.tuple_store_size;
    std::int32_t num_tuples = 0, info_bytes = 0;
    std::memcpy(&num_tuples, bytes + out.tuple_store_offset, 4);
    std::memcpy(&info_bytes, bytes + out.tuple_store_offset + 4, 4);
    if (num_tuples < 0 || info_bytes <= 0 || out.tuple_store_offset + 8 + static_cast<std::size_t>(info_bytes) > std::min(prefix_bytes, store_end)) {
      Malformed("compressed block info");
    }
    WireReader info{bytes + out.tuple_store_offset + 8, bytes + out.tuple_store_offset + 8 + info_bytes};
    std::vector<std::uint64_t> attribute_size, dictionary_size;
    std::vector<bool> has_nulls;
    bool have_bits = false;
    auto packed_fixed64 = [](WireReader r, std::vector<std::uint64_t> *into) {
      while (r.ok && r.at < r.end) into->push_back(r.fixed(8));
      return r.ok;
    };
    while (info.ok && info.at < info.end) {   // StorageBlockLayout.proto:128-150
      const std::uint64_t tag = info.varint();
      const int field = static_cast<int>(tag >> 3), wire = static_cast<int>(tag & 7);
      if (field == 1 && wire == 2) { if (!packed_fixed64(info.sub(), &attribute_size)) Malformed("attribute_size"); }
      else if (field == 1 && wire == 1) attribute_size.push_back(info.fixed(8));          // (unpacked encoding of the same field)
      else if (field == 2 && wire == 2) { if (!packed_fixed64(info.sub(), &dictionary_size)) Malformed("dictionary_size"); }
      else if (field == 2 && wire == 1) dictionary_size.push_back(info.fixed(8));
      else if (field == 3 && wire == 1) { out.null_bitmap_bits = static_cast<std::size_t>(info.fixed(8)); have_bits = true; }
      else if (field == 4 && wire == 2) { WireReader r = info.sub(); while (r.ok && r.at < r.end) has_nulls.push_back(r.varint() != 0); }
      else if (field == 4 && wire == 0) has_nulls.push_back(info.varint() != 0);
      else info.skip(wire);
    }
    if (!info.ok || !have_bits || attribute_size.size() != relation.size() || dictionary_size.size() != relation.size()) {
      Malformed("compressed block info");   // MalformedBlock (:288-291)
    }
    std::size_t at = out.tuple_store_offset + 8 + static_cast<std::size_t>(info_bytes), tuple_length = 0;
    for (std::size_t a = 0; a < relation.size(); ++a) {
      const Type &t = relation.getAttributeType(static_cast<attribute_id>(a));
      const std::size_t size = static_cast<std::size_t>(attribute_size[a]);
      out.attribute_size.push_back(size);
      tuple_length += size;
      if (dictionary_size[a] > 0) {
        if (size != 1 && size != 2 && size != 4) Malformed("code width of a dictionary-coded attribute");
        if (dictionary_size[a] < 8 || at + dictionary_size[a] > store_end) Malformed("dictionary size");
        out.dictionary_offset.push_back(at);
        out.dictionary_bytes.push_back(static_cast<std::size_t>(dictionary_size[a]));
        at += static_cast<std::size_t>(dictionary_size[a]);
      } else {
        out.dictionary_offset.push_back(static_cast<std::size_t>(-1));
        out.dictionary_bytes.push_back(0);
        if (size != static_cast<std::size_t>(t.width)) {   // truncation: INT / LONG only, to 1 / 2 / 4 bytes (:319-337)
          if ((t.id != kInt && t.id != kLong) || (size != 1 && size != 2 && size != 4) || size >= static_cast<std::size_t>(t.width)) Malformed("truncated attribute");
        }
      }
    }
    if (tuple_length == 0) Malformed("relation without attributes");
    const std::size_t per_bitmap = (out.null_bitmap_bits + 63) / 64 * 8;   // BitVector<false>::BytesNeeded
    for (std::size_t a = 0; a < relation.size(); ++a) {
      if (out.null_bitmap_bits > 0 && a < has_nulls.size() && has_nulls[a]) {
        out.null_bitmap_offset.push_back(at);
        at += per_bitmap;
      } else {
        out.null_bitmap_offset.push_back(static_cast<std::size_t>(-1));
      }
    }
    if (at > store_end) Malformed("dictionaries and null bitmaps exceed the tuple store");
    const std::size_t max_tuples = (store_end - at) / tuple_length;
    if (static_cast<std::size_t>(num_tuples) > max_tuples) Malformed("num_tuples");
    if (out.null_bitmap_bits > 0 && out.null_bitmap_bits < static_cast<std::size_t>(num_tuples)) Malformed("null bitmap shorter than the block");
    out.num_tuples = num_tuples;
    out.max_tuples = static_cast<std::int64_t>(max_tuples);
    for (std::size_t a = 0; a < relation.size(); ++a) {
      out.stripe_offset.push_back(at);
      at += max_tuples * out.attribute_size[a];
    }
    return out;
  }
  // BasicColumnStoreTupleStorageSubBlock.cpp:131-147
  std::size_t row_bytes = 0, nullable = 0;
  for (std::size_t a = 0; a < relation.size(); ++a) {
    const Type &t = relation.getAttributeType(static_cast<attribute_id>(a));
    row_bytes += static_cast<std::size_t>(t.width);
    nullable += t.nullable ? 1 : 0;
  }
  if (row_bytes == 0) Malformed("relation without attributes");
  auto bitmap_bytes = [](std::size_t bits) { return (bits + 63) / 64 * 8; };   // BitVector<false>::BytesNeeded
  std::size_t max_tuples = ((out.tuple_store_size - 8) << 3) / ((row_bytes << 3) + nullable);
  if (max_tuples == 0) Malformed("no room for one tuple");
  if (nullable * bitmap_bytes(max_tuples) + 8 > out.tuple_store_size) Malformed("no room for the null bitmaps");
  max_tuples = (out.tuple_store_size - 8 - nullable * bitmap_bytes(max_tuples)) / row_bytes;
  if (max_tuples == 0) Malformed("no room for one tuple");
  const std::size_t per_bitmap = bitmap_bytes(max_tuples);
  out.max_tuples = static_cast<std::int64_t>(max_tuples);
  std::int32_t num_tuples = 0;
  std::memcpy(&num_tuples, bytes + out.tuple_store_offset, sizeof(num_tuples));
  if (num_tuples < 0 || static_cast<std::size_t>(num_tuples) > max_tuples) Malformed("num_tuples");
  out.num_tuples = num_tuples;
  std::size_t at = out.tuple_store_offset + 8;
  for (std::size_t a = 0; a < relation.size(); ++a) {
    if (relation.getAttributeType(static_cast<attribute_id>(a)).nullable) {
      out.null_bitmap_offset.push_back(at);
      at += per_bitmap;
    } else {
      out.null_bitmap_offset.push_back(static_cast<std::size_t>(-1));
    }
  }
  for (std::size_t a = 0; a < relation.size(); ++a) {
    out.stripe_offset.push_back(at);
    at += max_tuples * static_cast<std::size_t>(relation.getAttributeType(static_cast<attribute_id>(a)).width);
  }
  if (at > out.tuple_store_offset + out.tuple_store_size) Malformed("stripes exceed the tuple store");
  if (out.sort_attribute != kInvalidAttributeID && (out.sort_attribute < 0 || static_cast<std::size_t>(out.sort_attribute) >= relation.size())) {
    Malformed("sort attribute");
  }
  return out;
}

block_id StorageManager::adoptBlockImage(CatalogRelation *relation, void *image_dev, std::size_t image_bytes, partition_id part) {
  // the block header is a few hundred bytes: fetch a prefix, parse on the host
  const std::size_t prefix_bytes = std::min<std::size_t>(image_bytes, 16384);
  std::vector<unsigned char> prefix(prefix_bytes);
  if (g_host_memory) {
    std::memcpy(prefix.data(), image_dev, prefix_bytes);
  } else {
    CheckStatus(qsx_copy_to_host(prefix.data(), image_dev, prefix_bytes, CurrentStream()), "qsx_copy_to_host(block header)");
  }
  const ReferenceBlockLayout layout = ParseReferenceBlockImage(*relation, prefix.data(), prefix_bytes, image_bytes);
  std::vector<void *> stripes, nulls;
  char *base = static_cast<char *>(image_dev);
  for (std::size_t a = 0; a < relation->size(); ++a) {
    stripes.push_back(base + layout.stripe_offset[a]);
    nulls.push_back(layout.null_bitmap_offset[a] == static_cast<std::size_t>(-1) ? nullptr : base + layout.null_bitmap_offset[a]);
  }
  BlockReference block = std::make_shared<StorageBlock>(*relation, layout.num_tuples, stripes, nulls);
  block->setExternalRange(image_dev, image_bytes);
  block->setSortColumn(layout.sort_attribute);
  auto fetch = [&](void *dst, std::size_t offset, std::size_t bytes) {
    if (g_host_memory) std::memcpy(dst, base + offset, bytes);
    else CheckStatus(qsx_copy_to_host(dst, base + offset, bytes, CurrentStream()), "qsx_copy_to_host(dictionary)");
  };
  for (std::size_t a = 0; layout.compressed && a < relation->size(); ++a) {
    const Type &t = relation->getAttributeType(static_cast<attribute_id>(a));
    const bool coded = layout.dictionary_offset[a] != static_cast<std::size_t>(-1);
    if (!coded && layout.attribute_size[a] == static_cast<std::size_t>(t.width)) continue;   // stored as values
    if (g_host_memory) throw ExecutionError("adoptBlockImage: compressed blocks need device memory", QSX_ERR_UNSUPPORTED);
    CompressedAttribute c;
    c.code_width = static_cast<int>(layout.attribute_size[a]);
    c.codes = base + layout.stripe_offset[a];
    if (!coded) {
      c.kind = CompressedAttribute::kTruncated;   // the value itself, zero-extended (CompressedBlockBuilder.cpp:508-566)
    } else {
      // {uint32 num_codes; uint32 null_code; num_codes values in ascending order} (compression/CompressionDictionary.hpp:46-58)
      std::uint32_t head[2] = {0, 0};
      fetch(head, layout.dictionary_offset[a], 8);
      const std::size_t value_bytes = static_cast<std::size_t>(head[0]) * static_cast<std::size_t>(t.width);
      if (8 + value_bytes > layout.dictionary_bytes[a]) {
        throw ExecutionError("malformed block image: dictionary shorter than its entry count (variable-length dictionaries are not adopted)",
                             QSX_ERR_INVALID_ARGUMENT);
      }
      c.kind = CompressedAttribute::kDictionary;
      c.num_codes = head[0];
      c.dictionary = base + layout.dictionary_offset[a] + 8;
      c.dictionary_host.resize(value_bytes);
      if (value_bytes != 0) fetch(c.dictionary_host.data(), layout.dictionary_offset[a] + 8, value_bytes);
      if (t.id == kChar || t.id == kDate) c.value_width = t.width;
      if (head[1] != 0xFFFFFFFFu && layout.num_tuples > 0) {
        // a NULL is the code num_codes (CompressionDictionary.hpp:49-52, 114-115): the tuples that carry it become the
        // attribute's null bitmap, which every operator of this layer already honours — a predicate's match under a NULL
        // is taken back, a NULL join key or group-by key drops the tuple, aggregates skip NULL arguments; the decoded value
        // under a NULL (one entry past the dictionary's last) is never looked at
        const std::size_t words = static_cast<std::size_t>((layout.num_tuples + 63) / 64);
        void *bitmap = nullptr, *count = nullptr;
        CheckStatus(qsx_device_alloc(words * 8 + 8, &bitmap), "qsx_device_alloc(null bitmap)");
        CheckStatus(qsx_device_alloc(8, &count), "qsx_device_alloc(count)");
        const int rc = qsx_select_codes(c.code_width, c.codes, layout.num_tuples, QSX_CODE_EQ, head[1], 0, nullptr, static_cast<std::uint64_t *>(bitmap),
                                        static_cast<std::int64_t *>(count), CurrentStream());
        if (rc == QSX_OK) (void)qsx_stream_synchronize(CurrentStream());
        qsx_device_free(count);
        if (rc != QSX_OK) {
          qsx_device_free(bitmap);
          CheckStatus(rc, "qsx_select_codes(NULL code)");
        }
        if (!t.nullable) {
          qsx_device_free(bitmap);
          throw ExecutionError("malformed block image: a NULL code in an attribute the relation declares NOT NULL", QSX_ERR_INVALID_ARGUMENT);
        }
        block->setNullBitmap(static_cast<attribute_id>(a), bitmap);
      }
    }
    block->adoptCompressedAttribute(static_cast<attribute_id>(a), std::move(c));
  }
  block_id id;
  {
    std::lock_guard<std::mutex> lock(mutex_);
    id = next_id_++;
    std::int64_t &rows = rows_in_relation_[relation->getID()];
    block->setFirstRow(rows);
    rows += layout.num_tuples;
    blocks_[id] = block;
  }
  relation->addBlockToPartition(id, part);
  return id;
}

block_id StorageManager::createViewBlock(block_id parent, std::int64_t first_tuple, std::int64_t num_tuples) {
  BlockReference p = getBlock(parent);
  BlockReference view = std::make_shared<StorageBlock>(p, first_tuple, num_tuples);
  std::lock_guard<std::mutex> lock(mutex_);
  const block_id id = next_id_++;
  blocks_[id] = view;
  return id;
}

void StorageManager::deleteBlockOrBlobFile(block_id id) {
  std::lock_guard<std::mutex> lock(mutex_);
  blocks_.erase(id);
}

namespace host_internal {
// The tuples of `block` selected by `filter` (nullptr = all) whose attributes `attrs` are all non-NULL, as a device
// bitmap — or nullptr when no attribute is nullable (the caller keeps using `filter`).  This is the reference's
// check_for_null_keys skip (HashTable.hpp:1409-1418, 2158-2160) and the NULL argument skip of the aggregate handles
// (AggregationHandleSum.hpp:105-120), done once per block on the TupleIdSequence instead of per value.
std::unique_ptr<DeviceBuffer> NotNullFilter(const StorageBlock &block, const std::vector<attribute_id> &attrs, const std::uint64_t *filter) {
  const std::int64_t n = block.numTuples();
  std::unique_ptr<DeviceBuffer> out;
  for (attribute_id a : attrs) {
    const std::uint64_t *nulls = block.nullBitmap(a);
    if (nulls == nullptr) continue;
    const bool first = out == nullptr;
    if (first) out.reset(new DeviceBuffer(static_cast<std::size_t>((n + 63) / 64) * 8 + 8));
    if (n == 0) continue;
    if (first && filter == nullptr) {
      CheckStatus(qsx_bitmap_combine(3, nulls, nullptr, n, static_cast<std::uint64_t *>(out->ptr), CurrentStream()), "qsx_bitmap_combine");
    } else {
      CheckStatus(qsx_bitmap_combine(2, first ? filter : static_cast<const std::uint64_t *>(out->ptr), nulls, n,
                                     static_cast<std::uint64_t *>(out->ptr), CurrentStream()), "qsx_bitmap_combine");
    }
  }
  return out;
}

// Null bits of the rows `tids` of one block's attribute -> dst (an output block's null bitmap).
void GatherBlockNulls(const StorageBlock &block, attribute_id attr, const void *tids, std::int64_t n, std::uint64_t *dst) {
  const std::uint64_t *seg = block.nullBitmap(attr);
  const std::int64_t zero = 0;
  CheckStatus(qsx_bitmap_gather_segmented(1, &seg, &zero, static_cast<const std::int32_t *>(tids), n, dst, CurrentStream()),
              "qsx_bitmap_gather_segmented");
}

// The null bits of the selected tuples of `block` follow the values of a projection: output attribute i takes the
// bits of input attribute selection[i] at the tuples set in `bitmap` (bulkInsertTuplesWithRemappedAttributes copies
// value and null bit together, storage/BasicColumnStoreTupleStorageSubBlock.cpp:339-425).
void ProjectNullBitmaps(const StorageBlock &block, const std::vector<attribute_id> &selection, const void *bitmap,
                        std::int64_t num_selected, StorageBlock *out) {
  const std::int64_t n = block.numTuples();
  std::unique_ptr<DeviceBuffer> tids;
  for (std::size_t i = 0; i < selection.size(); ++i) {
    if (selection[i] == kInvalidAttributeID) continue;          // an expression's value: no bitmap to carry over
    if (block.nullBitmap(selection[i]) == nullptr) continue;   // the output bitmap stays all-zero
    std::uint64_t *dst = out->nullBitmap(static_cast<attribute_id>(i));
    if (dst == nullptr) throw ExecutionError("projection of a nullable attribute into a non-nullable one", QSX_ERR_INVALID_ARGUMENT);
    if (num_selected == 0) continue;
    if (tids == nullptr) {
      tids.reset(new DeviceBuffer(static_cast<std::size_t>(n) * 4 + 16));
      const std::size_t ws_bytes = qsx_compact_workspace_bytes(n);
      DeviceBuffer ws(ws_bytes), count(8);
      CheckStatus(qsx_bitmap_to_tids(static_cast<const std::uint64_t *>(bitmap), n, 0, static_cast<std::int32_t *>(tids->ptr),
                                     static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()), "qsx_bitmap_to_tids");
      CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");   // ws / count are locals
    }
    GatherBlockNulls(block, selection[i], tids->ptr, num_selected, dst);
  }
  if (tids != nullptr) CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
}
}  // namespace host_internal

}  // namespace quickstep
