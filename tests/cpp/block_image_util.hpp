// block_image_util.hpp — test helper: WRITES a block image the way the reference's StorageBlock / StorageBlockLayout /
// BasicColumnStoreTupleStorageSubBlock lay one out (storage/StorageBlock.cpp:81-195, StorageBlockLayout.proto:96-124,
// BasicColumnStoreTupleStorageSubBlock.cpp:100-183), so that the adapter under test (ParseReferenceBlockImage /
// StorageManager::adoptBlockImage) reads bytes it did not produce itself.  max_tuples is computed here independently.
#ifndef QSX_TESTS_CPP_BLOCK_IMAGE_UTIL_HPP_
#define QSX_TESTS_CPP_BLOCK_IMAGE_UTIL_HPP_

#include <algorithm>
#include <cstring>
#include <vector>

#include "quickstep_gpu.hpp"

namespace block_image {

inline void PutVarint(std::vector<unsigned char> *out, std::uint64_t v) {
  while (v >= 0x80) {
    out->push_back(static_cast<unsigned char>(v | 0x80));
    v >>= 7;
  }
  out->push_back(static_cast<unsigned char>(v));
}
inline void PutBytes(std::vector<unsigned char> *out, const std::vector<unsigned char> &bytes) {
  PutVarint(out, bytes.size());
  out->insert(out->end(), bytes.begin(), bytes.end());
}

// StorageBlockHeader for a basic column store with `num_slots` slots, a tuple store of `tuple_store_size` bytes, no indices
inline std::vector<unsigned char> Header(std::uint64_t num_slots, std::uint64_t tuple_store_size, int sort_attribute, int sub_block_type = 0) {
  std::vector<unsigned char> store;                 // TupleStorageSubBlockDescription
  PutVarint(&store, (1 << 3) | 0);                  //   sub_block_type = 1 (varint)
  PutVarint(&store, static_cast<std::uint64_t>(sub_block_type));
  if (sort_attribute >= 0) {
    PutVarint(&store, (64 << 3) | 0);               //   [BasicColumnStore...Description.sort_attribute_id] = 64
    PutVarint(&store, static_cast<std::uint64_t>(sort_attribute));
  }
  std::vector<unsigned char> layout;                // StorageBlockLayoutDescription
  PutVarint(&layout, (1 << 3) | 0);                 //   num_slots = 1
  PutVarint(&layout, num_slots);
  PutVarint(&layout, (2 << 3) | 2);                 //   tuple_store_description = 2 (length-delimited)
  PutBytes(&layout, store);
  std::vector<unsigned char> header;                // StorageBlockHeader
  PutVarint(&header, (1 << 3) | 2);                 //   layout = 1
  PutBytes(&header, layout);
  PutVarint(&header, (2 << 3) | 1);                 //   tuple_store_size = 2 (fixed64)
  for (int i = 0; i < 8; ++i) header.push_back(static_cast<unsigned char>(tuple_store_size >> (8 * i)));
  return header;
}

// columns[a]: num_tuples values of attribute a; nulls[a]: num_tuples flags (empty = no NULLs); block_bytes = slots x 2 MB
inline std::vector<unsigned char> Build(const quickstep::CatalogRelation &relation, const std::vector<const void *> &columns,
                                        const std::vector<std::vector<bool>> &nulls, std::int64_t num_tuples, std::size_t block_bytes,
                                        int sort_attribute = -1, std::int64_t *max_tuples_out = nullptr) {
  // the header length does not depend on the value of the fixed64 field: size it with a placeholder first
  const std::size_t header_bytes = Header(block_bytes >> 21, 0, sort_attribute).size();
  const std::size_t tuple_store_size = block_bytes - sizeof(std::int32_t) - header_bytes;
  const std::vector<unsigned char> header = Header(block_bytes >> 21, tuple_store_size, sort_attribute);
  std::size_t row_bytes = 0, nullable = 0;
  for (std::size_t a = 0; a < relation.size(); ++a) {
    row_bytes += static_cast<std::size_t>(relation.getAttributeType(static_cast<quickstep::attribute_id>(a)).width);
    nullable += relation.getAttributeType(static_cast<quickstep::attribute_id>(a)).nullable ? 1 : 0;
  }
  // BasicColumnStoreTupleStorageSubBlock.cpp:131-147
  std::size_t max_tuples = ((tuple_store_size - 8) * 8) / (row_bytes * 8 + nullable);
  const std::size_t first_bitmap = (max_tuples + 63) / 64 * 8;
  max_tuples = (tuple_store_size - 8 - nullable * first_bitmap) / row_bytes;
  const std::size_t bitmap_bytes = (max_tuples + 63) / 64 * 8;
  if (max_tuples_out != nullptr) *max_tuples_out = static_cast<std::int64_t>(max_tuples);
  std::vector<unsigned char> image(block_bytes, 0xCD);             // unused bytes are not zero in a real buffer pool either
  const std::int32_t header_length = static_cast<std::int32_t>(header.size());
  std::memcpy(image.data(), &header_length, 4);
  std::memcpy(image.data() + 4, header.data(), header.size());
  unsigned char *at = image.data() + 4 + header.size();
  const std::int32_t sub_header[2] = {static_cast<std::int32_t>(num_tuples), 0};
  std::memcpy(at, sub_header, 8);
  at += 8;
  for (std::size_t a = 0; a < relation.size(); ++a) {
    if (!relation.getAttributeType(static_cast<quickstep::attribute_id>(a)).nullable) continue;
    std::memset(at, 0, bitmap_bytes);
    for (std::int64_t i = 0; i < num_tuples && !nulls[a].empty(); ++i) {
      if (!nulls[a][static_cast<std::size_t>(i)]) continue;
      std::uint64_t word;
      std::memcpy(&word, at + (i >> 6) * 8, 8);
      word |= 1ull << (63 - (i & 63));                             // BitVector<false>: MSB-first within a 64-bit word
      std::memcpy(at + (i >> 6) * 8, &word, 8);
    }
    at += bitmap_bytes;
  }
  for (std::size_t a = 0; a < relation.size(); ++a) {
    const std::size_t width = static_cast<std::size_t>(relation.getAttributeType(static_cast<quickstep::attribute_id>(a)).width);
    std::memcpy(at, columns[a], static_cast<std::size_t>(num_tuples) * width);
    at += max_tuples * width;
  }
  return image;
}

// ---- a CompressedColumnStoreTupleStorageSubBlock image ------------------------------------------------------------------------
// What CompressedBlockBuilder leaves behind (storage/CompressedBlockBuilder.cpp:262-420; layout read back by
// CompressedTupleStorageSubBlock::initializeCommon, storage/CompressedTupleStorageSubBlock.cpp:281-342, and
// CompressedColumnStoreTupleStorageSubBlock::initialize, .cpp:755-798).  The test decides per attribute HOW it is stored
// (the builder's size heuristics are pinned elsewhere: host_logic_test); here only the bytes matter.
struct Coding {
  enum Kind { kValues, kTruncated, kDictionary } kind = kValues;
  int code_width = 0;   // kTruncated / kDictionary
};
inline void PutFixed64(std::vector<unsigned char> *out, std::uint64_t v) {
  for (int i = 0; i < 8; ++i) out->push_back(static_cast<unsigned char>(v >> (8 * i)));
}
inline std::vector<unsigned char> HeaderCompressed(std::uint64_t num_slots, std::uint64_t tuple_store_size, int sort_attribute,
                                                   const std::vector<int> &compressed_attributes) {
  std::vector<unsigned char> store;                 // TupleStorageSubBlockDescription
  PutVarint(&store, (1 << 3) | 0);
  PutVarint(&store, 2);                             //   COMPRESSED_COLUMN_STORE
  PutVarint(&store, (128 << 3) | 0);                //   [CompressedColumnStore...Description.sort_attribute_id] = 128
  PutVarint(&store, static_cast<std::uint64_t>(sort_attribute));
  for (int a : compressed_attributes) {             //   repeated compressed_attribute_id = 129
    PutVarint(&store, (129 << 3) | 0);
    PutVarint(&store, static_cast<std::uint64_t>(a));
  }
  std::vector<unsigned char> layout;
  PutVarint(&layout, (1 << 3) | 0);
  PutVarint(&layout, num_slots);
  PutVarint(&layout, (2 << 3) | 2);
  PutBytes(&layout, store);
  std::vector<unsigned char> header;
  PutVarint(&header, (1 << 3) | 2);
  PutBytes(&header, layout);
  PutVarint(&header, (2 << 3) | 1);
  PutFixed64(&header, tuple_store_size);
  return header;
}
// columns / nulls as for Build(); the rows must be in ascending order of the sort attribute.  INT / LONG / DOUBLE attributes.
inline std::vector<unsigned char> BuildCompressed(const quickstep::CatalogRelation &relation, const std::vector<const void *> &columns,
                                                  const std::vector<std::vector<bool>> &nulls, std::int64_t num_tuples, std::size_t block_bytes,
                                                  int sort_attribute, const std::vector<Coding> &coding, std::int64_t *max_tuples_out = nullptr) {
  const std::size_t n = static_cast<std::size_t>(num_tuples), attrs = relation.size();
  std::vector<int> compressed_ids;
  for (std::size_t a = 0; a < attrs; ++a) if (coding[a].kind != Coding::kValues) compressed_ids.push_back(static_cast<int>(a));
  const std::size_t header_bytes = HeaderCompressed(block_bytes >> 21, 0, sort_attribute, compressed_ids).size();
  const std::size_t tuple_store_size = block_bytes - 4 - header_bytes;
  const std::vector<unsigned char> header = HeaderCompressed(block_bytes >> 21, tuple_store_size, sort_attribute, compressed_ids);
  auto value_of = [&](std::size_t a, std::size_t i) {   // the value's bytes as a 64-bit pattern (order-preserving compare below)
    const int width = relation.getAttributeType(static_cast<quickstep::attribute_id>(a)).width;
    std::uint64_t v = 0;
    std::memcpy(&v, static_cast<const char *>(columns[a]) + i * static_cast<std::size_t>(width), static_cast<std::size_t>(width));
    return v;
  };
  auto less = [&](std::size_t a, std::uint64_t x, std::uint64_t y) {
    switch (relation.getAttributeType(static_cast<quickstep::attribute_id>(a)).id) {
      case quickstep::kInt: return static_cast<std::int32_t>(x) < static_cast<std::int32_t>(y);
      case quickstep::kLong: return static_cast<std::int64_t>(x) < static_cast<std::int64_t>(y);
      default: { double dx, dy; std::memcpy(&dx, &x, 8); std::memcpy(&dy, &y, 8); return dx < dy; }
    }
  };
  // dictionaries: sorted distinct non-NULL values; a NULL is the code num_codes
  std::vector<std::vector<std::uint64_t>> dict(attrs);
  std::vector<bool> any_null(attrs, false);
  std::vector<unsigned char> dictionaries;
  std::vector<std::uint64_t> attribute_size(attrs), dictionary_size(attrs, 0);
  for (std::size_t a = 0; a < attrs; ++a) {
    const int width = relation.getAttributeType(static_cast<quickstep::attribute_id>(a)).width;
    for (std::size_t i = 0; i < n && !nulls[a].empty(); ++i) any_null[a] = any_null[a] || nulls[a][i];
    attribute_size[a] = coding[a].kind == Coding::kValues ? static_cast<std::uint64_t>(width) : static_cast<std::uint64_t>(coding[a].code_width);
    if (coding[a].kind != Coding::kDictionary) continue;
    for (std::size_t i = 0; i < n; ++i) if (nulls[a].empty() || !nulls[a][i]) dict[a].push_back(value_of(a, i));
    std::sort(dict[a].begin(), dict[a].end(), [&](std::uint64_t x, std::uint64_t y) { return less(a, x, y); });
    dict[a].erase(std::unique(dict[a].begin(), dict[a].end()), dict[a].end());
    const std::uint32_t head[2] = {static_cast<std::uint32_t>(dict[a].size()), any_null[a] ? static_cast<std::uint32_t>(dict[a].size()) : 0xFFFFFFFFu};
    const std::size_t before = dictionaries.size();
    dictionaries.insert(dictionaries.end(), reinterpret_cast<const unsigned char *>(head), reinterpret_cast<const unsigned char *>(head) + 8);
    for (std::uint64_t v : dict[a]) dictionaries.insert(dictionaries.end(), reinterpret_cast<const unsigned char *>(&v), reinterpret_cast<const unsigned char *>(&v) + width);
    dictionary_size[a] = dictionaries.size() - before;
  }
  // CompressedBlockInfo (StorageBlockLayout.proto:128-150)
  std::vector<unsigned char> info, packed;
  for (std::uint64_t v : attribute_size) PutFixed64(&packed, v);
  PutVarint(&info, (1 << 3) | 2);
  PutBytes(&info, packed);
  packed.clear();
  for (std::uint64_t v : dictionary_size) PutFixed64(&packed, v);
  PutVarint(&info, (2 << 3) | 2);
  PutBytes(&info, packed);
  bool bitmaps = false;
  for (std::size_t a = 0; a < attrs; ++a) bitmaps = bitmaps || (coding[a].kind == Coding::kValues && any_null[a]);
  PutVarint(&info, (3 << 3) | 1);
  PutFixed64(&info, bitmaps ? static_cast<std::uint64_t>(n) : 0);
  packed.clear();
  for (std::size_t a = 0; a < attrs; ++a) packed.push_back(coding[a].kind == Coding::kValues && any_null[a] ? 1 : 0);
  PutVarint(&info, (4 << 3) | 2);
  PutBytes(&info, packed);

  std::vector<unsigned char> image(block_bytes, 0xCD);
  const std::int32_t header_length = static_cast<std::int32_t>(header.size());
  std::memcpy(image.data(), &header_length, 4);
  std::memcpy(image.data() + 4, header.data(), header.size());
  unsigned char *at = image.data() + 4 + header.size();
  const std::int32_t sub_header[2] = {static_cast<std::int32_t>(num_tuples), static_cast<std::int32_t>(info.size())};
  std::memcpy(at, sub_header, 8);
  at += 8;
  std::memcpy(at, info.data(), info.size());
  at += info.size();
  std::memcpy(at, dictionaries.data(), dictionaries.size());
  at += dictionaries.size();
  const std::size_t bitmap_bytes = (n + 63) / 64 * 8;
  for (std::size_t a = 0; a < attrs && bitmaps; ++a) {
    if (!(coding[a].kind == Coding::kValues && any_null[a])) continue;
    std::memset(at, 0, bitmap_bytes);
    for (std::size_t i = 0; i < n; ++i) {
      if (!nulls[a][i]) continue;
      std::uint64_t word;
      std::memcpy(&word, at + (i >> 6) * 8, 8);
      word |= 1ull << (63 - (i & 63));
      std::memcpy(at + (i >> 6) * 8, &word, 8);
    }
    at += bitmap_bytes;
  }
  std::size_t tuple_length = 0;
  for (std::uint64_t v : attribute_size) tuple_length += static_cast<std::size_t>(v);
  const std::size_t max_tuples = static_cast<std::size_t>(image.data() + block_bytes - at) / tuple_length;
  if (max_tuples_out != nullptr) *max_tuples_out = static_cast<std::int64_t>(max_tuples);
  for (std::size_t a = 0; a < attrs; ++a) {
    const std::size_t size = static_cast<std::size_t>(attribute_size[a]);
    for (std::size_t i = 0; i < n && i < max_tuples; ++i) {
      std::uint64_t stored = value_of(a, i);
      const bool is_null = !nulls[a].empty() && nulls[a][i];
      if (coding[a].kind == Coding::kDictionary) {
        stored = is_null ? dict[a].size()
                         : static_cast<std::uint64_t>(std::lower_bound(dict[a].begin(), dict[a].end(), stored,
                                                                       [&](std::uint64_t x, std::uint64_t y) { return less(a, x, y); }) - dict[a].begin());
      }
      std::memcpy(at + i * size, &stored, size);   // (little-endian: the low `size` bytes = the truncated value / the code)
    }
    at += max_tuples * size;
  }
  return image;
}

}  // namespace block_image

#endif  // QSX_TESTS_CPP_BLOCK_IMAGE_UTIL_HPP_
