// block_image_util.hpp — test helper: WRITES a block image the way the reference's StorageBlock / StorageBlockLayout /
// BasicColumnStoreTupleStorageSubBlock lay one out (storage/StorageBlock.cpp:81-195, StorageBlockLayout.proto:96-124,
// BasicColumnStoreTupleStorageSubBlock.cpp:100-183), so that the adapter under test (ParseReferenceBlockImage /
// StorageManager::adoptBlockImage) reads bytes it did not produce itself.  max_tuples is computed here independently.
#ifndef QSX_TESTS_CPP_BLOCK_IMAGE_UTIL_HPP_
#define QSX_TESTS_CPP_BLOCK_IMAGE_UTIL_HPP_

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

}  // namespace block_image

#endif  // QSX_TESTS_CPP_BLOCK_IMAGE_UTIL_HPP_
