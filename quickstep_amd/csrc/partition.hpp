// partition.hpp — internal interface of the K9 partition scatter (partition.hip).
#ifndef QSX_CSRC_PARTITION_HPP_
#define QSX_CSRC_PARTITION_HPP_

#include "common.hpp"

namespace qsx {

size_t partition_workspace_bytes(int64_t n, int num_partitions);

// mode 0: the reference's partition function (identity hash, catalog/PartitionSchemeHeader.hpp:200-214);
// mode 1: top bits of a mixing hash, P a power of two (internal re-partitioning).  Stable; writes
// num_partitions + 1 row offsets to out_offsets_dev.
// align_rows > 0 (mode 1): every partition's run starts at a multiple of align_rows output rows (the output columns need
// room for n + align_rows * num_partitions rows) and out_offsets_dev receives 2 * num_partitions values instead:
// [p] = first row of piece p, [num_partitions + p] = its row count.
int partition_scatter_impl(int mode, int key_type, const void *keys_dev, int64_t n, int num_partitions, int ncols,
                           const void *const *cols, const int32_t *widths, void *const *out_cols, int64_t *out_offsets_dev,
                           void *workspace_dev, size_t workspace_bytes, hipStream_t stream, int align_rows = 0);

// Same, routing on the key code packed on the fly from several key columns (little-endian at bit offsets key_shifts,
// ThreadPrivateCompactKeyHashTable.cpp:216-232): mode 1 only, num_partitions a power of two >= 2.
int partition_scatter_packed_keys(int num_keys, const void *const *key_cols, const int *key_widths, const int *key_shifts,
                                  int64_t n, int num_partitions, int ncols, const void *const *cols, const int32_t *widths,
                                  void *const *out_cols, int64_t *out_offsets_dev, void *workspace_dev, size_t workspace_bytes,
                                  hipStream_t stream, int align_rows);

// One pass (stable or not: the first pass has no earlier order to keep) of an LSD ordering by the MIXING hash of the key code packed on the fly (the hash that addresses the aggregation's
// global table): 64 buckets by the digit (hash >> shift) & 63.  Two passes (shift 52, then 58) order the rows by the hash's top
// 12 bits — 4096 pieces whose groups are disjoint (the two-level partitioned aggregation).  out_offsets_dev: 65 int64.
// Workspace: partition_workspace_bytes(n, 64).
int partition_scatter_packed_digit(int num_keys, const void *const *key_cols, const int *key_widths, const int *key_shifts, int64_t n,
                                   int shift, bool stable, int ncols, const void *const *cols, const int32_t *widths, void *const *out_cols,
                                   int64_t *out_offsets_dev, void *workspace_dev, size_t workspace_bytes, hipStream_t stream);

// One stable LSD radix-sort pass over 64-bit keys: 64 buckets by the digit (key >> shift) & 63, ties keep their order.
// out_offsets_dev: 65 int64.  Workspace: partition_workspace_bytes(n, 64).
int partition_scatter_digit(const unsigned long long *keys64_dev, int64_t n, int shift, int ncols, const void *const *cols,
                            const int32_t *widths, void *const *out_cols, int64_t *out_offsets_dev, void *workspace_dev,
                            size_t workspace_bytes, hipStream_t stream);

}  // namespace qsx

#endif  // QSX_CSRC_PARTITION_HPP_
