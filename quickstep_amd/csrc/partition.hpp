// partition.hpp — internal interface of the K9 partition scatter (partition.hip).
#ifndef QSX_CSRC_PARTITION_HPP_
#define QSX_CSRC_PARTITION_HPP_

#include "common.hpp"

namespace qsx {

size_t partition_workspace_bytes(int64_t n, int num_partitions);

// mode 0: the reference's partition function (identity hash, catalog/PartitionSchemeHeader.hpp:200-214);
// mode 1: top bits of a mixing hash, P a power of two (internal re-partitioning).  Stable; writes
// num_partitions + 1 row offsets to out_offsets_dev.
int partition_scatter_impl(int mode, int key_type, const void *keys_dev, int64_t n, int num_partitions, int ncols,
                           const void *const *cols, const int32_t *widths, void *const *out_cols, int64_t *out_offsets_dev,
                           void *workspace_dev, size_t workspace_bytes, hipStream_t stream);

}  // namespace qsx

#endif  // QSX_CSRC_PARTITION_HPP_
