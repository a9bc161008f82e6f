// agg_pieces.hpp — the consumer of the TWO-LEVEL partitioned aggregation: more groups than one partition pass brings into LDS
// (one K9 pass makes 64 pieces: beyond ~10^5 groups a piece's groups no longer fit a workgroup's table and the rows pay
// NS + 1 global atomics each, at the atomic units' 23.7 G/s: COUNT + SUM 9.6 ms per 100 M rows at 10^6 groups, 20.9 at 10^7).
// Two K9 passes on digits of the mixing hash (partition.hpp partition_scatter_packed_digit: the low digit in any order, then
// the high digit keeping the first pass's order) order the rows by the hash's top 12 bits: 4096 pieces with disjoint groups — a few hundred to a few thousand groups each — and, because the global table is
// addressed by the same hash (agg_common.hpp code_slot), piece p's groups are one contiguous 1/4096 of the table.  A workgroup
// takes a piece at a time: its rows once through a workgroup-private LDS table (direct loads: a piece starts at any row),
// then one global update per group and accumulator.  3.4 ms at 10^6 groups, 5.0 at 10^7 (two scatters 0.86 + 1.01, two
// histograms 0.18 + 0.16, this kernel 0.53 / 1.78; tools/agg_large_groups.py, profiles/r06_two_level_kernel_stats.csv).
// A piece's table holds est_groups / 4096 groups at load <= 1/3 where LDS allows (128 KB), never beyond 0.7 (the path is not
// taken then: aggregate.hip two_level_slots); a row whose group finds no slot within 48 probes — an estimate that was too
// low — goes to the global table directly, like every row of the one-pass path does.
// Reference loops: storage/AggregationOperationState.cpp:548-614 (the partitioned aggregation), storage/
// PackedPayloadHashTable.hpp:838-909 (upsert per row).
// Plans it serves (aggregate.hip two_level_plan): hash states with a key code of <= 8 bytes, SUM / AVG / COUNT / MIN / MAX over
// plain DOUBLE, INT or LONG columns and over DOUBLE expressions (their values arrive as a stripe), COUNT(*); rows past the
// state's predicate (agg_update's K1 prepass + compaction); no NULLs or codes — the other plans keep the one-pass path.
#ifndef QSX_CSRC_AGG_PIECES_HPP_
#define QSX_CSRC_AGG_PIECES_HPP_

#include "agg_common.hpp"

namespace qsx {

constexpr int kPieceBits = 12;
constexpr int kNumPieces = 1 << kPieceBits;

struct PieceArgs {
  int num_keys;
  const void *key_col[QSX_MAX_KEYS];
  int key_width[QSX_MAX_KEYS];
  int key_shift[QSX_MAX_KEYS];
  int num_sums;
  const void *sum_col[kMaxSums];
  int sum_type[kMaxSums];     // QSX_DOUBLE / QSX_INT / QSX_LONG: the argument column's type
  int sum_kind[kMaxSums];     // AccKind: kAccSumF64 / kAccSumI64 / kAccMinI64 / kAccMaxI64
  long long *bounds;          // [kNumPieces + 1]: first row of every piece (written by the bounds kernel)
  int64_t n;
  int S;                      // slots of a workgroup's table (power of two)
};

// bounds of the 4096 pieces of rows ordered by (mixing hash >> 52), then every piece through a workgroup's LDS table into g.
int launch_agg_pieces(const PieceArgs &args, const HashTableView &g, hipStream_t stream);

}  // namespace qsx

#endif  // QSX_CSRC_AGG_PIECES_HPP_
