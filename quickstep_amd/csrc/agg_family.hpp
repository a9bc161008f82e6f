// agg_family.hpp — the ahead-of-time plan-shape FAMILY of the hash aggregation: every GROUP BY of one or two keys of CHAR(1) /
// INT / LONG type whose packed key fits 8 bytes, with one to six SUM / AVG aggregates over plain DOUBLE columns and any number of
// COUNT(*) — the reference's whole PackedPayload / ThreadPrivateCompactKey update loop for such plans
// (storage/AggregationOperationState.cpp:428-474, storage/ThreadPrivateCompactKeyHashTable.cpp:203-304,
// storage/PackedPayloadHashTable.hpp:838-909) is ONE template instantiation whatever the plan; here the kernel body
// (agg_hash_update.hpp) becomes straight-line code only when its configuration is a compile-time constant, which until round 6
// meant: the two registered shapes of agg_shapes.hpp, or a shape the run-time compiler had seen (a compiler on the box, or a code
// object recorded for exactly that plan) — anything else met the interpreter at ~0.18 of the HBM peak.
//
// A member of the family is named by (key type 0, key type 1 or none, number of DOUBLE sums): its CANONICAL configuration has
// the keys as columns 0 .. K-1 and the summed columns behind them in accumulator order.  A state whose translated plan
// (agg_translate.hpp) has that form is served by the member's kernel with its stripes handed over in canonical order — the state
// image (key codes, accumulator columns) is the same either way, so finalize, merge and export do not know the difference.
// 7 key signatures x 6 = 42 members, each with and without a filter bitmap, over one stripe per column or a run of blocks (168
// kernels), one translation unit per key signature (agg_family_part.hip, -DQSX_FAMILY_PART=n: ~20 s each).  A state's own
// predicate (attribute OP literal terms on plain columns) becomes the call's filter by a K1 pass in front of the update
// (one stripe per column; a run of blocks with a predicate keeps the other kernels).
// Not covered (they keep the run-time shapes / the interpreter): nullable or compressed columns, expressions, INT / LONG sums,
// MIN / MAX, keys wider than 8 packed bytes, the group directory's mid-size group counts.
#ifndef QSX_CSRC_AGG_FAMILY_HPP_
#define QSX_CSRC_AGG_FAMILY_HPP_

#include "agg_shapes.hpp"

namespace qsx {

// Key type codes of a member: the key's width in bytes (1 = CHAR(1), 4 = INT, 8 = LONG), 0 = no second key.
constexpr int kFamilyMaxSums = 6;
constexpr int kFamilyParts = 7;
constexpr int kFamilyKeySignatures[kFamilyParts][2] = {{1, 0}, {4, 0}, {8, 0}, {1, 1}, {1, 4}, {4, 1}, {4, 4}};

template <int KT0, int KT1, int NS>
struct ShapeFamily : ShapeBase<ShapeFamily<KT0, KT1, NS>> {
  static constexpr qsx_agg_config_t config() {
    ConfigBuilder b(QSX_AGG_COMPACT_KEY);
    auto key_column = [&](int kt) {
      if (kt == 1) b.column(QSX_CHAR, 1); else if (kt == 4) b.column(QSX_INT, 4); else b.column(QSX_LONG, 8);
    };
    key_column(KT0);
    if (KT1 != 0) key_column(KT1);
    constexpr int K = KT1 != 0 ? 2 : 1;
    for (int j = 0; j < NS; ++j) b.column(QSX_DOUBLE, 8);
    b.key(0);
    if (KT1 != 0) b.key(1);
    for (int j = 0; j < NS; ++j) b.agg(QSX_AGG_SUM, Col(K + j));
    b.agg(QSX_AGG_COUNT_STAR, Col(0));
    return b.c;
  }
};

// cols in CANONICAL order.  filter: the call's TupleIdSequence (nullptr: every row).  runs: the rows are a run of blocks —
// `pieces` is its table (agg_common.hpp BlockRunView) with the stripes of every block in CANONICAL order, cols is not read, and
// `filter` only says whether some block has a filter (the table carries them).
typedef int (*FamilyLauncher)(const void *const *cols, int num_columns, int64_t n, const uint64_t *filter, const HashTableView &g, int S, int ranges,
                              const long long *pieces, hipStream_t stream, bool runs);
struct FamilyEntry {
  int kt0, kt1, ns;
  FamilyLauncher launch;
};
// The member for a key signature and a sum count, or nullptr (agg_family_part.hip: one table per part).
const FamilyEntry *find_family_entry(int kt0, int kt1, int ns);

// Launch geometry of a shape kernel for NS sums, S slots and a tile of tile_bytes (aggregate.hip: the numbers launch_shape_v derives).
struct ShapeGeometry {
  int rep_shift, nbuf, per_cu;
  size_t lds;
};
int shape_launch_geometry(int NS, int S, int tile_bytes, ShapeGeometry *out);

}  // namespace qsx

#endif  // QSX_CSRC_AGG_FAMILY_HPP_
