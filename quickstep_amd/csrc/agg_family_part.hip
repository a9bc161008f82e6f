// agg_family_part.hip — one key signature of the ahead-of-time plan-shape family (agg_family.hpp), compiled once per part with
// -DQSX_FAMILY_PART=0..6: six kernels (one to six DOUBLE sums) of agg_hash_shape_kernel over the canonical configuration.
// Part 0 also holds the lookup over all parts.
#include "agg_family.hpp"
#include "agg_hash_update.hpp"

#ifndef QSX_FAMILY_PART
#error "compile with -DQSX_FAMILY_PART=<0..6>"
#endif

namespace qsx {

namespace {
constexpr int kPart = QSX_FAMILY_PART;
constexpr int kKT0 = kFamilyKeySignatures[kPart][0], kKT1 = kFamilyKeySignatures[kPart][1];
constexpr int kV = 4;   // rows per thread of a tile: what the registered shapes run with by default

template <int NS>
int launch_member(const void *const *cols, int num_columns, int64_t n, const HashTableView &g, int S, int ranges, const long long *pieces,
                  hipStream_t stream) {
  using Shape = ShapeFamily<kKT0, kKT1, NS>;
  constexpr int TR = kABlock * kV;
  constexpr Translated T = Shape::translated(TR);
  static_assert(T.status == QSX_OK && T.num_sums == NS, "the canonical configuration translates to NS accumulators");
  ShapeGeometry geo{};
  const int rc = shape_launch_geometry(NS, S, T.dev.tile_bytes, &geo);
  if (rc != QSX_OK) return rc;
  // (a property of (kernel, device); setting it again is a cheap host call)
  QSX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_hash_shape_kernel<Shape, kV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int64_t num_tiles = (n + TR - 1) / TR;
  const int64_t max_grid = static_cast<int64_t>(kCUs) * geo.per_cu;
  int grid = static_cast<int>(num_tiles * ranges < max_grid ? num_tiles * ranges : max_grid);
  grid = grid / ranges * ranges;
  if (grid < ranges) grid = ranges;
  ColumnPointers cp;
  for (int i = 0; i < QSX_MAX_COLUMNS; ++i) cp.p[i] = i < num_columns ? cols[i] : nullptr;
  hipLaunchKernelGGL((agg_hash_shape_kernel<Shape, kV>), dim3(grid), dim3(kABlock), geo.lds, stream, cp, n, g, S, geo.rep_shift, geo.nbuf, ranges, pieces);
  return QSX_OK;
}
}  // namespace

// the six kernels, instantiated in BOTH compilation passes (the table below is host code: the device pass would not ask for them)
#define QSX_FAMILY_KERNEL(NS)                                                                                                            \
  template __global__ void agg_hash_shape_kernel<ShapeFamily<kKT0, kKT1, NS>, kV>(ColumnPointers, int64_t, HashTableView, int, int, int, int, \
                                                                                  const long long *__restrict__);
QSX_FAMILY_KERNEL(1) QSX_FAMILY_KERNEL(2) QSX_FAMILY_KERNEL(3) QSX_FAMILY_KERNEL(4) QSX_FAMILY_KERNEL(5) QSX_FAMILY_KERNEL(6)
#undef QSX_FAMILY_KERNEL

#if !defined(__HIP_DEVICE_COMPILE__)
#define QSX_FAMILY_TABLE_NAME_(p) kFamilyPart##p
#define QSX_FAMILY_TABLE_NAME(p) QSX_FAMILY_TABLE_NAME_(p)
extern const FamilyEntry QSX_FAMILY_TABLE_NAME(QSX_FAMILY_PART)[kFamilyMaxSums] = {
    {kKT0, kKT1, 1, &launch_member<1>}, {kKT0, kKT1, 2, &launch_member<2>}, {kKT0, kKT1, 3, &launch_member<3>},
    {kKT0, kKT1, 4, &launch_member<4>}, {kKT0, kKT1, 5, &launch_member<5>}, {kKT0, kKT1, 6, &launch_member<6>},
};

#if QSX_FAMILY_PART == 0
extern const FamilyEntry kFamilyPart1[kFamilyMaxSums], kFamilyPart2[kFamilyMaxSums], kFamilyPart3[kFamilyMaxSums], kFamilyPart4[kFamilyMaxSums],
    kFamilyPart5[kFamilyMaxSums], kFamilyPart6[kFamilyMaxSums];
const FamilyEntry *find_family_entry(int kt0, int kt1, int ns) {
  static const FamilyEntry *const parts[kFamilyParts] = {kFamilyPart0, kFamilyPart1, kFamilyPart2, kFamilyPart3, kFamilyPart4, kFamilyPart5, kFamilyPart6};
  if (ns < 1 || ns > kFamilyMaxSums) return nullptr;
  for (int p = 0; p < kFamilyParts; ++p) {
    if (kFamilyKeySignatures[p][0] == kt0 && kFamilyKeySignatures[p][1] == kt1) return &parts[p][ns - 1];
  }
  return nullptr;
}
#endif
#endif  // host pass

}  // namespace qsx
