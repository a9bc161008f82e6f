// agg_family_part.hip — one key signature of the ahead-of-time plan-shape family (agg_family.hpp), compiled once per part with
// -DQSX_FAMILY_PART=0..6: for one to six DOUBLE sums the update body over the canonical configuration — one stripe per column or a run
// of blocks, with or without a filter bitmap: 24 kernels.
// Part 0 also holds the lookup over all parts.
#include "agg_family.hpp"
#include "agg_hash_update.hpp"

#include <type_traits>

#ifndef QSX_FAMILY_PART
#error "compile with -DQSX_FAMILY_PART=<0..6>"
#endif

namespace qsx {

namespace {
constexpr int kPart = QSX_FAMILY_PART;
constexpr int kKT0 = kFamilyKeySignatures[kPart][0], kKT1 = kFamilyKeySignatures[kPart][1];
constexpr int kV = 4;   // rows per thread of a tile: what the registered shapes run with by default
}  // namespace

// The member's configuration planned with or without a filter word per 64 rows of the tile.
template <typename Shape, bool kFilter>
struct Planned {
  static constexpr Translated get() {
    Translated t = translate(Shape::config());
    plan_tile(t.dev, t.used_columns, kABlock * kV, kFilter);
    return t;
  }
};
template <typename Shape, bool kFilter>
__global__ __launch_bounds__(kABlock) void family_kernel(ColumnPointers cols, int64_t n, const uint64_t *__restrict__ filter, HashTableView g, int S,
                                                        int rep_shift, int nbuf, int ranges, const long long *__restrict__ pieces) {
  static constexpr Translated T = Planned<Shape, kFilter>::get();
  agg_hash_update_body<true, false, T.num_sums, kV>(T.dev, cols.p, nullptr, n, kFilter ? filter : nullptr, g, DenseView{}, S, rep_shift, nbuf, ranges, pieces);
}
template <typename Shape, bool kFilter>
__global__ __launch_bounds__(kABlock) void family_runs_kernel(int64_t n, HashTableView g, int S, int rep_shift, int nbuf, int ranges,
                                                             const long long *__restrict__ block_run) {
  static constexpr Translated T = Planned<Shape, kFilter>::get();
  agg_hash_update_body<true, false, T.num_sums, kV, false, kABlock, false, true>(T.dev, nullptr, nullptr, n, nullptr, g, DenseView{}, S, rep_shift, nbuf, ranges,
                                                                                  block_run);
}

namespace {
template <int NS>
int launch_member(const void *const *cols, int num_columns, int64_t n, const uint64_t *filter, const HashTableView &g, int S, int ranges,
                  const long long *pieces, hipStream_t stream, bool runs) {
  using Shape = ShapeFamily<kKT0, kKT1, NS>;
  constexpr int TR = kABlock * kV;
  constexpr Translated T = Planned<Shape, false>::get(), TF = Planned<Shape, true>::get();
  static_assert(T.status == QSX_OK && T.num_sums == NS, "the canonical configuration translates to NS accumulators");
  const bool filtered = filter != nullptr;
  ShapeGeometry geo{};
  const int rc = shape_launch_geometry(NS, S, filtered ? TF.dev.tile_bytes : T.dev.tile_bytes, &geo);
  if (rc != QSX_OK) return rc;
  const int64_t num_tiles = (n + TR - 1) / TR;
  const int64_t max_grid = static_cast<int64_t>(kCUs) * geo.per_cu;
  int grid = static_cast<int>(num_tiles * ranges < max_grid ? num_tiles * ranges : max_grid);
  grid = grid / ranges * ranges;
  if (grid < ranges) grid = ranges;
  ColumnPointers cp;
  for (int i = 0; i < QSX_MAX_COLUMNS; ++i) cp.p[i] = (!runs && i < num_columns) ? cols[i] : nullptr;
  // (hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device); setting it again is a cheap host call)
  auto go = [&](auto filt) -> int {
    constexpr bool F = decltype(filt)::value;
    if (runs) {
      QSX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&family_runs_kernel<Shape, F>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      hipLaunchKernelGGL((family_runs_kernel<Shape, F>), dim3(grid), dim3(kABlock), geo.lds, stream, n, g, S, geo.rep_shift, geo.nbuf, ranges, pieces);
    } else {
      QSX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&family_kernel<Shape, F>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      hipLaunchKernelGGL((family_kernel<Shape, F>), dim3(grid), dim3(kABlock), geo.lds, stream, cp, n, filter, g, S, geo.rep_shift, geo.nbuf, ranges, pieces);
    }
    return QSX_OK;
  };
  return filtered ? go(std::true_type{}) : go(std::false_type{});
}
}  // namespace

// the kernels, instantiated in BOTH compilation passes (the table below is host code: the device pass would not ask for them)
#define QSX_FAMILY_KERNELS(NS)                                                                                                                       \
  template __global__ void family_kernel<ShapeFamily<kKT0, kKT1, NS>, false>(ColumnPointers, int64_t, const uint64_t *__restrict__, HashTableView, int, int, \
                                                                             int, int, const long long *__restrict__);                                     \
  template __global__ void family_kernel<ShapeFamily<kKT0, kKT1, NS>, true>(ColumnPointers, int64_t, const uint64_t *__restrict__, HashTableView, int, int,  \
                                                                            int, int, const long long *__restrict__);                                      \
  template __global__ void family_runs_kernel<ShapeFamily<kKT0, kKT1, NS>, false>(int64_t, HashTableView, int, int, int, int, const long long *__restrict__); \
  template __global__ void family_runs_kernel<ShapeFamily<kKT0, kKT1, NS>, true>(int64_t, HashTableView, int, int, int, int, const long long *__restrict__);
QSX_FAMILY_KERNELS(1) QSX_FAMILY_KERNELS(2) QSX_FAMILY_KERNELS(3) QSX_FAMILY_KERNELS(4) QSX_FAMILY_KERNELS(5) QSX_FAMILY_KERNELS(6)
#undef QSX_FAMILY_KERNELS

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
