// agg_factored.hip — the kernels of the factored aggregation over code stripes (agg_factored.hpp) and their launchers, in a
// translation unit of their own: the direct-load kernel is instantiated per signature (128 code objects), which aggregate.hip's
// already long compile should not wait for.
// Reference loops: storage/AggregationOperationState.cpp:428-474, storage/ThreadPrivateCompactKeyHashTable.cpp:216-304 over
// storage/CompressedColumnStoreValueAccessor.hpp:90-150.
#include "agg_factored_kernels.hpp"

#include <mutex>
#include <type_traits>

namespace qsx {

int launch_factored_coef(const DevConfig &dc, const FactoredCoefArgs &ca_in, hipStream_t s, int num_blocks) {
  FactoredCoefArgs ca = ca_in;
  ca.num_blocks = num_blocks;
  const int threads = ca.cells + kFacMaxDict;   // a thread per cell and per dictionary code
  const int across = (threads + kABlock - 1) / kABlock;
  // (a block's tables are ~10 us of dependent loads and interpreted arithmetic whatever the grid: as many workgroups as fit the
  // device at once — eight per CU — and each walks its share of the blocks; 1860 blocks: 2 per CU 0.29 ms, one per block 0.19 ms)
  const int fit = 8 * kCUs / across + 1;
  const int rows = num_blocks < fit ? num_blocks : fit;
  hipLaunchKernelGGL(factored_coef_kernel, dim3(across, rows), dim3(kABlock), 0, s, dc, ca);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int launch_factored_predicate(const FactoredPredArgs &pa, long long tiles, hipStream_t s) {
  const long long groups = (tiles + kABlock / kWave - 1) / (kABlock / kWave), limit = 8ll * kCUs;   // a wave per tile
  hipLaunchKernelGGL(factored_predicate_kernel, dim3(static_cast<unsigned>(groups < limit ? groups : limit)), dim3(kABlock), 0, s, pa, tiles);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int launch_factored_staged(const FactoredArgs &a, size_t lds_bytes, int per_cu, int64_t n, const uint64_t *filter_dev, const HashTableView &g, hipStream_t s) {
  static std::mutex attr_mutex;
  static bool attr_set[2][16] = {};
  int device = 0;
  (void)hipGetDevice(&device);
  const int has_filter = filter_dev != nullptr ? 1 : 0;
  if (lds_bytes > 48 * 1024 && device >= 0 && device < 16) {
    std::lock_guard<std::mutex> lock(attr_mutex);
    if (!attr_set[has_filter][device]) {
      const void *kernel = has_filter ? reinterpret_cast<const void *>(&agg_factored_kernel<true>) : reinterpret_cast<const void *>(&agg_factored_kernel<false>);
      QSX_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      attr_set[has_filter][device] = true;
    }
  }
  const int64_t tiles = (n + kFacTileRows - 1) / kFacTileRows;
  const int grid = static_cast<int>(tiles < static_cast<int64_t>(per_cu) * kCUs ? tiles : static_cast<int64_t>(per_cu) * kCUs);
  if (has_filter) {
    hipLaunchKernelGGL(agg_factored_kernel<true>, dim3(grid), dim3(kABlock), lds_bytes, s, a, n, filter_dev, g);
  } else {
    hipLaunchKernelGGL(agg_factored_kernel<false>, dim3(grid), dim3(kABlock), lds_bytes, s, a, n, filter_dev, g);
  }
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

bool factored_direct_signature(const FactoredArgs &a, int key_width) {
  return !(a.nkeys < 1 || a.nkeys > 2 || a.ncell < 1 || a.ncell > 2 || a.nhist > 1 || a.ncar > 1 || (key_width != 1 && key_width != 4));
}

bool launch_factored_direct(const FactoredArgs &a, const FactoredArgs *a_dev, const FactoredDirectArgs &da, int key_width, size_t lds_bytes, int grid, int64_t n,
                            const uint64_t *filter_dev, const HashTableView &g, hipStream_t s, const FactoredRunArgs *runs) {
  if (!factored_direct_signature(a, key_width)) return false;
  auto launch = [&](auto filt, auto kw, auto nk, auto nc, auto nh, auto car) {
    constexpr bool F = decltype(filt)::value, CAR = decltype(car)::value;
    constexpr int KW = decltype(kw)::value, NK = decltype(nk)::value, NC = decltype(nc)::value, NH = decltype(nh)::value;
    if (runs != nullptr) {   // (a run of blocks: F = some block has a filter)
      hipLaunchKernelGGL((agg_factored_direct_kernel<F, KW, NK, NC, NH, CAR, true>), dim3(grid), dim3(kABlock), lds_bytes, s, a_dev, da, n, nullptr, g, *runs);
    } else {
      hipLaunchKernelGGL((agg_factored_direct_kernel<F, KW, NK, NC, NH, CAR, false>), dim3(grid), dim3(kABlock), lds_bytes, s, a_dev, da, n, filter_dev, g,
                         FactoredRunArgs{});
    }
  };
  using std::integral_constant;
  auto by_car = [&](auto filt, auto kw, auto nk, auto nc, auto nh) {
    if (a.ncar == 1) launch(filt, kw, nk, nc, nh, std::true_type{}); else launch(filt, kw, nk, nc, nh, std::false_type{});
  };
  auto by_nh = [&](auto filt, auto kw, auto nk, auto nc) {
    if (a.nhist == 1) by_car(filt, kw, nk, nc, integral_constant<int, 1>{}); else by_car(filt, kw, nk, nc, integral_constant<int, 0>{});
  };
  auto by_nc = [&](auto filt, auto kw, auto nk) {
    if (a.ncell == 2) by_nh(filt, kw, nk, integral_constant<int, 2>{}); else by_nh(filt, kw, nk, integral_constant<int, 1>{});
  };
  auto by_nk = [&](auto filt, auto kw) {
    if (a.nkeys == 2) by_nc(filt, kw, integral_constant<int, 2>{}); else by_nc(filt, kw, integral_constant<int, 1>{});
  };
  auto by_kw = [&](auto filt) {
    if (key_width == 4) by_nk(filt, integral_constant<int, 4>{}); else by_nk(filt, integral_constant<int, 1>{});
  };
  if (filter_dev != nullptr) by_kw(std::true_type{}); else by_kw(std::false_type{});
  return true;
}

}  // namespace qsx
