// scan.hpp — single-workgroup exclusive scan over per-tile counts.  Tile counts
// are tiny next to the data they summarise (one int32 per 4096 rows), so one
// 1024-thread workgroup walking them in chunks is far below the cost of the
// passes around it.
#ifndef QSX_CSRC_SCAN_HPP_
#define QSX_CSRC_SCAN_HPP_

#include "common.hpp"

namespace qsx {

// offsets has num_tiles + 1 entries; the total also goes to *total_out when non-null.
static __global__ __launch_bounds__(1024) void tile_scan_kernel(const int32_t *__restrict__ counts,
                                                                int64_t num_tiles,
                                                                int64_t *__restrict__ offsets,
                                                                int64_t *__restrict__ total_out) {
  __shared__ int64_t wave_totals[16];
  __shared__ int64_t carry;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < num_tiles; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const int64_t c = i < num_tiles ? counts[i] : 0;
    int64_t incl = c;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int64_t up = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += up;
    }
    if (lane == kWave - 1) wave_totals[wave] = incl;
    __syncthreads();
    int64_t wave_base = 0;
    for (int w = 0; w < wave; ++w) wave_base += wave_totals[w];
    const int64_t block_carry = carry;
    if (i < num_tiles) offsets[i] = block_carry + wave_base + incl - c;
    __syncthreads();
    if (threadIdx.x == 1023) carry = block_carry + wave_base + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    offsets[num_tiles] = carry;
    if (total_out != nullptr) *total_out = carry;
  }
}

}  // namespace qsx

#endif  // QSX_CSRC_SCAN_HPP_
