// common.hpp — shared host/device helpers of the gfx950 execution kernel.
// Wave size is 64 everywhere (CDNA4); nothing here is written for 32-wide warps.
#ifndef QSX_CSRC_COMMON_HPP_
#define QSX_CSRC_COMMON_HPP_

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/qsx.h"

namespace qsx {

constexpr int kWave = 64;
constexpr int kCUs = 256;            // MI355X
constexpr int kMaxGridBlocks = 2048; // 8 x 256-thread blocks per CU: grid-stride above this

// ---- host-side error plumbing -------------------------------------------
void set_last_error(const char *what, hipError_t err);
int device_ready();  // QSX_OK or QSX_ERR_NO_DEVICE

#define QSX_HIP_TRY(expr)                                        \
  do {                                                           \
    hipError_t qsx_err__ = (expr);                               \
    if (qsx_err__ != hipSuccess) {                               \
      ::qsx::set_last_error(#expr, qsx_err__);                   \
      return qsx_err__ == hipErrorOutOfMemory ? QSX_ERR_OUT_OF_MEMORY : QSX_ERR_HIP; \
    }                                                            \
  } while (0)

#define QSX_REQUIRE_DEVICE()                 \
  do {                                       \
    int qsx_dev__ = ::qsx::device_ready();   \
    if (qsx_dev__ != QSX_OK) return qsx_dev__; \
  } while (0)

#define QSX_CHECK_LAUNCH() QSX_HIP_TRY(hipGetLastError())

inline hipStream_t as_stream(qsx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int grid_for(int64_t work_items, int items_per_block) {
  int64_t blocks = (work_items + items_per_block - 1) / items_per_block;
  if (blocks < 1) blocks = 1;
  if (blocks > kMaxGridBlocks) blocks = kMaxGridBlocks;
  return static_cast<int>(blocks);
}

inline int type_width(int type) {
  switch (type) {
    case QSX_INT: case QSX_FLOAT: return 4;
    case QSX_LONG: case QSX_DOUBLE: return 8;
    default: return 0;
  }
}

inline uint64_t next_pow2(uint64_t v) {
  uint64_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

// ---- device helpers ---------------------------------------------------------
#if defined(__HIPCC__)

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// TupleIdSequence words are MSB-first (utility/BitVector.hpp:893-935); a wave
// ballot is LSB-first (bit l = lane l).  One s_brev_b64 converts either way.
__device__ __forceinline__ uint64_t msb_first(uint64_t ballot_mask) { return __brevll(ballot_mask); }

// Bit of row (64*w + lane) in an MSB-first word.
__device__ __forceinline__ bool msb_bit(uint64_t word, int lane) { return (word >> (63 - lane)) & 1u; }

// Number of set bits of an LSB-first wave mask strictly below the calling lane
// (v_mbcnt_lo/hi: no lane-id arithmetic needed).
__device__ __forceinline__ int rank_below(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

template <typename T>
__device__ __forceinline__ T wave_reduce_add(T v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

__device__ __forceinline__ uint64_t wave_broadcast_first(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
  return (static_cast<uint64_t>(hi) << 32) | lo;
}

template <typename T>
__device__ __forceinline__ bool compare_op(T a, int op, T b) {
  switch (op) {
    case QSX_EQ: return a == b;
    case QSX_NE: return a != b;
    case QSX_LT: return a < b;
    case QSX_LE: return a <= b;
    case QSX_GT: return a > b;
    default: return a >= b;
  }
}

// Multiplicative (Fibonacci) hashes for the device tables.  Hash values never
// show in results (only row order depends on them and that is unspecified,
// relational_operators/tests/HashJoinOperator_unittest.cpp:480), so the device
// tables are free not to use the reference's identity hash.
__device__ __forceinline__ uint32_t mix32(uint32_t k) { return k * 0x9E3779B1u; }
__device__ __forceinline__ uint64_t mix64(uint64_t k) {
  k ^= k >> 32;
  k *= 0x9E3779B97F4A7C15ull;
  k ^= k >> 29;
  return k;
}

// 64-bit global atomics (native on gfx950: global_atomic_add_x2 / _add_f64).
__device__ __forceinline__ void atomic_add_i64(int64_t *p, int64_t v) {
  atomicAdd(reinterpret_cast<unsigned long long *>(p), static_cast<unsigned long long>(v));
}
__device__ __forceinline__ void atomic_add_f64(double *p, double v) { unsafeAtomicAdd(p, v); }

#endif  // __HIPCC__

}  // namespace qsx

#endif  // QSX_CSRC_COMMON_HPP_
