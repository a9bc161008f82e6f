// device_common.hpp — device-side helpers shared by every kernel.  Kept free of host / libc++
// includes on purpose: this file is also part of the source bundle the run-time plan-shape
// compiler hands to hipRTC (agg_jit.hip), where only the HIP device built-ins exist.
#ifndef QSX_CSRC_DEVICE_COMMON_HPP_
#define QSX_CSRC_DEVICE_COMMON_HPP_

namespace qsx {

constexpr int kWave = 64;
constexpr int kCUs = 256;            // MI355X
constexpr int kMaxGridBlocks = 2048; // 8 x 256-thread blocks per CU: grid-stride above this

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

// A pointer that was loaded from memory (a table of a run of blocks) or that went through a struct is "generic" to the
// compiler: accesses through it become flat_load / flat_store, which count against the LDS counter as well — next to LDS
// DMA that serialises a kernel (the Q1 aggregation under a filter: 5.0 instead of 3.4 ms per 600 M rows when the filter
// pointer travelled through a struct).  This tells the compiler that the pointer is to device memory.
// (Through an INTEGER: generic -> global -> generic pointer casts are looked through by the address-space inference only
// when the generic pointer has a known origin; a pointer made from a loaded table word has none, and the round trip through
// the pointer types left every access of the run kernels a flat_load / flat_store.  integer -> global pointer -> generic is
// a global pointer by construction.)
template <typename T>
__device__ __forceinline__ T *as_global(T *p) {
  return (T *)(__attribute__((address_space(1))) T *)reinterpret_cast<uintptr_t>(p);
}

// The same for ONE access, with nothing left to inference: the access itself goes through a global-address-space pointer
// (as_global() relies on the compiler propagating the cast to the uses, which it does not always do — join_dense.hpp's run
// kernels kept their flat_load / flat_store through it).
template <typename T>
__device__ __forceinline__ T load_global(const T *p) {
  return *(const __attribute__((address_space(1))) T *)p;
}
// (no <type_traits> here: this header is part of the run-time plan shapes' source, which hipRTC compiles without the
// standard library)
template <int BYTES> struct BitsOfSize;
template <> struct BitsOfSize<1> { typedef uint8_t type; };
template <> struct BitsOfSize<2> { typedef uint16_t type; };
template <> struct BitsOfSize<4> { typedef uint32_t type; };
template <> struct BitsOfSize<8> { typedef uint64_t type; };
template <typename T>
__device__ __forceinline__ T load_global_nt(const T *p) {
  if constexpr (__is_arithmetic(T)) {
    return __builtin_nontemporal_load((const __attribute__((address_space(1))) T *)p);
  } else {   // a value struct (DateValue): as the integer of its size
    typedef typename BitsOfSize<sizeof(T)>::type Bits;
    const Bits bits = __builtin_nontemporal_load((const __attribute__((address_space(1))) Bits *)p);
    T v;
    __builtin_memcpy(&v, &bits, sizeof(T));
    return v;
  }
}
template <typename T>
__device__ __forceinline__ void store_global(T v, T *p) {
  *(__attribute__((address_space(1))) T *)p = v;
}
template <typename T>
__device__ __forceinline__ void store_global_nt(T v, T *p) {
  __builtin_nontemporal_store(v, (__attribute__((address_space(1))) T *)p);
}

// 16 bytes of a stripe that is read once (a scan): non-temporal, through a global pointer.  A kernel that only reads gets
// 10 % more out of HBM this way (tools/ubench/read_ceiling.hip: 6.2 -> 6.8-7.2 TB/s).
__device__ __forceinline__ uint4 stream_load16(const void *p) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_nontemporal_load((const __attribute__((address_space(1))) u32x4 *)reinterpret_cast<uintptr_t>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
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

// QSX_DATE: the reference's DateLit {int32 year; uint8 month; uint8 day; 2 bytes of padding}
// (types/DatetimeLit.hpp:38-43) as the 8 bytes it occupies in a column stripe.  Ordered by year, month, day (:65-90);
// the padding is never looked at.
__host__ __device__ constexpr long long date_ordered(unsigned long long raw) {
  return static_cast<long long>(static_cast<int>(raw & 0xFFFFFFFFull)) * 65536 +
         static_cast<long long>(((raw >> 32) & 0xFFull) << 8 | ((raw >> 40) & 0xFFull));
}
constexpr unsigned long long kDateValueMask = 0x0000FFFFFFFFFFFFull;   // year, month, day
struct DateValue {
  unsigned long long raw;
  __host__ __device__ constexpr long long key() const { return date_ordered(raw); }
  __host__ __device__ constexpr bool operator==(const DateValue &o) const { return key() == o.key(); }
  __host__ __device__ constexpr bool operator!=(const DateValue &o) const { return key() != o.key(); }
  __host__ __device__ constexpr bool operator<(const DateValue &o) const { return key() < o.key(); }
  __host__ __device__ constexpr bool operator<=(const DateValue &o) const { return key() <= o.key(); }
  __host__ __device__ constexpr bool operator>(const DateValue &o) const { return key() > o.key(); }
  __host__ __device__ constexpr bool operator>=(const DateValue &o) const { return key() >= o.key(); }
};

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


}  // namespace qsx

#endif  // QSX_CSRC_DEVICE_COMMON_HPP_
