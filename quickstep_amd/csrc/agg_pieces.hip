// agg_pieces.hip — see agg_pieces.hpp.
#include "agg_pieces.hpp"

#include "common.hpp"

namespace qsx {

namespace {
constexpr int kPBlockThreads = 256;      // the bounds kernel; the smallest workgroup of the pieces kernel
constexpr int kPMaxBlockThreads = 1024;  // a table that leaves room for one workgroup per CU gets a large one (launch_agg_pieces)
constexpr int kPieceMaxProbes = 48;      // then the row takes the global path: a piece with more groups than its table holds stays O(1) per row

__device__ __forceinline__ unsigned long long piece_key_code(const PieceArgs &a, int64_t row) {
  unsigned long long code = 0;
#pragma unroll
  for (int k = 0; k < QSX_MAX_KEYS; ++k) {
    if (k < a.num_keys) {
      unsigned long long v;
      switch (a.key_width[k]) {
        case 1: v = load_global(&static_cast<const uint8_t *>(a.key_col[k])[row]); break;
        case 2: v = load_global(&static_cast<const uint16_t *>(a.key_col[k])[row]); break;
        case 4: v = load_global(&static_cast<const uint32_t *>(a.key_col[k])[row]); break;
        default: v = load_global(&static_cast<const unsigned long long *>(a.key_col[k])[row]); break;
      }
      code |= v << a.key_shift[k];
    }
  }
  return code;
}
__device__ __forceinline__ unsigned int piece_of(unsigned long long code) {
  return static_cast<unsigned int>((mix64(code) * 0x9E3779B97F4A7C15ull) >> (64 - kPieceBits));
}

// bounds[p] = first row whose piece is >= p (the rows are ordered by piece): one binary search per piece.
__global__ __launch_bounds__(kPBlockThreads) void piece_bounds_kernel(PieceArgs a) {
  const int p = static_cast<int>(blockIdx.x) * kPBlockThreads + threadIdx.x;
  if (p > kNumPieces) return;
  int64_t lo = 0, hi = a.n;   // first row in [lo, hi] with piece >= p
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (piece_of(piece_key_code(a, mid)) >= static_cast<unsigned int>(p)) hi = mid; else lo = mid + 1;
  }
  a.bounds[p] = p == kNumPieces ? a.n : lo;
}

// The workgroup's table: bounded linear probing on 64-bit codes (any cheap hash: the table is private).
__device__ __forceinline__ int piece_slot(unsigned long long *l_keys, int S, unsigned long long code) {
  if (code == kEmptyCode) return -1;
  int s = static_cast<int>(((static_cast<uint32_t>(code) ^ static_cast<uint32_t>(code >> 32)) * 0x9E3779B9u) >> 16) & (S - 1);
  const int limit = S < kPieceMaxProbes ? S : kPieceMaxProbes;
  for (int probes = 0; probes < limit; ++probes) {
    unsigned long long k = l_keys[s];
    if (k == kEmptyCode) k = atomicCAS(&l_keys[s], kEmptyCode, code);
    if (k == kEmptyCode || k == code) return s;
    s = (s + 1) & (S - 1);
  }
  return -1;
}

// global_find_or_insert (agg_common.hpp) without its add to the state's group counter: every group of a piece is new to the
// table the first time its piece is flushed — 10^7 adds to one address — so the flush counts its inserts and adds them once a wave.
__device__ __forceinline__ unsigned long long piece_global_slot(const HashTableView &g, unsigned long long code, unsigned int &fresh) {
  if (code == kEmptyCode) return g.cap;
  unsigned long long s = code_slot(code, g.shift);
  const unsigned long long limit = g.cap < kGlobalMaxProbes ? g.cap : kGlobalMaxProbes;
  for (unsigned long long probes = 0; probes < limit; ++probes) {
    unsigned long long k = __hip_atomic_load(&g.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == code) return s;
    if (k == kEmptyCode) {
      k = atomicCAS(&g.keys[s], kEmptyCode, code);
      if (k == kEmptyCode) {
        ++fresh;
        return s;
      }
      if (k == code) return s;
    }
    s = (s + 1) & (g.cap - 1);
  }
  return global_find_or_insert(g, code);   // (the spill log / the overflow flag: the rare way out stays in one place)
}

template <int NS>
struct PieceRow {
  unsigned long long code;
  unsigned long long val[NS > 0 ? NS : 1];
};
template <int NS>
__device__ __forceinline__ void load_piece_row(const PieceArgs &a, int64_t row, PieceRow<NS> &r) {
  r.code = piece_key_code(a, row);
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    switch (a.sum_type[j]) {
      case QSX_INT: r.val[j] = static_cast<unsigned long long>(static_cast<long long>(load_global(&static_cast<const int32_t *>(a.sum_col[j])[row]))); break;
      case QSX_LONG: r.val[j] = static_cast<unsigned long long>(load_global(&static_cast<const long long *>(a.sum_col[j])[row])); break;
      default: r.val[j] = load_global(&static_cast<const unsigned long long *>(a.sum_col[j])[row]); break;   // the DOUBLE's bits
    }
    // MIN / MAX combine as int64 (agg_common.hpp AccKind): a DOUBLE through the order-preserving map
    if (a.sum_kind[j] >= kAccMinI64 && a.sum_type[j] == QSX_DOUBLE) {
      r.val[j] = static_cast<unsigned long long>(ordered_from_bits(static_cast<long long>(r.val[j])));
    }
  }
}
template <int NS>
__device__ __forceinline__ void add_piece_row(const PieceArgs &a, const HashTableView &g, unsigned long long *l_keys, unsigned long long *l_acc, int S,
                                              const PieceRow<NS> &r, unsigned int &fresh) {
  const int slot = piece_slot(l_keys, S, r.code);
  if (slot >= 0) {
    atomicAdd(&l_acc[slot], 1ull);
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      unsigned long long *p = &l_acc[static_cast<size_t>(j + 1) * S + slot];
      if (a.sum_kind[j] == kAccSumI64) atomicAdd(p, r.val[j]);
      else if (a.sum_kind[j] == kAccSumF64) atomic_add_f64(reinterpret_cast<double *>(p), __longlong_as_double(static_cast<long long>(r.val[j])));
      else lds_add(p, r.val[j], a.sum_kind[j]);   // MIN / MAX
    }
  } else {   // the sentinel code, or more groups in this piece than the table holds: straight to the state
    const unsigned long long gs = piece_global_slot(g, r.code, fresh);
    if (gs != ~0ull) {
      global_add(g, 0, gs, 1ull, kAccSumI64);
#pragma unroll
      for (int j = 0; j < NS; ++j) global_add(g, j + 1, gs, r.val[j], a.sum_kind[j]);
    }
  }
}

template <int NS>
__global__ __launch_bounds__(kPMaxBlockThreads) void agg_pieces_kernel(PieceArgs a, HashTableView g) {
  const int threads = static_cast<int>(blockDim.x);
  extern __shared__ __align__(16) char lds[];
  const int S = a.S;
  unsigned long long *l_keys = reinterpret_cast<unsigned long long *>(lds);
  unsigned long long *l_acc = l_keys + S;      // [NS + 1][S]: row count, then the sums
  unsigned int fresh = 0;                      // groups this thread entered into the state's table
  for (int piece = blockIdx.x; piece < kNumPieces; piece += gridDim.x) {
    const int64_t lo = a.bounds[piece], hi = a.bounds[piece + 1];
    if (lo >= hi) continue;   // (wave-uniform: the bounds are the same for every thread)
    for (int i = threadIdx.x; i < S; i += threads) l_keys[i] = kEmptyCode;
    for (int i = threadIdx.x; i < S; i += threads) l_acc[i] = 0;   // the row counts
#pragma unroll
    for (int j = 0; j < NS; ++j) {   // every accumulator starts from its kind's identity (0 for the sums)
      const unsigned long long identity = static_cast<unsigned long long>(acc_identity(a.sum_kind[j]));
      for (int i = threadIdx.x; i < S; i += threads) l_acc[static_cast<size_t>(j + 1) * S + i] = identity;
    }
    __syncthreads();
    int64_t row = lo + threadIdx.x;
    for (; row + threads < hi; row += 2 * threads) {   // two rows' loads in flight: a wave waits for its loads and little else
      PieceRow<NS> r0, r1;
      load_piece_row<NS>(a, row, r0);
      load_piece_row<NS>(a, row + threads, r1);
      add_piece_row<NS>(a, g, l_keys, l_acc, S, r0, fresh);
      add_piece_row<NS>(a, g, l_keys, l_acc, S, r1, fresh);
    }
    if (row < hi) {
      PieceRow<NS> r0;
      load_piece_row<NS>(a, row, r0);
      add_piece_row<NS>(a, g, l_keys, l_acc, S, r0, fresh);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < S; i += threads) {   // one global update per group and accumulator
      const unsigned long long code = l_keys[i];
      if (code == kEmptyCode) continue;
      const unsigned long long gs = piece_global_slot(g, code, fresh);
      if (gs == ~0ull) continue;   // (the overflow flag is raised: the host reports the lost rows)
      global_add(g, 0, gs, l_acc[i], kAccSumI64);
#pragma unroll
      for (int j = 0; j < NS; ++j) global_add(g, j + 1, gs, l_acc[static_cast<size_t>(j + 1) * S + i], a.sum_kind[j]);
    }
    __syncthreads();
  }
  // the state's group counter: one add per wave
  unsigned long long sum = fresh;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, kWave);
  if (lane_id() == 0 && sum != 0) atomicAdd(g.ngroups, sum);
}
}  // namespace

int launch_agg_pieces(const PieceArgs &args, const HashTableView &g, hipStream_t stream) {
  hipLaunchKernelGGL(piece_bounds_kernel, dim3((kNumPieces + 1 + kPBlockThreads - 1) / kPBlockThreads), dim3(kPBlockThreads), 0, stream, args);
  QSX_CHECK_LAUNCH();
  const size_t lds = static_cast<size_t>(args.S) * 8 * (args.num_sums + 2);
  if (lds > 160 * 1024) return QSX_ERR_CAPACITY;
  int per_cu = static_cast<int>((160 * 1024) / (lds + 1024));
  per_cu = per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu);
  const int grid = kNumPieces < per_cu * kCUs ? kNumPieces : per_cu * kCUs;
  // the loads of a row are what a wave waits for: 16 waves per CU whatever the table leaves room for
  const int threads = per_cu >= 4 ? kPBlockThreads : (per_cu >= 2 ? 2 * kPBlockThreads : kPMaxBlockThreads);
  auto launch = [&](auto ns) -> int {
    constexpr int NS = decltype(ns)::value;
    if (lds > 48 * 1024) {   // (a property of (kernel, device); setting it again is a cheap host call)
      QSX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_pieces_kernel<NS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL(agg_pieces_kernel<NS>, dim3(grid), dim3(threads), lds, stream, args, g);
    return QSX_OK;
  };
  int rc = QSX_ERR_UNSUPPORTED;
  switch (args.num_sums) {
    case 0: rc = launch(std::integral_constant<int, 0>{}); break;
    case 1: rc = launch(std::integral_constant<int, 1>{}); break;
    case 2: rc = launch(std::integral_constant<int, 2>{}); break;
    case 3: rc = launch(std::integral_constant<int, 3>{}); break;
    case 4: rc = launch(std::integral_constant<int, 4>{}); break;
    case 5: rc = launch(std::integral_constant<int, 5>{}); break;
    case 6: rc = launch(std::integral_constant<int, 6>{}); break;
    case 7: rc = launch(std::integral_constant<int, 7>{}); break;
    case 8: rc = launch(std::integral_constant<int, 8>{}); break;
    default: break;
  }
  if (rc != QSX_OK) return rc;
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

}  // namespace qsx
