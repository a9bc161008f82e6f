// aggregate.hip — K6/K7/K8 group-by aggregation with fused K11 expression
// evaluation and the state's predicate, K10 finalize, partial-state merge.
//
// Reference loops replaced (paths in the Quickstep tree):
//   storage/AggregationOperationState.cpp:428-474 (aggregateBlock), :476-519,
//   storage/ThreadPrivateCompactKeyHashTable.cpp:203-304 (+ .hpp:125-169),
//   storage/CollisionFreeVectorTable.hpp:530-645,
//   storage/PackedPayloadHashTable.hpp:838-909,
//   expressions/scalar/ScalarBinaryExpression.cpp:100-195 (temp vectors -> fused),
//   finalize: AggregationOperationState.cpp:641-948,
//             ThreadPrivateCompactKeyHashTable.cpp:365-421,
//             CollisionFreeVectorTable.hpp:647-727.
//
// State layout in HBM (one allocation per state, "image" = what
// qsx_agg_state_export copies; all words 8 bytes):
//   hash strategies (SINGLE_STATE / COMPACT_KEY / GENERIC), capacity C = 2^k:
//     keys   [C + 1]            packed key code, ~0 = empty; slot C is reserved
//                               for the one code that equals ~0
//     col 0  [C + 1] int64      row count of the group
//     col j  [C + 1] int64|f64  j-th accumulator (j = 1..NS): SUM as int64 / f64, MIN / MAX as int64
//                               (doubles through the order-preserving map of agg_common.hpp)
//   COLLISION_FREE, E = num_entries (max_key + 1):
//     exist  [ceil(E/64)]       existence bits, LSB-first (bit k of word k>>6)
//     col 0  [E] int64          row count (only when some COUNT/AVG needs it)
//     col j  [E] int64|f64      j-th accumulator
//
// The update kernel itself lives in agg_hash_update.hpp (one body for the hash strategies and the
// dense sink); this file holds the host side: state, translation, launch geometry, the choice
// between AOT plan shape / run-time plan shape (agg_jit.hip) / interpreter, merge and finalize.
// Integer SUM/COUNT/MIN/MAX use integer atomics (exact, order-independent); double
// sums are order-dependent in the last bits, the contract is 1e-6 relative.

#include "agg_common.hpp"
#include "agg_translate.hpp"
#include "agg_hash_update.hpp"
#include "agg_factored.hpp"
#include "agg_shapes.hpp"
#include "agg_family.hpp"
#include "agg_pieces.hpp"
#include "agg_jit.hpp"
#include "comm.hpp"
#include "partition.hpp"
#include "scan.hpp"

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <vector>

namespace qsx {

// AccKind of every state column, by value into the kernels that combine whole columns.
struct ColKinds {
  int kind[QSX_MAX_AGGS + 1];
};

// Columns whose identity is not the all-zero word (MIN / MAX) are filled after the state's memset.
__global__ __launch_bounds__(kABlock) void fill_identity_kernel(unsigned long long *__restrict__ states, long long col_stride,
                                                               long long col_words, int num_cols, ColKinds kinds) {
  for (int col = 0; col < num_cols; ++col) {
    if (kinds.kind[col] < kAccMinI64) continue;
    const unsigned long long identity = static_cast<unsigned long long>(acc_identity(kinds.kind[col]));
    unsigned long long *p = states + static_cast<long long>(col) * col_stride;
    for (long long i = static_cast<long long>(blockIdx.x) * kABlock + threadIdx.x; i < col_words;
         i += static_cast<long long>(gridDim.x) * kABlock) {
      p[i] = identity;
    }
  }
}

// ---------------------------------------------------------------------------
// merge of exported images
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kABlock) void merge_hash_kernel(const unsigned long long *__restrict__ image,
                                                            unsigned long long src_cap, int num_cols,
                                                            ColKinds kinds, HashTableView g) {
  const unsigned long long *src_keys = image;
  const unsigned long long *src_states = image + (src_cap + 1);
  for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * kABlock + threadIdx.x;
       i <= src_cap; i += static_cast<unsigned long long>(gridDim.x) * kABlock) {
    const unsigned long long cnt = src_states[i];
    if (cnt == 0) continue;
    const unsigned long long code = i == src_cap ? kEmptyCode : src_keys[i];
    if (i != src_cap && code == kEmptyCode) continue;
    const unsigned long long gs = global_find_or_insert(g, code);
    if (gs == ~0ull) continue;
    for (int col = 0; col < num_cols; ++col) {
      global_add(g, col, gs, src_states[static_cast<unsigned long long>(col) * (src_cap + 1) + i], kinds.kind[col]);
    }
  }
}

__global__ __launch_bounds__(kABlock) void merge_dense_kernel(const unsigned long long *__restrict__ image,
                                                             unsigned long long *__restrict__ dst,
                                                             long long exist_words, long long num_entries,
                                                             int num_cols, ColKinds kinds) {
  const long long total = exist_words + static_cast<long long>(num_cols) * num_entries;
  for (long long i = static_cast<long long>(blockIdx.x) * kABlock + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * kABlock) {
    const unsigned long long v = image[i];
    if (i < exist_words) {
      if (v != 0) atomicOr(&dst[i], v);
    } else {
      const int kind = kinds.kind[static_cast<int>((i - exist_words) / num_entries)];
      if (v == static_cast<unsigned long long>(acc_identity(kind))) continue;
      global_accumulate(&dst[i], v, kind);
    }
  }
}

// ---------------------------------------------------------------------------
// growth of the hash-strategy table: spill log + published control words
// ---------------------------------------------------------------------------
// Records [first, first + count) of a spill log back to "unused": every column word holds its identity (the record's
// owner accumulates into it without initialising anything).  count = *count_dev when that is given (clear: only the
// records a previous run touched are rewritten).
__global__ __launch_bounds__(kABlock) void init_log_kernel(unsigned long long *__restrict__ log, int stride,
                                                          unsigned long long count, const unsigned int *__restrict__ count_dev,
                                                          unsigned int log_cap, ColKinds kinds) {
  if (count_dev != nullptr) count = *count_dev < log_cap ? *count_dev : log_cap;
  const unsigned long long words = count * stride;
  for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * kABlock + threadIdx.x; i < words;
       i += static_cast<unsigned long long>(gridDim.x) * kABlock) {
    const int w = static_cast<int>(i % stride);
    log[i] = w == 0 ? kEmptyCode : static_cast<unsigned long long>(acc_identity(kinds.kind[w - 1]));
  }
}

// Fold spill records into the (grown) table; a record that still finds no slot is appended to `g`'s own log.
__global__ __launch_bounds__(kABlock) void drain_log_kernel(const unsigned long long *__restrict__ records, unsigned int count,
                                                           int stride, int num_cols, ColKinds kinds, HashTableView g) {
  for (unsigned int r = blockIdx.x * kABlock + threadIdx.x; r < count; r += gridDim.x * kABlock) {
    const unsigned long long *rec = records + static_cast<unsigned long long>(r) * stride;
    const unsigned long long gs = global_find_or_insert(g, rec[0]);
    if (gs == ~0ull) continue;
    for (int col = 0; col < num_cols; ++col) global_add(g, col, gs, rec[1 + col], kinds.kind[col]);
  }
}

// One lane copies the control words to host-visible memory behind an update: {groups, spilled records, overflow flag,
// sequence}.  The host reads them without synchronising (maybe_grow), so a table that is filling up grows between two
// update calls instead of failing at finalize.
__global__ void publish_control_kernel(const unsigned long long *__restrict__ control, unsigned long long *host_slot,
                                       unsigned long long seq) {
  if (threadIdx.x != 0) return;
  __hip_atomic_store(&host_slot[0], control[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&host_slot[1], control[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&host_slot[2], control[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&host_slot[3], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------
// finalize
// ---------------------------------------------------------------------------

__device__ __forceinline__ void write_values(const FinalizeDesc &f, const unsigned long long *states,
                                             unsigned long long col_stride, unsigned long long idx,
                                             int count_col, bool empty_group, long long out_row) {
  const long long cnt = count_col >= 0 ? static_cast<long long>(states[static_cast<unsigned long long>(count_col) * col_stride + idx]) : 0;
  for (int a = 0; a < f.num_aggs; ++a) {
    bool is_null = false;
    // rows this aggregate saw: those with a non-NULL argument when the argument is nullable, else every row of the group
    const bool has_nn = f.nn_col[a] >= 0;
    const long long seen = has_nn ? static_cast<long long>(states[static_cast<unsigned long long>(f.nn_col[a]) * col_stride + idx]) : cnt;
    if (f.fn[a] == QSX_AGG_COUNT_STAR) {
      static_cast<long long *>(f.out_vals[a])[out_row] = cnt;
    } else if (f.fn[a] == QSX_AGG_COUNT) {
      static_cast<long long *>(f.out_vals[a])[out_row] = seen;
    } else {
      const unsigned long long raw = states[static_cast<unsigned long long>(f.sum_col[a]) * col_stride + idx];
      if (f.fn[a] == QSX_AGG_SUM) {
        // int64 and double results are both stored as their 8 raw bytes; NULL when no non-NULL argument was seen
        // (AggregationHandleSum.cpp:100-120)
        is_null = empty_group || (has_nn && seen == 0);
        static_cast<unsigned long long *>(f.out_vals[a])[out_row] = is_null ? 0ull : raw;
      } else if (f.fn[a] == QSX_AGG_MIN || f.fn[a] == QSX_AGG_MAX) {
        // typed like the argument; the accumulator is the int value or the order-mapped double.  A dense-table key
        // that only has its existence bit (BuildAggregationExistenceMapOperator) saw no value: NULL.
        is_null = empty_group || (has_nn ? seen == 0 : (count_col >= 0 && cnt == 0));
        const long long word = static_cast<long long>(raw);
        switch (f.val_type[a]) {
          case QSX_INT: static_cast<int32_t *>(f.out_vals[a])[out_row] = is_null ? 0 : static_cast<int32_t>(word); break;
          case QSX_LONG: static_cast<long long *>(f.out_vals[a])[out_row] = is_null ? 0 : word; break;
          case QSX_FLOAT:
            static_cast<float *>(f.out_vals[a])[out_row] = is_null ? 0.0f : static_cast<float>(__longlong_as_double(ordered_from_bits(word)));
            break;
          default:
            static_cast<double *>(f.out_vals[a])[out_row] = is_null ? 0.0 : __longlong_as_double(ordered_from_bits(word));
            break;
        }
      } else {
        const double sum = f.is_int[a] ? static_cast<double>(static_cast<long long>(raw))
                                       : __longlong_as_double(static_cast<long long>(raw));
        is_null = seen == 0;
        static_cast<double *>(f.out_vals[a])[out_row] = is_null ? 0.0 : sum / static_cast<double>(seen);
      }
    }
    if (f.out_nulls[a] != nullptr) f.out_nulls[a][out_row] = is_null ? 1 : 0;
  }
}

// Key word `w` of the group in slot i: the code itself, or for a wide key the MIN column of that word — which must
// equal its MAX column, or two different keys met under one hash (DevConfig::wide_words).
__device__ __forceinline__ unsigned long long group_key_word(const FinalizeDesc &f, unsigned long long code,
                                                             const unsigned long long *states, unsigned long long stride,
                                                             unsigned long long i, int w) {
  if (f.wide_words == 0) return code;
  const unsigned long long lo = states[static_cast<unsigned long long>(f.wide_min_col[w]) * stride + i];
  const unsigned long long hi = states[static_cast<unsigned long long>(f.wide_max_col[w]) * stride + i];
  if (lo != hi) atomicExch(f.collision, 1);
  return lo;
}

__device__ __forceinline__ void write_keys_from_code(const FinalizeDesc &f, unsigned long long code, long long out_row,
                                                     const unsigned long long *states = nullptr, unsigned long long stride = 0,
                                                     unsigned long long slot = 0) {
  for (int k = 0; k < f.num_keys; ++k) {
    const unsigned long long v = group_key_word(f, code, states, stride, slot, f.key_word[k]) >> f.key_shift[k];
    switch (f.key_width[k]) {
      case 1: static_cast<uint8_t *>(f.out_keys[k])[out_row] = static_cast<uint8_t>(v); break;
      case 2: static_cast<uint16_t *>(f.out_keys[k])[out_row] = static_cast<uint16_t>(v); break;
      case 4: static_cast<uint32_t *>(f.out_keys[k])[out_row] = static_cast<uint32_t>(v); break;
      default: static_cast<unsigned long long *>(f.out_keys[k])[out_row] = v; break;
    }
  }
}

// Reference hash of one key component (types/TypedValue.hpp:575-592): the
// zero-extended bit pattern, -0.0 canonicalised for FLOAT/DOUBLE.
__device__ __forceinline__ unsigned long long reference_scalar_hash(int type, unsigned long long bits) {
  switch (type) {
    case QSX_INT: return bits & 0xFFFFFFFFull;
    case QSX_FLOAT: return (bits & 0xFFFFFFFFull) == 0x80000000ull ? 0ull : (bits & 0xFFFFFFFFull);
    case QSX_DOUBLE: return bits == 0x8000000000000000ull ? 0ull : bits;
    default: return bits;
  }
}
// CombineHashes (utility/HashPair.hpp:47-58).
__device__ __forceinline__ unsigned long long reference_combine(unsigned long long a, unsigned long long b) {
  const unsigned long long kMul = 0x9ddfea08eb382d69ull;
  unsigned long long x = (a ^ b) * kMul;
  x ^= (x >> 47);
  unsigned long long y = (b ^ x) * kMul;
  y ^= (y >> 47);
  y *= kMul;
  return y;
}

// One output row per occupied slot whose reference partition matches.
__global__ __launch_bounds__(kABlock) void finalize_hash_kernel(HashTableView g, FinalizeDesc f,
                                                               int partition, int num_partitions,
                                                               int partition_by_hash, long long capacity,
                                                               unsigned long long *__restrict__ out_groups) {
  const unsigned long long stride = g.cap + 1;
  const unsigned long long total = g.cap + 1;
  const unsigned long long rounded = (total + kWave - 1) / kWave * kWave;
  for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * kABlock + threadIdx.x;
       i < rounded; i += static_cast<unsigned long long>(gridDim.x) * kABlock) {
    bool emit = false;
    unsigned long long code = kEmptyCode;
    if (i < total) {
      code = i == g.cap ? kEmptyCode : g.keys[i];
      const unsigned long long cnt = g.states[i];
      emit = (i == g.cap) ? (cnt != 0) : (code != kEmptyCode);
      if (emit && partition_by_hash) {
        // partitioned aggregation routes a group to HashCompositeKey % P
        // (storage/AggregationOperationState.cpp:576-583, utility/CompositeHash.hpp:39-48)
        unsigned long long h = 0;
        for (int k = 0; k < f.num_keys; ++k) {
          unsigned long long bits = group_key_word(f, code, g.states, stride, i, f.key_word[k]) >> f.key_shift[k];
          if (f.key_width[k] < 8) bits &= (1ull << (8 * f.key_width[k])) - 1;
          const unsigned long long hk = reference_scalar_hash(f.key_type[k], bits);
          h = k == 0 ? hk : reference_combine(h, hk);
        }
        emit = static_cast<int>(h % static_cast<unsigned long long>(num_partitions)) == partition;
      }
    }
    const uint64_t m = __ballot(emit);
    if (m == 0) continue;
    const int leader = __ffsll(static_cast<long long>(m)) - 1;
    unsigned long long base = 0;
    if (lane_id() == leader) base = atomicAdd(out_groups, static_cast<unsigned long long>(__popcll(m)));
    base = __shfl(base, leader, kWave);
    if (!emit) continue;
    const long long out_row = static_cast<long long>(base) + rank_below(m);
    if (out_row >= capacity) continue;
    write_keys_from_code(f, code, out_row, g.states, stride, i);
    write_values(f, g.states, stride, i, 0, false, out_row);
  }
}

// ---- K11 standalone: a scalar expression projected into a column (qsx_eval_expression) --------------------------------
// The expression program of the aggregation kernel over global stripes instead of a staged tile: kExprRows rows per
// thread, every node rounded on its own (-ffp-contract=off) like the reference's materialised temp vectors.
constexpr int kExprRows = 2;
struct ExprProgram {
  const void *cols[QSX_MAX_COLUMNS];
  int types[QSX_MAX_COLUMNS];
  int num_instrs;
  DevInstr instrs[QSX_MAX_INSTRS];
  double consts[QSX_MAX_CONSTS];
  DevOperand result;
};
static_assert(sizeof(ExprProgram) % 4 == 0, "store_struct_kernel copies words");
__device__ __forceinline__ void expr_operand(const ExprProgram &p, const DevOperand &o, const Temps<kExprRows> &t, const int64_t (&row)[kExprRows],
                                             double (&out)[kExprRows]) {
  switch (o.kind) {
    case QSX_OPD_COLUMN: {
      const void *col = as_global(p.cols[o.index]);   // (loaded from the program in device memory)
#pragma unroll
      for (int v = 0; v < kExprRows; ++v) {
        switch (p.types[o.index]) {
          case QSX_INT: out[v] = static_cast<double>(static_cast<const int32_t *>(col)[row[v]]); break;
          case QSX_LONG: out[v] = static_cast<double>(static_cast<const long long *>(col)[row[v]]); break;
          case QSX_FLOAT: out[v] = static_cast<double>(static_cast<const float *>(col)[row[v]]); break;
          default: out[v] = static_cast<const double *>(col)[row[v]]; break;
        }
      }
      break;
    }
    case QSX_OPD_CONST:
#pragma unroll
      for (int v = 0; v < kExprRows; ++v) out[v] = p.consts[o.index];
      break;
    default:
      temps_get<kExprRows>(t, o.index, out);
      break;
  }
}
__global__ __launch_bounds__(kABlock) void eval_expression_kernel(const ExprProgram *__restrict__ program, int64_t n, double *__restrict__ out) {
  const ExprProgram &p = *program;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kABlock;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * kABlock + threadIdx.x; base < n; base += stride * kExprRows) {
    int64_t row[kExprRows];
#pragma unroll
    for (int v = 0; v < kExprRows; ++v) row[v] = base + v * stride < n ? base + v * stride : n - 1;   // clamped, stores are guarded
    Temps<kExprRows> temps;
    for (int k = 0; k < p.num_instrs; ++k) {
      const DevInstr in = p.instrs[k];
      double a[kExprRows], b[kExprRows], r[kExprRows];
      expr_operand(p, in.a, temps, row, a);
      expr_operand(p, in.b, temps, row, b);
#pragma unroll
      for (int v = 0; v < kExprRows; ++v) {
        switch (in.op) {
          case QSX_EX_ADD: r[v] = a[v] + b[v]; break;
          case QSX_EX_SUB: r[v] = a[v] - b[v]; break;
          case QSX_EX_MUL: r[v] = a[v] * b[v]; break;
          default: r[v] = a[v] / b[v]; break;
        }
      }
      temps_set<kExprRows>(temps, in.dst, r);
    }
    double value[kExprRows];
    expr_operand(p, p.result, temps, row, value);
#pragma unroll
    for (int v = 0; v < kExprRows; ++v) {
      if (base + v * stride < n) out[base + v * stride] = value[v];
    }
  }
}

// The same program over INT / LONG operands in integer arithmetic (ArithmeticBinaryOperators.hpp:203-340 instantiated for
// integer types: C++ +, -, *, / on the values; INT op INT is an INT, anything with a LONG a LONG).  Evaluated in 64 bits and
// narrowed at the store: the low 32 bits of a 64-bit +, -, * are those of the 32-bit operation.  x / 0 gives 0 (the
// reference's division by zero is undefined behaviour).
struct IntExprProgram {
  const void *cols[QSX_MAX_COLUMNS];
  int32_t types[QSX_MAX_COLUMNS];
  DevInstr instrs[QSX_MAX_INSTRS];
  long long consts[QSX_MAX_CONSTS];
  DevOperand result;
  int32_t num_instrs;
  int32_t narrow[QSX_MAX_INSTRS];   // instruction k produces an INT (both operands INT): wrap to 32 bits like the reference's int arithmetic
};
static_assert(sizeof(IntExprProgram) % 4 == 0, "store_struct_kernel copies words");
__global__ __launch_bounds__(kABlock) void eval_expression_long_kernel(const IntExprProgram *__restrict__ program, int64_t n, int out_width,
                                                                      void *__restrict__ out) {
  const IntExprProgram &p = *program;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * kABlock + threadIdx.x; row < n; row += static_cast<int64_t>(gridDim.x) * kABlock) {
    long long temps[QSX_MAX_TEMPS];
    auto operand = [&](const DevOperand &o) -> long long {
      switch (o.kind) {
        case QSX_OPD_COLUMN:
          return p.types[o.index] == QSX_INT ? static_cast<long long>(static_cast<const int32_t *>(as_global(p.cols[o.index]))[row])
                                             : static_cast<const long long *>(as_global(p.cols[o.index]))[row];
        case QSX_OPD_CONST: return p.consts[o.index];
        default: return temps[o.index];
      }
    };
    for (int k = 0; k < p.num_instrs; ++k) {
      const DevInstr in = p.instrs[k];
      const long long a = operand(in.a), b = operand(in.b);
      long long r;
      switch (in.op) {
        case QSX_EX_ADD: r = static_cast<long long>(static_cast<unsigned long long>(a) + static_cast<unsigned long long>(b)); break;
        case QSX_EX_SUB: r = static_cast<long long>(static_cast<unsigned long long>(a) - static_cast<unsigned long long>(b)); break;
        case QSX_EX_MUL: r = static_cast<long long>(static_cast<unsigned long long>(a) * static_cast<unsigned long long>(b)); break;
        default: r = b == 0 ? 0 : (b == -1 ? static_cast<long long>(0ull - static_cast<unsigned long long>(a)) : a / b); break;
      }
      if (p.narrow[k]) r = static_cast<long long>(static_cast<int32_t>(r));
      temps[in.dst] = r;
    }
    const long long value = operand(p.result);
    if (out_width == 4) static_cast<int32_t *>(out)[row] = static_cast<int32_t>(value);
    else static_cast<long long *>(out)[row] = value;
  }
}

// A wide-key state whose finalize saw MIN != MAX in a key word reports it in the group count (include/qsx.h).
__global__ void report_collision_kernel(const int *collision, unsigned long long *out_groups) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && *collision != 0) *out_groups = static_cast<unsigned long long>(QSX_GROUPS_HASH_COLLISION);
}

// SINGLE_STATE: exactly one row, NULL sums when no row was aggregated
// (storage/AggregationOperationState.cpp:652-670).
__global__ void finalize_single_kernel(HashTableView g, FinalizeDesc f, unsigned long long *out_groups) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  // the only possible code is 0
  unsigned long long s = code_slot(0ull, g.shift);
  unsigned long long found = ~0ull;
  for (unsigned long long probes = 0; probes < g.cap; ++probes) {
    const unsigned long long k = g.keys[s];
    if (k == 0ull) { found = s; break; }
    if (k == kEmptyCode) break;
    s = (s + 1) & (g.cap - 1);
  }
  if (found == ~0ull) {
    // nothing aggregated: COUNT = 0, SUM/AVG NULL. Slot `cap` is never used by
    // code 0 and holds all-zero state words.
    write_values(f, g.states, g.cap + 1, g.cap, 0, true, 0);
  } else {
    write_values(f, g.states, g.cap + 1, found, 0, false, 0);
  }
  *out_groups = 1;
}

// COLLISION_FREE finalize: ordered compaction of the existence bits of the
// partition's key range [begin, end) (CollisionFreeVectorTable.hpp:192-208, 647-727).
constexpr int kDenseTileWords = 64;

__device__ __forceinline__ unsigned long long ranged_word(const unsigned long long *exist, long long w,
                                                          long long begin, long long end) {
  // bits of word w restricted to keys in [begin, end)
  const long long lo = w * 64, hi = lo + 64;
  if (hi <= begin || lo >= end) return 0;
  unsigned long long v = exist[w];
  if (begin > lo) v &= ~0ull << (begin - lo);
  if (end < hi) v &= ~0ull >> (hi - end);
  return v;
}

__global__ __launch_bounds__(kABlock) void dense_tile_count_kernel(const unsigned long long *__restrict__ exist,
                                                                  long long first_word, long long num_words,
                                                                  long long begin, long long end,
                                                                  long long num_tiles,
                                                                  int32_t *__restrict__ tile_counts) {
  const int lane = lane_id();
  for (long long tile = static_cast<long long>(blockIdx.x) * (kABlock / kWave) + (threadIdx.x >> 6);
       tile < num_tiles; tile += static_cast<long long>(gridDim.x) * (kABlock / kWave)) {
    const long long w = tile * kDenseTileWords + lane;
    int c = w < num_words ? __popcll(ranged_word(exist, first_word + w, begin, end)) : 0;
    c = wave_reduce_add(c);
    if (lane == 0) tile_counts[tile] = c;
  }
}

__global__ __launch_bounds__(kABlock) void finalize_dense_kernel(DenseView d, FinalizeDesc f,
                                                                long long first_word, long long num_words,
                                                                long long begin, long long end,
                                                                long long num_tiles,
                                                                const int64_t *__restrict__ tile_offsets,
                                                                long long capacity) {
  // Positions (0..4095 inside the tile) of the tile's existing keys, in key order.  Each lane expands the set bits of
  // its word at its prefix offset; the wave then handles 64 existing keys per step with every lane busy — a sparse
  // existence map (Q3: 8 % of the order keys) used to take 64 steps of ~5 active lanes per tile.
  __shared__ uint16_t s_pos[kABlock / kWave][kDenseTileWords * 64];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  for (long long tile = static_cast<long long>(blockIdx.x) * (kABlock / kWave) + wave; tile < num_tiles;
       tile += static_cast<long long>(gridDim.x) * (kABlock / kWave)) {
    const long long w = tile * kDenseTileWords + lane;
    unsigned long long mine = w < num_words ? ranged_word(d.exist, first_word + w, begin, end) : 0;
    const int pc = __popcll(mine);
    int incl = pc;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int up = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += up;
    }
    const int total = __shfl(incl, kWave - 1, kWave);
    int at = incl - pc;
    while (mine != 0) {
      const int bit = __ffsll(static_cast<long long>(mine)) - 1;
      s_pos[wave][at++] = static_cast<uint16_t>(lane * 64 + bit);
      mine &= mine - 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const long long tile_off = tile_offsets[tile];
    const long long tile_loc0 = (first_word + tile * kDenseTileWords) * 64;
    for (int i = lane; i < total; i += kWave) {
      const long long out_row = tile_off + i;
      const long long loc = tile_loc0 + s_pos[wave][i];
      if (out_row < capacity) {
        if (f.key_width[0] == 4) static_cast<int32_t *>(f.out_keys[0])[out_row] = static_cast<int32_t>(loc);
        else static_cast<long long *>(f.out_keys[0])[out_row] = loc;
        write_values(f, d.states, static_cast<unsigned long long>(d.num_entries),
                     static_cast<unsigned long long>(loc), d.has_count ? 0 : -1, false, out_row);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// BuildAggregationExistenceMapWorkOrder::execute (relational_operators/BuildAggregationExistenceMapOperator.cpp:50-67):
// existence_map->setBit(key) for every (selected) row.  Most keys of a block hit distinct words, so the bit is tested
// before the atomic; keys outside [0, num_entries) are skipped (the reference would write out of bounds).
template <typename KeyT>
__global__ __launch_bounds__(kABlock) void mark_existence_kernel(const KeyT *__restrict__ keys, int64_t n,
                                                                const uint64_t *__restrict__ filter,
                                                                unsigned long long *__restrict__ exist, long long num_entries) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kABlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kABlock) {
    if (filter != nullptr && !msb_bit(filter[i >> 6], static_cast<int>(i & 63))) continue;
    const long long loc = static_cast<long long>(keys[i]);
    if (loc < 0 || loc >= num_entries) continue;
    const unsigned long long bit = 1ull << (loc & 63);
    unsigned long long *word = &exist[loc >> 6];
    if ((__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) == 0) atomicOr(word, bit);
  }
}

// How clustered on the key do the rows arrive?  Counts key[i] == key[i - 1] over the first rows of an input.
template <typename KeyT>
__global__ __launch_bounds__(1024) void adjacent_equal_kernel(const KeyT *__restrict__ keys, int rows, unsigned int *__restrict__ out) {
  unsigned int equal = 0;
  for (int i = 1 + static_cast<int>(threadIdx.x); i < rows; i += 1024) equal += keys[i] == keys[i - 1] ? 1u : 0u;
  equal = static_cast<unsigned int>(wave_reduce_add(static_cast<unsigned long long>(equal)));
  if (lane_id() == 0 && equal != 0) atomicAdd(out, equal);
}

struct DictTable {
  const void *p[QSX_MAX_COLUMNS];
  int entries[QSX_MAX_COLUMNS];   // entries of p[c] when the caller said so (qsx_agg_update_coded_sized), else 0; read by the plan
                                  // shapes right behind the pointers (agg_hash_update.hpp dict_entries_behind)
};
// The dictionary sizes of the call in progress on this thread (qsx_agg_update_coded_sized sets them around agg_update, which
// enqueues its launches before it returns; the launchers copy them into the call's DictTable).
static thread_local const int32_t *tl_dictionary_entries = nullptr;
struct NullTable {   // the null bitmaps of one call by null slot (DevConfig::null_column), for the run-time plan shapes
  const unsigned long long *p[QSX_MAX_COLUMNS];
};
__global__ __launch_bounds__(kABlock) void popcount_words_kernel(const unsigned long long *__restrict__ words,
                                                                long long num_words,
                                                                unsigned long long *__restrict__ out) {
  unsigned long long c = 0;
  for (long long w = static_cast<long long>(blockIdx.x) * kABlock + threadIdx.x; w < num_words;
       w += static_cast<long long>(gridDim.x) * kABlock) {
    c += __popcll(words[w]);
  }
  // one atomic per workgroup: same-address atomics on the counter take ~12 ns each, one per wave of a 2 K-block grid
  // was 100 us of a 110 us kernel
  __shared__ unsigned long long s_part[kABlock / kWave];
  c = wave_reduce_add(c);
  if (lane_id() == 0) s_part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long all = 0;
    for (int w = 0; w < kABlock / kWave; ++w) all += s_part[w];
    if (all != 0) atomicAdd(out, all);
  }
}

}  // namespace qsx

using namespace qsx;

// ===========================================================================
// host side
// ===========================================================================
// Spill-log records per growable state: what one update call may leave outside the table before the host grows it
// (a 0.5 M-row block whose groups the estimate missed entirely still fits).
constexpr unsigned int kLogRecords = 1u << 20;

constexpr int kJitVariants = 10;   // (filter) x (tile path, partitioned path, group directory, run of blocks, directory over a run)
constexpr int kDirBoundSlots = 32;
constexpr size_t kDirControlBytes = 16 + sizeof(unsigned long long) * 2 * QSX_MAX_KEYS * kDirBoundSlots;
struct qsx_agg_state {
  qsx_agg_config_t config;
  bool has_coded_columns = false;   // some column arrives as codes of a compressed attribute
  FactoredStatic factored;          // the plan seen through its dictionary columns (agg_factored.hpp); ok = it factors
  bool has_date_key = false;        // a group-by key is a DATE: its padding bytes are masked when the key is packed
  // run-time plan shapes asked for, owned by the cache: index = (filter ? 1 : 0) + (partitioned input ? 2 : 0)
  JitRequest *jit_request[kJitVariants] = {};
  JitGeometry jit_geometry[kJitVariants] = {};
  size_t jit_lds[kJitVariants] = {};
  DevConfig dev;            // everything but cols[]
  FinalizeDesc fin;         // everything but the output pointers
  int num_sums = 0;
  int num_cols = 0;         // state columns in the image
  unsigned int int_col_mask = 0;
  ColKinds col_kinds{};
  bool has_min_max = false;  // some state column needs a non-zero identity
  bool dense = false;
  bool dense_has_count = false;
  // A dense state too large for one workgroup's LDS: 0 = not decided, 1 = key-range families of workgroups in LDS (every
  // family reads every row), 2 = one global atomic per run of equal adjacent keys — decided once, from how clustered the
  // first rows the state sees are (decide_dense_families).
  std::atomic<int> dense_families{0};

  // one allocation: the image
  unsigned long long *image = nullptr;
  size_t image_bytes = 0;
  unsigned long long cap = 0;      // hash strategies
  long long exist_words = 0;       // dense
  // control words: [0] ngroups (u64), [1] overflow|error flag (int), [2] scratch counter, [3] spilled records (u32)
  unsigned long long *control = nullptr;
  // ---- growth (COMPACT_KEY / GENERIC) ----
  // Update / merge launches hold `table_mutex` shared (they read image, cap, geometry); growth holds it exclusive after
  // draining the device — the role of the reference's resize lock (storage/HashTable.hpp:1215 for the join table,
  // PackedPayloadHashTable::resize for this one).
  bool growable = false;
  mutable std::shared_mutex table_mutex;
  unsigned long long *log = nullptr;          // spill log, kLogRecords records of (num_cols + 1) words
  unsigned long long *published = nullptr;    // host-visible copy of {ngroups, spilled, overflow, seq} (publish_control_kernel)
  unsigned long long *published_dev = nullptr;
  std::atomic<unsigned long long> publish_seq{0};
  int64_t geometry_est = 1;                   // the group count the LDS geometry below was derived for
  // Group directory (agg_common.hpp DirView): mid-size group counts whose accumulators fit one CU's LDS when addressed by
  // a dense group number.  dir_gids == 0: not in use.
  int dir_gids = 0;
  int dir_nbuf = 2;
  unsigned long long *dir_entries = nullptr;
  unsigned long long dir_cap = 0;
  unsigned long long *dir_codes = nullptr;
  int dir_codes_len = 0;
  unsigned long long *dir_words = nullptr;    // wide keys: the key words of every gid (DirView::words_by_gid)
  int dir_entry_words = 2;                    // words per directory entry the allocation was made for
  unsigned int *dir_ngids = nullptr;          // [4] counter words, then the key bounds (kDirControlBytes in all)
  std::atomic<unsigned> dir_calls{0};         // rotates the build pass's sample
  // The key bounds belong to ONE update call (build pass writes, accumulate pass reads): calls on the same stream are
  // ordered, calls on different streams (worker threads) are not, so every stream gets its own bounds words.
  std::mutex dir_mutex;
  std::vector<hipStream_t> dir_streams;
  int dir_bounds_slot(hipStream_t s) {
    std::lock_guard<std::mutex> lock(dir_mutex);
    for (size_t i = 0; i < dir_streams.size(); ++i) {
      if (dir_streams[i] == s) return static_cast<int>(i);
    }
    if (dir_streams.size() >= static_cast<size_t>(kDirBoundSlots)) return -1;   // that stream's calls take the other paths
    dir_streams.push_back(s);
    return static_cast<int>(dir_streams.size()) - 1;
  }
  DirView dir_view(int bounds_slot = 0) const {
    DirView d;
    d.entries = dir_entries;
    d.dmask = dir_cap - 1;
    int log2 = 0;
    while ((1ull << log2) < dir_cap) ++log2;
    d.dshift = 64 - log2;
    d.codes_by_gid = dir_codes;
    d.ngids = dir_ngids;
    d.lds_gids = static_cast<unsigned int>(dir_gids);
    d.bounds = reinterpret_cast<unsigned long long *>(dir_ngids + 4) + 2 * QSX_MAX_KEYS * bounds_slot;   // same allocation, 16 bytes in
    d.sample_stride = 1;
    d.sample_phase = 0;
    d.build_step = 0;
    d.wide_words = dev.wide_words;
    d.words_by_gid = dir_words;
    return d;
  }
  // scratch for ordered dense finalize
  int32_t *tile_counts = nullptr;
  int64_t *tile_offsets = nullptr;
  long long max_tiles = 0;
  int lds_slots = 64;
  int lds_ranges = 1;  // > 1: hash-range families of workgroups (agg_hash_update.hpp)
  // More groups than one workgroup-private LDS table holds: big inputs are first hash-partitioned on the
  // key code (K9, mixing-hash mode) into part_count pieces whose groups fit a part_slots-slot table, then
  // aggregated piece by piece — one read + one write + one read of the used columns instead of
  // lds_ranges reads of the whole input at one workgroup per CU.
  int part_count = 1;
  int part_slots = 64;
  unsigned used_columns = 0;
  const struct ShapeEntry *shape = nullptr;  // AOT plan shape matching this configuration, if any
  // ... or the member of the AOT family (agg_family.hpp) whose canonical configuration is this plan up to the numbering of its
  // columns: family_cols[c] = the state's column behind canonical column c
  std::atomic<int> two_level_mode{0};   // the two-level partitioned aggregation for this state's keys: 0 = not decided, 1 = yes, 2 = the one-pass path (decide_two_level)
  const FamilyEntry *family = nullptr;
  int family_num_columns = 0;
  int family_cols[QSX_MAX_COLUMNS] = {};
  // Run-time plan shapes (agg_jit.hpp), one per filter variant; compiled once the state has seen enough
  // rows to pay for the 1-2 s of hipRTC.
  std::mutex jit_mutex;
  const JitKernel *jit[kJitVariants] = {};
  bool jit_tried[kJitVariants] = {};
  int jit_tile_bytes[kJitVariants] = {};
  std::atomic<long long> rows_seen{0};

  HashTableView hash_view() const {
    HashTableView g;
    g.keys = image;
    g.states = image + (cap + 1);
    g.cap = cap;
    int log2 = 0;
    while ((1ull << log2) < cap) ++log2;
    g.shift = 64 - log2;
    g.ngroups = control;
    g.overflow = reinterpret_cast<int *>(control + 1);
    g.log = log;
    g.log_count = reinterpret_cast<unsigned int *>(control + 3);
    g.log_cap = log != nullptr ? kLogRecords : 0u;
    g.log_stride = num_cols + 1;
    return g;
  }
  DenseView dense_view() const {
    DenseView d;
    d.exist = image;
    d.states = image + exist_words;
    d.num_entries = config.num_entries;
    d.has_count = dense_has_count ? 1 : 0;
    d.error = reinterpret_cast<int *>(control + 1);
    return d;
  }
};

// Host-visible control words of growable states: 32 bytes of pinned, mapped memory each.  hipHostMalloc / hipHostFree take
// 0.1-0.2 ms apiece — on the critical path of every query that creates and destroys an aggregation state — so freed slots
// are kept (a page holds 128 of them; pages are never returned).
struct PublishedSlots {
  std::mutex mutex;
  std::vector<unsigned long long *> free_slots;
  unsigned long long *take() {
    std::lock_guard<std::mutex> lock(mutex);
    if (free_slots.empty()) {
      constexpr size_t kSlotsPerPage = 128;
      unsigned long long *page = nullptr;
      if (hipHostMalloc(reinterpret_cast<void **>(&page), kSlotsPerPage * 4 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
      }
      for (size_t i = 0; i < kSlotsPerPage; ++i) free_slots.push_back(page + 4 * i);
    }
    unsigned long long *slot = free_slots.back();
    free_slots.pop_back();
    return slot;
  }
  void give(unsigned long long *slot) {
    if (slot == nullptr) return;
    std::lock_guard<std::mutex> lock(mutex);
    free_slots.push_back(slot);
  }
};
static PublishedSlots &published_slots() {
  static PublishedSlots *slots = new PublishedSlots;   // (never destroyed: states may outlive static destruction order)
  return *slots;
}

// Fills the state from the constexpr-capable translation shared with the AOT plan shapes.
static int translate_config(const qsx_agg_config_t &c, qsx_agg_state *st) {
  const Translated t = translate(c);
  if (t.status != QSX_OK) return t.status;
  st->dev = t.dev;
  if (const char *e = getenv("QSX_AGG_WIDE_HASH_BITS")) {   // test hook: a short hash makes different wide keys collide
    const int bits = atoi(e);
    if (bits > 0 && bits < 64) st->dev.wide_hash_mask = (1ull << bits) - 1;
  }
  st->fin = t.fin;
  st->num_sums = t.num_sums;
  st->num_cols = t.num_cols;
  st->int_col_mask = t.int_col_mask;
  for (int col = 0; col < t.num_cols; ++col) {
    st->col_kinds.kind[col] = t.col_kind[col];
    if (t.col_kind[col] >= kAccMinI64) st->has_min_max = true;
  }
  st->used_columns = t.used_columns;
  for (int i = 0; i < t.dev.num_columns; ++i) st->has_coded_columns = st->has_coded_columns || t.dev.code_width[i] != 0;
  for (int k = 0; k < t.dev.num_keys; ++k) st->has_date_key = st->has_date_key || t.dev.column_type[t.dev.key_column[k]] == QSX_DATE;
  st->dense = t.dense;
  st->dense_has_count = t.dense_has_count;
  if (st->has_coded_columns) st->factored = factored_analyse(t.dev, t.dense);
  return QSX_OK;
}

static size_t align16(size_t v) { return (v + 15) & ~static_cast<size_t>(15); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device): one flag per device behind a lock —
// AggregationWorkOrders of several worker threads reach the launchers concurrently.
constexpr int kMaxDevices = 64;
struct PerDeviceOnce {
  std::mutex mutex;
  bool done[kMaxDevices] = {};
};
template <typename F>
static int once_per_device(PerDeviceOnce &flags, F &&set_attribute) {
  int device = 0;
  QSX_HIP_TRY(hipGetDevice(&device));
  if (device < 0 || device >= kMaxDevices) return QSX_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> lock(flags.mutex);
  if (flags.done[device]) return QSX_OK;
  QSX_HIP_TRY(set_attribute());
  flags.done[device] = true;
  return QSX_OK;
}

// After the zeroing memset: MIN / MAX columns start from their identity.
static int fill_identities(qsx_agg_state *st, hipStream_t stream) {
  if (!st->has_min_max) return QSX_OK;
  const long long words = st->dense ? st->config.num_entries : static_cast<long long>(st->cap + 1);
  unsigned long long *states = st->dense ? st->image + st->exist_words : st->image + (st->cap + 1);
  hipLaunchKernelGGL(fill_identity_kernel, dim3(grid_for(words, kABlock * 4)), dim3(kABlock), 0, stream, states, words, words,
                     st->num_cols, st->col_kinds);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

// Launch geometry of the hash-strategy update kernel; the environment overrides exist for
// tuning sweeps on the GPU box (tools/agg_probe.py), the defaults are the measured best.
struct AggTuning {
  int rows_per_thread;   // V of the interpreter kernel: 2 or 4 (tile = 256 * V rows)
  int shape_rows_per_thread;  // V of the AOT plan-shape kernels
  int buffers;           // 1 or 2 tile buffers per workgroup
  int acc_kib;           // LDS budget of the replicated accumulators
  int max_blocks_per_cu;
  int jit_waves_per_eu;  // run-time shapes whose LDS admits >= 5 workgroups per CU are built for this many waves per SIMD (0: do not ask)
};
// Small hash tables of plan shapes: per-wave register accumulators (agg_hash_update.hpp, REG).  Measured slower than the LDS
// atomics they replace: off unless QSX_AGG_REG_GROUPS=1 (read per call: the parity tests run both).
// Rows per thread of the group-directory plan shapes (2: one 2048-row tile instead of two 1024-row buffers); QSX_AGG_DIR_ROWS=1
// keeps the one-row form.
static int dir_rows_per_thread() {
  const char *e = getenv("QSX_AGG_DIR_ROWS");
  return e != nullptr && atoi(e) == 1 ? 1 : 2;
}
static bool reg_groups_enabled() {
  const char *e = getenv("QSX_AGG_REG_GROUPS");
  return e != nullptr && atoi(e) != 0;
}
static const AggTuning &agg_tuning() {
  static AggTuning t = []() {
    // measured on MI355X (tools/agg_sweep.sh, 600 M Q1 rows): shape kernel V=4, 1 buffer, 16 KiB
    // accumulators, 4 workgroups/CU = 3.47 ms; interpreter V=2 = 10.9 ms
    AggTuning v{2, 4, 1, 16, 4, 5};
    if (const char *e = getenv("QSX_AGG_JIT_WAVES")) v.jit_waves_per_eu = atoi(e) >= 0 && atoi(e) <= 8 ? atoi(e) : 5;
    if (const char *e = getenv("QSX_AGG_ROWS_PER_THREAD")) v.rows_per_thread = v.shape_rows_per_thread = atoi(e) == 4 ? 4 : 2;
    if (const char *e = getenv("QSX_AGG_BUFFERS")) v.buffers = atoi(e) == 2 ? 2 : (atoi(e) == 0 ? 0 : 1);   // (0: per shape, jit_geometry_for)
    if (const char *e = getenv("QSX_AGG_ACC_KIB")) v.acc_kib = atoi(e) > 0 ? atoi(e) : 12;
    if (const char *e = getenv("QSX_AGG_BLOCKS_PER_CU")) v.max_blocks_per_cu = atoi(e) > 0 ? atoi(e) : 4;
    return v;
  }();
  return t;
}

// Resolves the expression program and the aggregate arguments of a run-time configuration to LDS
// offsets (PlanInstr / PlanSum) for tiles of `tile_rows` rows whose column offsets (lds_off) are set.
// Temps get slots by liveness: a value's slot is free again after its last reader, and the reader
// itself may reuse it for its result (a thread reads its rows before it writes them).
static void plan_interpreter(DevConfig &dc, int tile_rows) {
  auto resolve = [&](const DevOperand &o, const int (&slot_of_temp)[QSX_MAX_TEMPS]) {
    PlanOperand p{};
    switch (o.kind) {
      case QSX_OPD_CONST:
        p.mode = kPlanImm;
        p.imm = dc.consts[o.index];
        break;
      case QSX_OPD_TEMP:
        p.mode = kPlanTempF64;
        p.off = slot_of_temp[o.index] * tile_rows * 8;
        break;
      default:
        p.off = dc.lds_off[o.index];
        switch (dc.column_type[o.index]) {
          case QSX_INT: p.mode = kPlanTileI32; break;
          case QSX_LONG: p.mode = kPlanTileI64; break;
          case QSX_FLOAT: p.mode = kPlanTileF32; break;
          default: p.mode = kPlanTileF64; break;
        }
        break;
    }
    return p;
  };
  const int n = dc.num_instrs;
  // last reader of the value defined by instruction k (n = an aggregate argument reads it; -1 = nobody)
  int last_use[QSX_MAX_INSTRS];
  for (int k = 0; k < n; ++k) {
    last_use[k] = -1;
    const int temp = dc.instrs[k].dst;
    for (int j = k + 1; j < n; ++j) {
      if ((dc.instrs[j].a.kind == QSX_OPD_TEMP && dc.instrs[j].a.index == temp) ||
          (dc.instrs[j].b.kind == QSX_OPD_TEMP && dc.instrs[j].b.index == temp)) {
        last_use[k] = j;
      }
      if (dc.instrs[j].dst == temp) break;  // redefined: later readers see the new value
    }
    bool redefined = false;
    for (int j = k + 1; j < n; ++j) redefined = redefined || dc.instrs[j].dst == temp;
    if (!redefined) {
      for (int j = 0; j < dc.num_sums; ++j) {
        if (dc.sums[j].arg.kind == QSX_OPD_TEMP && dc.sums[j].arg.index == temp) last_use[k] = n;
      }
    }
  }
  int slot_of_temp[QSX_MAX_TEMPS];   // slot currently holding temp t
  int slot_free_after[QSX_MAX_INSTRS];  // per slot: instruction index after which it is free (-1 = free)
  int num_slots = 0;
  for (int t = 0; t < QSX_MAX_TEMPS; ++t) slot_of_temp[t] = 0;
  for (int k = 0; k < n; ++k) {
    PlanInstr &pi = dc.plan_instrs[k];
    pi.op = dc.instrs[k].op;
    pi.a = resolve(dc.instrs[k].a, slot_of_temp);   // operands see the slots before this result is placed
    pi.b = resolve(dc.instrs[k].b, slot_of_temp);
    if (last_use[k] < 0) {
      pi.dst_off = -1;
      continue;
    }
    int slot = -1;
    for (int s2 = 0; s2 < num_slots && slot < 0; ++s2) {
      if (slot_free_after[s2] <= k) slot = s2;   // free, or its last reader is this very instruction
    }
    if (slot < 0) slot = num_slots++;
    slot_free_after[slot] = last_use[k];
    slot_of_temp[dc.instrs[k].dst] = slot;
    pi.dst_off = slot * tile_rows * 8;
  }
  dc.temps_bytes = num_slots * tile_rows * 8;
  for (int j = 0; j < dc.num_sums; ++j) {
    PlanSum &ps = dc.plan_sums[j];
    ps.is_int = dc.sums[j].is_int;
    if (dc.sums[j].arg.kind == kOpdKeyWord) {   // hidden accumulator of a wide key: the kernel packs the word itself
      ps.arg = PlanOperand{};
      ps.width = 8;
      continue;
    }
    ps.arg = resolve(dc.sums[j].arg, slot_of_temp);
    ps.width = dc.sums[j].arg.kind == QSX_OPD_COLUMN ? dc.column_width[dc.sums[j].arg.index] : 8;
  }
}

// Lays the referenced columns of one TR-row tile out in LDS and launches the
// update kernel with two tile buffers (DMA double buffering).
// Replication of the LDS accumulators vs workgroups per CU.  More replication = fewer same-address LDS atomics, but the
// planes compete with the tile for the 160 KiB of a CU, and what decides the kernel's speed is first of all how many
// workgroups are resident (their stage / compute phases overlap each other).  Measured on the Q1 shape (tools/agg_sweep*.sh,
// ms per 200 M rows): 64 KiB-ish allocations are granted in 1 KiB steps; rep 16 with 3 workgroups 1.28, with 2 workgroups
// 1.52; rep 8 with 3 workgroups 1.44, with 4 workgroups 1.44; rep 4: 1.92.  Start from the budgeted replication and give
// one step of it up when that admits one more workgroup below three per CU.
static int choose_replication(int NS, int S, size_t fixed_bytes, const AggTuning &tune, size_t *lds_out) {
  constexpr size_t kMaxLds = 160 * 1024;
  auto lds_of = [&](int rep) {
    // (+ kHashCtlWords behind the planes: pressure flags and flush statistics of the write-combining mode, agg_hash_update.hpp)
    return fixed_bytes + sizeof(unsigned long long) * (S + static_cast<size_t>(NS + 1) * ((static_cast<size_t>(S) << rep) + kWave) + kHashCtlWords);
  };
  auto per_cu_of = [&](size_t lds) {
    const size_t granted = (lds + 1023) / 1024 * 1024;
    return granted > kMaxLds ? 0 : static_cast<int>(kMaxLds / granted);
  };
  int rep_shift = 6;
  while (rep_shift > 0 && (static_cast<size_t>(NS + 1) * S * 8) << rep_shift > static_cast<size_t>(tune.acc_kib) * 1024) --rep_shift;
  if (rep_shift > 0 && per_cu_of(lds_of(rep_shift)) < 3 && per_cu_of(lds_of(rep_shift - 1)) > per_cu_of(lds_of(rep_shift))) --rep_shift;
  *lds_out = lds_of(rep_shift);
  return rep_shift;
}

template <int NS, int V>
static int launch_hash_v(DevConfig dc, unsigned used_columns, int64_t n, const uint64_t *filter,
                         const HashTableView &g, int S, int ranges, const long long *pieces, hipStream_t stream, bool dry_run,
                         bool runs = false) {
  constexpr int TR = kABlock * V;
  const size_t off = static_cast<size_t>(plan_tile(dc, used_columns, TR, filter != nullptr));
  plan_interpreter(dc, TR);
  // replicate every accumulator as far as the budget allows (64 = one bank column per lane)
  const AggTuning &tune = agg_tuning();
  const int nbuf = tune.buffers == 0 ? 1 : tune.buffers;   // (0 = chosen per run-time shape: jit_geometry_for)
  size_t lds = 0;
  const int rep_shift = choose_replication(NS, S, nbuf * off + dc.temps_bytes, tune, &lds);
  constexpr size_t kMaxLds = 160 * 1024;
  if (lds > kMaxLds) return QSX_ERR_CAPACITY;
  if (dry_run) return QSX_OK;
  static PerDeviceOnce attribute_set, runs_attribute_set;  // per instantiation
  {
    const int rc = runs ? once_per_device(runs_attribute_set, [] {
      return hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_hash_update_runs_kernel<NS, V>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
    }) : once_per_device(attribute_set, [] {
      return hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_hash_update_kernel<NS, V>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
    });
    if (rc != QSX_OK) return rc;
  }
  // grid = what is resident at once (LDS-limited workgroups per CU), tiles are strided over it
  int per_cu = static_cast<int>(kMaxLds / ((lds + 1023) / 1024 * 1024));   // LDS is granted in 1 KiB steps
  if (per_cu > tune.max_blocks_per_cu) per_cu = tune.max_blocks_per_cu;
  if (per_cu < 1) per_cu = 1;
  const int64_t num_tiles = (n + TR - 1) / TR;
  const int64_t max_grid = static_cast<int64_t>(kCUs) * per_cu;
  int grid = static_cast<int>(num_tiles * ranges < max_grid ? num_tiles * ranges : max_grid);
  grid = grid / ranges * ranges;
  if (grid < ranges) grid = ranges;
  if (runs) {   // (pieces = the block-run table)
    hipLaunchKernelGGL((agg_hash_update_runs_kernel<NS, V>), dim3(grid), dim3(kABlock), lds, stream, dc, n, g, S, rep_shift, nbuf, ranges, pieces);
    return QSX_OK;
  }
  hipLaunchKernelGGL((agg_hash_update_kernel<NS, V>), dim3(grid), dim3(kABlock), lds, stream, dc, n, filter, g, S,
                     rep_shift, nbuf, ranges, pieces);
  return QSX_OK;
}

static int agg_rows_per_thread() { return agg_tuning().rows_per_thread; }

// ---- AOT plan shapes (agg_shapes.hpp) -------------------------------------------------
typedef int (*ShapeLauncher)(const void *const *cols, int num_columns, int64_t n, const HashTableView &g, int S,
                             int ranges, const long long *pieces, hipStream_t stream, bool runs);
typedef int (*ShapeDirLauncher)(const void *const *cols, int num_columns, int64_t n, const HashTableView &g, const DirView &d, int gids,
                                int nbuf, hipStream_t stream, const long long *block_run);
struct ShapeEntry {
  const char *name;
  qsx_agg_config_t config;
  ShapeLauncher launch;
  ShapeDirLauncher launch_dir;
};

// Group-directory launch: one 1024-thread workgroup per CU, tiles strided over them.
static int dir_grid(int64_t n) {
  const int64_t num_tiles = (n + kDirBlock - 1) / kDirBlock;
  return static_cast<int>(num_tiles < kCUs ? num_tiles : kCUs);
}
// num_sums: the accumulators with an LDS plane — a wide key's hidden MIN / MAX accumulators have none (agg_hash_update.hpp)
static int dir_plane_sums(const DevConfig &dev) { return dev.num_sums - 2 * dev.wide_words; }
static size_t dir_lds_bytes(int tile_bytes, int temps_bytes, int num_sums, int gids, int nbuf) {
  return static_cast<size_t>(nbuf) * tile_bytes + temps_bytes + static_cast<size_t>(gids + kWave) * (8 * static_cast<size_t>(num_sums) + 4) + 16;
}

// Dense states of few entries live in LDS during an update (agg_hash_update.hpp, kDense && kDir): entries != 0 when the
// state qualifies, rep_shift = log2 of the copies per entry (few entries: same-address LDS atomics pile up otherwise).
// QSX_AGG_DENSE_LDS=0: the per-row global atomics for every dense state.
struct DenseLdsGeometry {
  int entries;     // per workgroup (0: the state does not qualify)
  int rep_shift;
  int ranges;      // families of workgroups, each reading every row and keeping `entries` consecutive entries
};
static DenseLdsGeometry dense_lds_geometry(long long num_entries, int num_sums, size_t tile_and_temps_bytes, bool allow_families) {
  const char *e = getenv("QSX_AGG_DENSE_LDS");   // read per call: tests and tools compare the paths
  // Up to 8 families: eight reads of the input (~0.45 ms per 100 M rows each) still beat 2 global atomics per row at 24 G/s
  // (8.9 ms); beyond that they do not.  What a workgroup holds is what one tile buffer leaves of the CU's LDS.
  constexpr long long kMaxRanges = 8;
  const size_t per_entry = 8 * static_cast<size_t>(num_sums) + 4;
  const size_t fixed = tile_and_temps_bytes + kWave * per_entry + 16;
  if ((e != nullptr && e[0] == '0') || num_entries <= 0 || fixed + 64 * per_entry > 160 * 1024) return DenseLdsGeometry{0, 0, 1};
  const long long capacity = static_cast<long long>((160 * 1024 - fixed) / per_entry / 64 * 64);
  if (num_entries > capacity * kMaxRanges) return DenseLdsGeometry{0, 0, 1};
  if (num_entries > capacity) {
    // (several reads of the input only pay when the rows do not come clustered on the key: decide_dense_families)
    if (!allow_families) return DenseLdsGeometry{0, 0, 1};
    const long long ranges = (num_entries + capacity - 1) / capacity;
    const long long per = ((num_entries + ranges - 1) / ranges + 63) / 64 * 64;   // whole existence words per family
    return DenseLdsGeometry{static_cast<int>(per), 0, static_cast<int>(ranges)};
  }
  int rep_shift = 0;
  while (rep_shift < 6 && (num_entries << (rep_shift + 1)) <= 2048) ++rep_shift;
  return DenseLdsGeometry{static_cast<int>(num_entries), rep_shift, 1};
}
// One 1024-thread workgroup per CU, a multiple of the families.
static int dense_lds_grid(int64_t n, int ranges) {
  int grid = dir_grid(n * ranges) / ranges * ranges;
  return grid < ranges ? ranges : grid;
}

template <typename Shape>
static int launch_shape_dir(const void *const *cols, int num_columns, int64_t n, const HashTableView &g, const DirView &d, int gids,
                            int nbuf, hipStream_t stream, const long long *block_run) {
  constexpr Translated T = Shape::translated(kDirBlock);
  static_assert(T.status == QSX_OK, "plan shape does not translate");
  constexpr size_t kMaxLds = 160 * 1024;
  const size_t lds = dir_lds_bytes(T.dev.tile_bytes, 0, T.num_sums - 2 * T.dev.wide_words, gids, nbuf);
  if (lds > kMaxLds) return QSX_ERR_CAPACITY;
  static PerDeviceOnce attribute_set;
  {
    const int rc = once_per_device(attribute_set, [] {
      return hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_dir_shape_kernel<Shape>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
    });
    if (rc != QSX_OK) return rc;
  }
  if (block_run != nullptr) {
    static PerDeviceOnce runs_attribute_set;
    const int rc = once_per_device(runs_attribute_set, [] {
      return hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_dir_shape_runs_kernel<Shape>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
    });
    if (rc != QSX_OK) return rc;
    hipLaunchKernelGGL((agg_dir_shape_runs_kernel<Shape>), dim3(dir_grid(n)), dim3(kDirBlock), lds, stream, n, g, d, gids, nbuf, block_run);
    return QSX_OK;
  }
  ColumnPointers cp;
  for (int i = 0; i < QSX_MAX_COLUMNS; ++i) cp.p[i] = i < num_columns ? cols[i] : nullptr;
  if (nbuf == 2 && dir_rows_per_thread() == 2) {
    // (two buffers fit: one 2048-row tile in their place, two rows per thread — see state_jit_kernel)
    constexpr Translated W = Shape::translated(2 * kDirBlock);
    const size_t wide_lds = dir_lds_bytes(W.dev.tile_bytes, 0, W.num_sums - 2 * W.dev.wide_words, gids, 1);
    if (wide_lds <= kMaxLds) {
      static PerDeviceOnce wide_attribute_set;
      const int rc = once_per_device(wide_attribute_set, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_dir_shape_wide_kernel<Shape>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
      });
      if (rc != QSX_OK) return rc;
      hipLaunchKernelGGL((agg_dir_shape_wide_kernel<Shape>), dim3(dir_grid(n)), dim3(kDirBlock), wide_lds, stream, cp, n, g, d, gids);
      return QSX_OK;
    }
  }
  hipLaunchKernelGGL((agg_dir_shape_kernel<Shape>), dim3(dir_grid(n)), dim3(kDirBlock), lds, stream, cp, n, g, d, gids, nbuf);
  return QSX_OK;
}

// Build pass of the group directory: stages the key and predicate columns only.
static int launch_dir_build(DevConfig dc, unsigned key_columns, int64_t n, const uint64_t *filter, DirView d, int gids,
                            unsigned call, hipStream_t stream, const long long *block_run) {
  // the sample: every stride-th tile, at least ~4 M rows of a large input (all of a small one), another phase every call
  const char *e = getenv("QSX_AGG_DIR_SAMPLE_ROWS");   // tests shrink the sample
  const int64_t sample_rows = e != nullptr && atoll(e) > 0 ? atoll(e) : (4 << 20);
  const int64_t stride = std::max<int64_t>(1, std::min<int64_t>(64, n / sample_rows));
  d.sample_stride = static_cast<int>(stride);
  d.sample_phase = static_cast<int>(call % stride);
  QSX_HIP_TRY(hipMemsetAsync(d.bounds, 0, sizeof(unsigned long long) * 2 * QSX_MAX_KEYS, stream));
  plan_tile(dc, key_columns, kDirBlock, filter != nullptr);
  dc.temps_bytes = 0;
  constexpr size_t kMaxLds = 160 * 1024;
  // LDS set of the workgroup's distinct codes: twice the gids, what the CU has room for at most
  // (a wide key: its words behind the codes)
  const size_t slot_bytes = dc.wide_words != 0 ? 8 * (1 + static_cast<size_t>(kMaxKeyWords)) : 8;
  int slots = static_cast<int>(next_pow2(static_cast<uint64_t>(gids) * 2));
  int nbuf = 2;
  while (slots > 1024 && static_cast<size_t>(nbuf) * dc.tile_bytes + slot_bytes * static_cast<size_t>(slots) + 16 > kMaxLds) slots >>= 1;
  const size_t lds = static_cast<size_t>(nbuf) * dc.tile_bytes + slot_bytes * static_cast<size_t>(slots) + 16;
  if (lds > kMaxLds) return QSX_ERR_CAPACITY;
  static PerDeviceOnce attribute_set;
  {
    const int rc = once_per_device(attribute_set, [] {
      hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_dir_build_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
      if (err == hipSuccess) {
        err = hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_dir_build_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(kMaxLds));
      }
      return err;
    });
    if (rc != QSX_OK) return rc;
  }
  const int64_t sampled_tiles = ((n + kDirBlock - 1) / kDirBlock - d.sample_phase + stride - 1) / stride;
  // (all CUs: the pass is bound by the LDS compare-and-swaps of the workgroups' code sets — 2.7 us per 1024-row tile —
  // not by the inserts at its end: 64 workgroups took 0.54 ms for the sample that 256 read in 0.32 ms)
  const int grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(sampled_tiles, kCUs)));
  for (int step = 0; step < 2; ++step) {   // bounds, then the directory itself (agg_common.hpp DirView::build_step)
    d.build_step = step;
    const size_t step_lds = step == 0 ? static_cast<size_t>(nbuf) * dc.tile_bytes + 16 : lds;
    if (block_run != nullptr) {
      hipLaunchKernelGGL(agg_dir_build_kernel<true>, dim3(grid), dim3(kDirBlock), step_lds, stream, dc, n, filter, d, slots, nbuf, block_run);
    } else {
      hipLaunchKernelGGL(agg_dir_build_kernel<false>, dim3(grid), dim3(kDirBlock), step_lds, stream, dc, n, filter, d, slots, nbuf, block_run);
    }
  }
  return QSX_OK;
}

template <int NS>
static int launch_dir(DevConfig dc, unsigned used_columns, int64_t n, const uint64_t *filter, const HashTableView &g, const DirView &d,
                      int gids, int nbuf, hipStream_t stream, const long long *block_run) {
  plan_tile(dc, used_columns, kDirBlock, filter != nullptr);
  plan_interpreter(dc, kDirBlock);
  constexpr size_t kMaxLds = 160 * 1024;
  const size_t lds = dir_lds_bytes(dc.tile_bytes, dc.temps_bytes, dir_plane_sums(dc), gids, nbuf);
  if (lds > kMaxLds) return QSX_ERR_CAPACITY;
  static PerDeviceOnce attribute_set;
  {
    const int rc = once_per_device(attribute_set, [] {
      hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_dir_update_kernel<NS, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
      if (err == hipSuccess) {
        err = hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_dir_update_kernel<NS, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
      }
      return err;
    });
    if (rc != QSX_OK) return rc;
  }
  if (block_run != nullptr) {
    hipLaunchKernelGGL((agg_dir_update_kernel<NS, true>), dim3(dir_grid(n)), dim3(kDirBlock), lds, stream, dc, n, filter, g, d, gids, nbuf, block_run);
  } else {
    hipLaunchKernelGGL((agg_dir_update_kernel<NS, false>), dim3(dir_grid(n)), dim3(kDirBlock), lds, stream, dc, n, filter, g, d, gids, nbuf, block_run);
  }
  return QSX_OK;
}

template <typename Shape, int V>
static int launch_shape_v(const void *const *cols, int num_columns, int64_t n, const HashTableView &g, int S,
                          int ranges, const long long *pieces, hipStream_t stream, bool runs) {
  constexpr int TR = kABlock * V;
  constexpr Translated T = Shape::translated(TR);
  static_assert(T.status == QSX_OK, "plan shape does not translate");
  constexpr int NS = T.num_sums;
  const AggTuning &tune = agg_tuning();
  const int nbuf = tune.buffers == 0 ? 1 : tune.buffers;   // (0 = chosen per run-time shape: jit_geometry_for)
  size_t lds = 0;
  const int rep_shift = choose_replication(NS, S, static_cast<size_t>(nbuf) * T.dev.tile_bytes, tune, &lds);
  constexpr size_t kMaxLds = 160 * 1024;
  if (lds > kMaxLds) return QSX_ERR_CAPACITY;
  static PerDeviceOnce attribute_set;
  {
    const int rc = once_per_device(attribute_set, [] {
      return hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_hash_shape_kernel<Shape, V>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
    });
    if (rc != QSX_OK) return rc;
  }
  int per_cu = static_cast<int>(kMaxLds / ((lds + 1023) / 1024 * 1024));   // LDS is granted in 1 KiB steps
  if (per_cu > tune.max_blocks_per_cu) per_cu = tune.max_blocks_per_cu;
  if (per_cu < 1) per_cu = 1;
  const int64_t num_tiles = (n + TR - 1) / TR;
  const int64_t max_grid = static_cast<int64_t>(kCUs) * per_cu;
  int grid = static_cast<int>(num_tiles * ranges < max_grid ? num_tiles * ranges : max_grid);
  grid = grid / ranges * ranges;
  if (grid < ranges) grid = ranges;
  ColumnPointers cp;
  for (int i = 0; i < QSX_MAX_COLUMNS; ++i) cp.p[i] = i < num_columns ? cols[i] : nullptr;   // (AOT shapes: plain columns only)
  if (getenv("QSX_DEBUG_LAUNCH") != nullptr) {
    std::fprintf(stderr, "[qsx] shape launch grid=%d lds=%zu S=%d rep_shift=%d nbuf=%d ranges=%d tile_bytes=%d n=%lld\n", grid, lds, S,
                 rep_shift, nbuf, ranges, T.dev.tile_bytes, static_cast<long long>(n));
  }
  // the two geometries the defaults produce — a handful of groups in one family (Q1), and the partitioned path's 32 pieces
  // with 1024-slot tables (10 k groups) — have kernels with those numbers as constants
#define QSX_LAUNCH_FIXED(S_, REP_, RANGES_, REG_)                                                                                  \
  do {                                                                                                                             \
    static PerDeviceOnce fixed_attribute_set;                                                                                      \
    const int attr_rc = once_per_device(fixed_attribute_set, [] {                                                                  \
      return hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_hash_shape_fixed_kernel<Shape, V, S_, REP_, RANGES_, REG_>),  \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));                          \
    });                                                                                                                            \
    if (attr_rc != QSX_OK) return attr_rc;                                                                                         \
    hipLaunchKernelGGL((agg_hash_shape_fixed_kernel<Shape, V, S_, REP_, RANGES_, REG_>), dim3(grid), dim3(kABlock), lds, stream,   \
                       cp, n, g, pieces);                                                                                          \
    return QSX_OK;                                                                                                                 \
  } while (0)
  if (runs && nbuf == 1 && S == 16 && rep_shift == 4 && ranges == 1) {   // the default small-group geometry as constants (Q1)
    static PerDeviceOnce fixed_runs_attribute_set;
    const int rc = once_per_device(fixed_runs_attribute_set, [] {
      hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_hash_shape_fixed_runs_kernel<Shape, V, 16, 4, 1, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
      if (err == hipSuccess) {
        err = hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_hash_shape_fixed_runs_kernel<Shape, V, 16, 4, 1, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
      }
      return err;
    });
    if (rc != QSX_OK) return rc;
    if (reg_groups_enabled()) {
      hipLaunchKernelGGL((agg_hash_shape_fixed_runs_kernel<Shape, V, 16, 4, 1, true>), dim3(grid), dim3(kABlock), lds, stream, n, g, pieces);
    } else {
      hipLaunchKernelGGL((agg_hash_shape_fixed_runs_kernel<Shape, V, 16, 4, 1, false>), dim3(grid), dim3(kABlock), lds, stream, n, g, pieces);
    }
    return QSX_OK;
  }
  if (runs) {   // a run of blocks (pieces = its table): stripes come from the table, not from the argument list
    static PerDeviceOnce runs_attribute_set;
    const int rc = once_per_device(runs_attribute_set, [] {
      return hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_hash_shape_runs_kernel<Shape, V>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
    });
    if (rc != QSX_OK) return rc;
    hipLaunchKernelGGL((agg_hash_shape_runs_kernel<Shape, V>), dim3(grid), dim3(kABlock), lds, stream, n, g, S, rep_shift, nbuf, ranges, pieces);
    return QSX_OK;
  }
  if (nbuf == 1 && S == 16 && rep_shift == 4 && ranges == 1 && pieces == nullptr) {
    if (reg_groups_enabled()) QSX_LAUNCH_FIXED(16, 4, 1, true);
    QSX_LAUNCH_FIXED(16, 4, 1, false);
  }
  if (nbuf == 1 && S == 1024 && rep_shift == 0 && ranges == 32 && pieces != nullptr) QSX_LAUNCH_FIXED(1024, 0, 32, false);
#undef QSX_LAUNCH_FIXED
  hipLaunchKernelGGL((agg_hash_shape_kernel<Shape, V>), dim3(grid), dim3(kABlock), lds, stream, cp, n, g, S, rep_shift, nbuf,
                     ranges, pieces);
  return QSX_OK;
}

template <typename Shape>
static int launch_shape(const void *const *cols, int num_columns, int64_t n, const HashTableView &g, int S,
                        int ranges, const long long *pieces, hipStream_t stream, bool runs) {
  if (agg_tuning().shape_rows_per_thread == 4) return launch_shape_v<Shape, 4>(cols, num_columns, n, g, S, ranges, pieces, stream, runs);
  return launch_shape_v<Shape, 2>(cols, num_columns, n, g, S, ranges, pieces, stream, runs);
}

// The numbers launch_shape_v derives, for the family's launchers (agg_family_part.hip).
int qsx::shape_launch_geometry(int NS, int S, int tile_bytes, ShapeGeometry *out) {
  const AggTuning &tune = agg_tuning();
  out->nbuf = tune.buffers == 0 ? 1 : tune.buffers;
  out->rep_shift = choose_replication(NS, S, static_cast<size_t>(out->nbuf) * tile_bytes, tune, &out->lds);
  constexpr size_t kMaxLds = 160 * 1024;
  if (out->lds > kMaxLds) return QSX_ERR_CAPACITY;
  int per_cu = static_cast<int>(kMaxLds / ((out->lds + 1023) / 1024 * 1024));   // LDS is granted in 1 KiB steps
  if (per_cu > tune.max_blocks_per_cu) per_cu = tune.max_blocks_per_cu;
  out->per_cu = per_cu < 1 ? 1 : per_cu;
  return QSX_OK;
}

// The family member that serves this state's plan (agg_family.hpp), with the state's column behind every canonical column; nullptr:
// the plan is not of the family.  Decided on the TRANSLATED plan: what the kernel body sees.
static const FamilyEntry *find_family(const qsx_agg_config_t &c, const DevConfig &d, int num_sums, int *num_columns, int *cols) {
  if (getenv("QSX_AGG_NO_SPECIALIZE") != nullptr && atoi(getenv("QSX_AGG_NO_SPECIALIZE")) != 0) return nullptr;
  if (getenv("QSX_AGG_FAMILY") != nullptr && atoi(getenv("QSX_AGG_FAMILY")) == 0) return nullptr;
  if (c.strategy != QSX_AGG_COMPACT_KEY && c.strategy != QSX_AGG_GENERIC) return nullptr;
  if (d.num_keys < 1 || d.num_keys > 2 || d.wide_words != 0 || d.num_instrs != 0 || d.num_null_cols != 0) return nullptr;
  for (int p = 0; p < d.num_pred; ++p) {   // the state's own predicate: a K1 pass per term in front of the update (family_filter)
    const int col = d.pred[p].column, type = d.column_type[col];
    if (d.code_width[col] != 0 || c.column_nullable[col] != 0) return nullptr;
    if (type != QSX_INT && type != QSX_LONG && type != QSX_FLOAT && type != QSX_DOUBLE && type != QSX_DATE) return nullptr;
  }
  if (num_sums < 1 || num_sums > kFamilyMaxSums) return nullptr;
  int kt[2] = {0, 0}, n = 0;
  for (int k = 0; k < d.num_keys; ++k) {
    const int col = d.key_column[k], type = d.column_type[col];
    if (d.code_width[col] != 0 || c.column_nullable[col] != 0) return nullptr;
    if (!((type == QSX_CHAR && d.column_width[col] == 1) || type == QSX_INT || type == QSX_LONG)) return nullptr;
    kt[k] = d.column_width[col];
    cols[n++] = col;
  }
  for (int j = 0; j < num_sums; ++j) {
    const DevSum &sum = d.sums[j];
    if (sum.kind != kAccSumF64 || sum.count_valid != 0 || sum.null_mask != 0 || sum.is_int != 0 || sum.arg.kind != QSX_OPD_COLUMN) return nullptr;
    const int col = sum.arg.index;
    if (d.column_type[col] != QSX_DOUBLE || d.code_width[col] != 0 || c.column_nullable[col] != 0) return nullptr;
    cols[n++] = col;
  }
  const FamilyEntry *e = find_family_entry(kt[0], kt[1], num_sums);
  if (e != nullptr) *num_columns = n;
  return e;
}

// A run of blocks of this state goes to the family's run kernels (agg_update_blocks then numbers the run table canonically):
// a hash state of the family without a predicate of its own, outside the group directory's mid-size group counts.
static bool family_serves_runs(const qsx_agg_state *st) { return st->family != nullptr && !st->dense && st->dev.num_pred == 0 && st->dir_gids == 0; }
static std::atomic<long long> g_family_launches{0};
// Test hook (not part of include/qsx.h): update calls this process has issued through a kernel of the AOT family.
extern "C" long long qsx_debug_agg_family_launches(void) { return g_family_launches.load(std::memory_order_relaxed); }

static const ShapeEntry *find_shape(const qsx_agg_config_t &c) {
  static const ShapeEntry table[] = {
      {"tpch_q1", ShapeTpchQ1::config(), &launch_shape<ShapeTpchQ1>, &launch_shape_dir<ShapeTpchQ1>},
      {"two_int_keys_sum_count_avg", ShapeTwoIntKeysSumCountAvg::config(), &launch_shape<ShapeTwoIntKeysSumCountAvg>,
       &launch_shape_dir<ShapeTwoIntKeysSumCountAvg>},
  };
  if (getenv("QSX_AGG_NO_SPECIALIZE") != nullptr && atoi(getenv("QSX_AGG_NO_SPECIALIZE")) != 0) return nullptr;
  for (const ShapeEntry &e : table) {
    if (same_plan(e.config, c)) return &e;
  }
  return nullptr;
}

// ---- run-time plan shapes (agg_jit.hpp) ---------------------------------------------------------
static long long jit_min_rows() {
  const char *e = getenv("QSX_AGG_JIT_MIN_ROWS");   // read per call: tests switch it at run time
  // The compile runs on a background thread and its result is kept per process by source text, so what the threshold
  // guards is a CPU core for 1-2 s per DISTINCT plan shape, not the caller's time: a shape is worth that once a state of
  // it has seen 256 Ki rows (6 us through the interpreter: what keeps the compiler away are states of a few blocks, tests) —
  // every later state of the shape, however small, then starts on the compiled kernel as soon as it has seen as many rows
  // itself.  (16 Mi until round 3, 2 Mi in round 3.)
  // Without a compiler driver the shape is built by hipRTC in the CALLING thread (agg_jit.hip: a background hipRTC compile
  // does not survive process exit): a Worker's update then stalls 1-2 s per distinct plan shape, which only a state of
  // round 3's 2 Mi rows is worth.
  if (e != nullptr) return atoll(e);
  return jit_compiles_out_of_process() ? 256ll * 1024 : 2048ll * 1024;
}

// The specialised kernel of this state for the filter variant, compiling it on first use once the state
// has aggregated jit_min_rows() rows; nullptr -> use the interpreter.
// Geometry of a shape launch for a state: the same numbers launch_hash_v / launch_shape_v derive, fixed per state and path.
static JitGeometry jit_geometry_for(const qsx_agg_state *st, int tile_bytes, int slots, int num_ranges, size_t *lds) {
  const int NS = st->num_sums;
  JitGeometry g{};
  if (st->dense) {
    g = JitGeometry{8, 0, 1, 1};
    *lds = static_cast<size_t>(tile_bytes) + sizeof(unsigned long long) * (8 + static_cast<size_t>(NS + 1) * (8 + kWave));
    return g;
  }
  const AggTuning &tune = agg_tuning();
  g.S = slots;
  g.ranges = num_ranges;
  g.nbuf = tune.buffers;
  // (room for the dictionaries the shape keeps in LDS: kDictLdsEntries 8-byte entries per compressed attribute it decodes
  // into registers, behind the control words — agg_hash_update.hpp)
  int decoded_columns = 0;
  for (int col = 0; col < st->dev.num_columns; ++col) {
    decoded_columns += st->dev.code_width[col] != 0 && ((st->used_columns >> col) & 1u) != 0 ? 1 : 0;
  }
  const size_t dict_bytes = static_cast<size_t>(decoded_columns) * kDictLdsEntries * 8;
  if (tune.buffers == 0) {
    // auto: a second tile buffer when three workgroups per CU still fit with it (small tiles: code stripes, narrow plans) —
    // the DMA of tile i + 1 then runs under the compute of tile i inside the workgroup as well
    size_t with_two = 0;
    choose_replication(NS, g.S, 2 * static_cast<size_t>(tile_bytes) + dict_bytes, tune, &with_two);
    g.nbuf = (with_two + 1023) / 1024 * 1024 * 3 <= 160 * 1024 ? 2 : 1;
  }
  g.rep_shift = choose_replication(NS, g.S, static_cast<size_t>(g.nbuf) * tile_bytes + dict_bytes, tune, lds);
  g.reg_groups = reg_groups_enabled() && g.ranges == 1 ? reg_groups_for(g.S, NS) : 0;
  // Small tiles (code stripes, narrow plans): LDS admits five or more workgroups per CU, the shape's ~100 registers four.
  // The kernel's phases (tile copy, argument reads, LDS atomics) overlap across workgroups only, so one more resident
  // workgroup is worth asking the register allocator for (Q1 over codes: 107 -> 96 registers without spilling).
  const int by_lds = static_cast<int>(160 * 1024 / ((*lds + 1023) / 1024 * 1024));
  g.waves_per_eu = by_lds >= 5 && g.reg_groups == 0 && tune.jit_waves_per_eu != 0 ? tune.jit_waves_per_eu : 0;
  return g;
}

// The specialised kernel of this state for the (filter, partitioned) variant, requested once the state has aggregated
// jit_min_rows() rows; nullptr -> use the interpreter (not requested yet, still compiling, or given up).
static const JitKernel *state_jit_kernel(qsx_agg_state *st, bool has_filter, bool partitioned, int slots, int num_ranges, int64_t n,
                                         int *variant, bool directory = false, bool runs = false) {
  const long long seen = st->rows_seen.fetch_add(n) + n;
  const int v = (has_filter ? 1 : 0) + (runs && directory ? 8 : (runs ? 6 : (directory ? 4 : (partitioned ? 2 : 0))));
  *variant = v;
  std::lock_guard<std::mutex> lock(st->jit_mutex);
  if (st->jit_tried[v]) return st->jit[v];          // settled: ready or given up
  if (st->jit_request[v] == nullptr) {
    const long long min_rows = jit_min_rows();
    if (seen < min_rows) return nullptr;
    if (getenv("QSX_AGG_NO_SPECIALIZE") != nullptr && atoi(getenv("QSX_AGG_NO_SPECIALIZE")) != 0) {
      st->jit_tried[v] = true;
      return nullptr;
    }
    DevConfig dev = st->dev;
    DenseLdsGeometry dense_lds{0, 0, 1};
    if (st->dense) {
      plan_tile(dev, st->used_columns, kDirBlock, has_filter);
      dense_lds = dense_lds_geometry(st->config.num_entries, st->num_sums, static_cast<size_t>(dev.tile_bytes), st->dense_families.load() == 1);
    }
    // (the shapes of the hash path and of the dense per-row path keep only the codes of compressed attributes in the tile:
    // their values live in registers; the directory and dense-in-LDS shapes decode into LDS slots)
    plan_tile(dev, st->used_columns, directory || dense_lds.entries != 0 ? kDirBlock : kABlock * jit_rows_per_thread(), has_filter,
              /*reg_decode=*/!directory && dense_lds.entries == 0);
    st->jit_tile_bytes[v] = dev.tile_bytes;
    if (dense_lds.entries != 0) {
      // a dense state in LDS (agg_hash_update.hpp, kDense && kDir): the directory kernels' geometry, S = entries
      const int copies = dense_lds.entries << dense_lds.rep_shift;
      int nbuf = 2;
      size_t lds = dir_lds_bytes(dev.tile_bytes, 0, st->num_sums, copies, nbuf);
      if (lds > 160 * 1024) lds = dir_lds_bytes(dev.tile_bytes, 0, st->num_sums, copies, nbuf = 1);
      st->jit_geometry[v] = JitGeometry{dense_lds.entries, dense_lds.rep_shift, nbuf, dense_lds.ranges, copies, runs ? 1 : 0};
      st->jit_lds[v] = lds;
      if (dir_rows_per_thread() == 2 && nbuf == 2 && !runs) {
        // one tile of 2048 rows instead of two buffers (see the directory below) — or of 4096 where few entries leave the room
        for (int rows_per_thread = 4; rows_per_thread >= 2; rows_per_thread >>= 1) {
          DevConfig wide = dev;
          plan_tile(wide, st->used_columns, rows_per_thread * kDirBlock, has_filter);
          const size_t wide_lds = dir_lds_bytes(wide.tile_bytes, 0, st->num_sums, copies, 1);
          if (wide_lds > 160 * 1024) continue;
          dev = wide;
          st->jit_tile_bytes[v] = dev.tile_bytes;
          st->jit_geometry[v].nbuf = 1;
          st->jit_geometry[v].dir_rows = rows_per_thread;
          st->jit_lds[v] = wide_lds;
          break;
        }
      }
    } else if (directory) {
      st->jit_geometry[v] = JitGeometry{st->dir_gids, 0, st->dir_nbuf, 1, st->dir_gids, runs ? 1 : 0};
      st->jit_lds[v] = dir_lds_bytes(dev.tile_bytes, 0, dir_plane_sums(dev), st->dir_gids, st->dir_nbuf);
      // Where the accumulators left room for two 1024-row buffers the shape takes ONE 2048-row tile, two rows per thread —
      // the same LDS.  The directory kernels run one workgroup per CU, and the copies of a tile do not run under that
      // workgroup's own LDS work: what a CU has in flight is one tile, so the tile is made as large as fits.
      if (dir_rows_per_thread() == 2 && st->dir_nbuf == 2 && !runs) {
        plan_tile(dev, st->used_columns, 2 * kDirBlock, has_filter);
        st->jit_tile_bytes[v] = dev.tile_bytes;
        st->jit_geometry[v].nbuf = 1;
        st->jit_geometry[v].dir_rows = 2;
        st->jit_lds[v] = dir_lds_bytes(dev.tile_bytes, 0, dir_plane_sums(dev), st->dir_gids, 1);
      }
    } else {
      st->jit_geometry[v] = jit_geometry_for(st, dev.tile_bytes, slots, num_ranges, &st->jit_lds[v]);
      st->jit_geometry[v].runs = runs ? 1 : 0;
    }
    if (st->jit_lds[v] > 160 * 1024) {               // the shape would not fit a CU: the interpreter's smaller tiles stay in use
      st->jit_tried[v] = true;
      return nullptr;
    }
    // hipRTC takes 1-2 s — three orders of magnitude more than interpreting the 16 Mi rows that trigger it — so the
    // compile runs in the background and the interpreter stays in use until the shape is ready.  QSX_AGG_JIT_MIN_ROWS=0
    // (compile at first use: tests, tools) or QSX_AGG_JIT_SYNC=1 wait for it instead.
    const char *sync_env = getenv("QSX_AGG_JIT_SYNC");
    const bool synchronous = min_rows == 0 || (sync_env != nullptr && atoi(sync_env) != 0);
    st->jit_request[v] = jit_agg_request(dev, st->num_sums, st->dense, st->jit_geometry[v], synchronous);
    if (st->jit_request[v] == nullptr) {             // run-time compilation is off
      st->jit_tried[v] = true;
      return nullptr;
    }
  }
  const JitKernel *k = nullptr;
  const int state = jit_request_state(st->jit_request[v], &k);
  if (state == 0) return nullptr;                    // still compiling
  st->jit[v] = k;
  st->jit_tried[v] = true;
  return k;
}

// The null bitmaps of a call behind a pointer: a device slot per (thread, stream), filled in stream order.
static int upload_null_table(const DevConfig &dc, hipStream_t stream, const unsigned long long *const **out) {
  *out = nullptr;
  if (dc.num_null_cols == 0) return QSX_OK;
  NullTable host_table{};
  for (int sl = 0; sl < dc.num_null_cols; ++sl) host_table.p[sl] = dc.nulls[sl];
  NullTable *slot = device_slot<NullTable>(stream);
  if (slot == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  hipLaunchKernelGGL(store_struct_kernel<NullTable>, dim3(1), dim3(64), 0, stream, host_table, slot);
  *out = slot->p;
  return QSX_OK;
}

static int launch_jit(qsx_agg_state *st, const JitKernel *k, int variant, const void *const *cols, const void *const *dicts, int64_t n,
                      const uint64_t *filter, int slots, int num_ranges, const long long *pieces, hipStream_t stream,
                      const DevConfig *call_config = nullptr) {
  const int TR = kABlock * jit_rows_per_thread();
  constexpr size_t kMaxLds = 160 * 1024;
  const AggTuning &tune = agg_tuning();
  // the geometry the shape was compiled for (constants inside it); a call with another one cannot use it
  const JitGeometry &geo = st->jit_geometry[variant];
  if (!st->dense && (geo.S != slots || geo.ranges != num_ranges)) return QSX_ERR_UNSUPPORTED;
  int S = geo.S, ranges = geo.ranges, rep_shift = geo.rep_shift, nbuf = geo.nbuf;
  const size_t lds = st->jit_lds[variant];
  const int tile_bytes = st->jit_tile_bytes[variant];
  if (lds > kMaxLds) return QSX_ERR_CAPACITY;
  int per_cu = static_cast<int>(kMaxLds / ((lds + 1023) / 1024 * 1024));   // LDS is granted in 1 KiB steps
  int most = geo.waves_per_eu > tune.max_blocks_per_cu ? geo.waves_per_eu : tune.max_blocks_per_cu;
  // (the dense per-row shapes have no accumulators in LDS, a 20 KiB tile and ~60 registers: as many workgroups as fit —
  // K7 over 200 M clustered rows 1.09 ms with four per CU, 0.95 with six, 0.93 with seven)
  if (st->dense && geo.dir_gids == 0 && getenv("QSX_AGG_BLOCKS_PER_CU") == nullptr) most = 8;
  if (per_cu > most) per_cu = most;
  {
    // ... and the registers may admit fewer: workgroups beyond what is resident would run as a second, thinner round
    const int resident = jit_resident_blocks(k, kABlock, lds);
    if (resident > 0 && per_cu > resident) per_cu = resident;
  }
  if (per_cu < 1) per_cu = 1;
  const int64_t num_tiles = (n + TR - 1) / TR;
  const int64_t max_grid = static_cast<int64_t>(kCUs) * per_cu;
  int grid = static_cast<int>(num_tiles * ranges < max_grid ? num_tiles * ranges : max_grid);
  grid = grid / ranges * ranges;
  if (grid < ranges) grid = ranges;
  const bool dense_lds = st->dense && geo.dir_gids != 0;   // one 1024-thread workgroup per CU (state_jit_kernel)
  if (dense_lds) grid = dense_lds_grid(n, ranges);
  if (getenv("QSX_DEBUG_LAUNCH") != nullptr) {
    std::fprintf(stderr, "[qsx] jit launch grid=%d lds=%zu S=%d rep_shift=%d nbuf=%d ranges=%d tile_bytes=%d n=%lld\n", grid, lds, S,
                 rep_shift, nbuf, ranges, tile_bytes, static_cast<long long>(n));
  }
  ColumnPointers cp;
  for (int i = 0; i < QSX_MAX_COLUMNS; ++i) cp.p[i] = i < st->config.num_columns ? cols[i] : nullptr;
  // the dictionaries of this call (per block) go behind a pointer: see make_source on the kernarg size
  const void **dict_table = nullptr;
  if (st->has_coded_columns) {
    DictTable host_table;
    for (int i = 0; i < QSX_MAX_COLUMNS; ++i) {
      host_table.p[i] = i < st->config.num_columns ? dicts[i] : nullptr;
      host_table.entries[i] = i < st->config.num_columns && tl_dictionary_entries != nullptr && dicts[i] != nullptr ? tl_dictionary_entries[i] : 0;
    }
    DictTable *slot = device_slot<DictTable>(stream);
    if (slot == nullptr) return QSX_ERR_OUT_OF_MEMORY;
    hipLaunchKernelGGL(store_struct_kernel<DictTable>, dim3(1), dim3(64), 0, stream, host_table, slot);
    dict_table = slot->p;
  }
  const unsigned long long *const *null_table = nullptr;
  if (call_config != nullptr) {
    const int rc_nulls = upload_null_table(*call_config, stream, &null_table);
    if (rc_nulls != QSX_OK) return rc_nulls;
  }
  const int rc = jit_agg_launch(k, grid, lds, stream, cp, dict_table, n, filter, st->dense ? HashTableView{} : st->hash_view(),
                                st->dense ? st->dense_view() : DenseView{}, st->dense, S, rep_shift, nbuf, ranges, pieces,
                                dense_lds ? kDirBlock : kABlock, null_table);
  return rc;
}

static int launch_jit_dir(qsx_agg_state *st, const JitKernel *k, int variant, const void *const *cols, const void *const *dicts, int64_t n,
                          const uint64_t *filter, const DirView &dir, hipStream_t stream, const long long *block_run,
                          const DevConfig *call_config = nullptr) {
  const size_t lds = st->jit_lds[variant];
  if (lds > 160 * 1024) return QSX_ERR_CAPACITY;
  ColumnPointers cp;
  for (int i = 0; i < QSX_MAX_COLUMNS; ++i) cp.p[i] = i < st->config.num_columns ? cols[i] : nullptr;
  const void **dict_table = nullptr;
  if (st->has_coded_columns) {
    DictTable host_table;
    for (int i = 0; i < QSX_MAX_COLUMNS; ++i) {
      host_table.p[i] = i < st->config.num_columns ? dicts[i] : nullptr;
      host_table.entries[i] = i < st->config.num_columns && tl_dictionary_entries != nullptr && dicts[i] != nullptr ? tl_dictionary_entries[i] : 0;
    }
    DictTable *slot = device_slot<DictTable>(stream);
    if (slot == nullptr) return QSX_ERR_OUT_OF_MEMORY;
    hipLaunchKernelGGL(store_struct_kernel<DictTable>, dim3(1), dim3(64), 0, stream, host_table, slot);
    dict_table = slot->p;
  }
  const unsigned long long *const *null_table = nullptr;
  if (call_config != nullptr) {
    const int rc_nulls = upload_null_table(*call_config, stream, &null_table);
    if (rc_nulls != QSX_OK) return rc_nulls;
  }
  const int rc = jit_agg_launch_dir(k, dir_grid(n), lds, stream, cp, dict_table, n, filter, st->hash_view(), dir, block_run, null_table);
  if (rc != QSX_OK || hipGetLastError() != hipSuccess) return QSX_ERR_HIP;
  return QSX_OK;
}

template <int NS>
static int launch_hash(const DevConfig &dc, unsigned used_columns, int64_t n, const uint64_t *filter,
                       const HashTableView &g, int S, int ranges, const long long *pieces, hipStream_t stream, bool runs = false) {
  // 1024-row tiles when two of them (plus the group tables) fit the CU's LDS twice over, else 512-row tiles
  if (agg_rows_per_thread() == 4 && launch_hash_v<NS, 4>(dc, used_columns, n, filter, g, S, ranges, pieces, stream, true, runs) == QSX_OK) {
    return launch_hash_v<NS, 4>(dc, used_columns, n, filter, g, S, ranges, pieces, stream, false, runs);
  }
  int rc = launch_hash_v<NS, 2>(dc, used_columns, n, filter, g, S, ranges, pieces, stream, false, runs);
  // (with pieces the family count is fixed by the partitioning: keep it, only shrink the table)
  if (rc == QSX_ERR_CAPACITY) {
    rc = launch_hash_v<NS, 2>(dc, used_columns, n, filter, g, 64, pieces != nullptr && !runs ? ranges : 1, pieces, stream, false, runs);
  }
  return rc;
}
template <int NS>
static int launch_dense(DevConfig dc, unsigned used_columns, int64_t n, const uint64_t *filter, const DenseView &d,
                        hipStream_t stream, const long long *block_run, bool allow_families) {   // block_run: the rows are a run of blocks
  {
    // a state of few entries: accumulated in LDS, one workgroup per CU (agg_hash_update.hpp, kDense && kDir)
    DevConfig lc = dc;
    plan_tile(lc, used_columns, kDirBlock, filter != nullptr);
    plan_interpreter(lc, kDirBlock);
    const DenseLdsGeometry geo = dense_lds_geometry(d.num_entries, NS, static_cast<size_t>(lc.tile_bytes) + lc.temps_bytes, allow_families);
    constexpr size_t kMaxLds = 160 * 1024;
    int nbuf = 2;
    size_t lds = dir_lds_bytes(lc.tile_bytes, lc.temps_bytes, NS, geo.entries << geo.rep_shift, nbuf);
    if (lds > kMaxLds) lds = dir_lds_bytes(lc.tile_bytes, lc.temps_bytes, NS, geo.entries << geo.rep_shift, nbuf = 1);
    if (geo.entries != 0 && lds <= kMaxLds) {
      static PerDeviceOnce lds_attribute_set;
      const int rc = once_per_device(lds_attribute_set, [] {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_dense_lds_kernel<NS, false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
        if (err == hipSuccess) {
          err = hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_dense_lds_kernel<NS, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
        }
        return err;
      });
      if (rc != QSX_OK) return rc;
      const int grid = dense_lds_grid(n, geo.ranges);
      if (block_run != nullptr) {
        hipLaunchKernelGGL((agg_dense_lds_kernel<NS, true>), dim3(grid), dim3(kDirBlock), lds, stream, lc, n, filter, d, geo.entries,
                           geo.rep_shift, nbuf, geo.ranges, block_run);
      } else {
        hipLaunchKernelGGL((agg_dense_lds_kernel<NS, false>), dim3(grid), dim3(kDirBlock), lds, stream, lc, n, filter, d, geo.entries,
                           geo.rep_shift, nbuf, geo.ranges, block_run);
      }
      return QSX_OK;
    }
  }
  constexpr int V = 4;
  constexpr int TR = kABlock * V;
  plan_tile(dc, used_columns, TR, filter != nullptr);
  plan_interpreter(dc, TR);
  const int nbuf = 1;
  const size_t lds = static_cast<size_t>(nbuf) * dc.tile_bytes + dc.temps_bytes +
                     sizeof(unsigned long long) * (8 + static_cast<size_t>(NS + 1) * (8 + kWave));
  constexpr size_t kMaxLds = 160 * 1024;
  if (lds > kMaxLds) return QSX_ERR_CAPACITY;
  static PerDeviceOnce attribute_set;
  {
    const int rc = once_per_device(attribute_set, [] {
      return hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_dense_update_kernel<NS, V>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
    });
    if (rc != QSX_OK) return rc;
  }
  int per_cu = static_cast<int>(kMaxLds / ((lds + 1023) / 1024 * 1024));   // LDS is granted in 1 KiB steps
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  const int64_t num_tiles = (n + TR - 1) / TR;
  const int64_t max_grid = static_cast<int64_t>(kCUs) * per_cu;
  const int grid = static_cast<int>(num_tiles < max_grid ? num_tiles : max_grid);
  if (block_run != nullptr) {
    static PerDeviceOnce runs_attribute_set;
    const int rc = once_per_device(runs_attribute_set, [] {
      return hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_dense_update_runs_kernel<NS, V>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
    });
    if (rc != QSX_OK) return rc;
    hipLaunchKernelGGL((agg_dense_update_runs_kernel<NS, V>), dim3(grid), dim3(kABlock), lds, stream, dc, n, d, nbuf, block_run);
    return QSX_OK;
  }
  hipLaunchKernelGGL((agg_dense_update_kernel<NS, V>), dim3(grid), dim3(kABlock), lds, stream, dc, n, filter, d, nbuf);
  return QSX_OK;
}

#define QSX_DISPATCH_NS(ns, FN, ...)      \
  switch (ns) {                           \
    case 0: FN<0>(__VA_ARGS__); break;    \
    case 1: FN<1>(__VA_ARGS__); break;    \
    case 2: FN<2>(__VA_ARGS__); break;    \
    case 3: FN<3>(__VA_ARGS__); break;    \
    case 4: FN<4>(__VA_ARGS__); break;    \
    case 5: FN<5>(__VA_ARGS__); break;    \
    case 6: FN<6>(__VA_ARGS__); break;    \
    case 7: FN<7>(__VA_ARGS__); break;    \
    default: FN<8>(__VA_ARGS__); break;   \
  }

static bool dir_enabled() {
  const char *e = getenv("QSX_AGG_DIRECTORY");   // read per call: tests and tools compare the paths
  return e == nullptr || atoi(e) != 0;
}

// LDS-table geometry of the update kernel for `est` groups (launch_hash_v / launch_shape_v / the run-time shapes).
static void derive_geometry(qsx_agg_state *st, int64_t est) {
  st->geometry_est = est;
  st->lds_ranges = 1;
  st->part_count = 1;
  st->part_slots = 64;
  // workgroup-private LDS table: up to 512 slots (<= 40 KiB at NS = 8)
  uint64_t s = next_pow2(static_cast<uint64_t>(est) * 2);
  if (s < 8) s = 8;
  if (s > 512) {
    // More groups than a replicated 512-slot table holds: take the biggest unreplicated LDS
    // table that fits 104 KiB and split the groups over up to 8 hash ranges (each range reads
    // the whole input); beyond that, one range and the overflow goes to the global table.
    uint64_t smax = 4096;
    while (smax > 512 && 8 * (smax + static_cast<uint64_t>(st->num_sums + 1) * (smax + 64)) > 104 * 1024) smax >>= 1;
    const uint64_t ranges = (static_cast<uint64_t>(est) * 10 + smax * 7 - 1) / (smax * 7);  // load <= 0.7
    if (smax > 512 && ranges <= 8) {
      s = smax;
      st->lds_ranges = static_cast<int>(ranges < 1 ? 1 : ranges);
    } else {
      s = 512;
    }
  }
  st->lds_slots = static_cast<int>(s);
  if (st->lds_ranges > 1 || (static_cast<uint64_t>(est) * 10 > s * 7 && est > 256)) {
    // ~350 groups per piece -> a 1024-slot table at load <= 0.35 with room for replication
    uint64_t pieces = next_pow2((static_cast<uint64_t>(est) + 349) / 350);
    if (pieces > 64) pieces = 64;
    if (pieces > 1) {
      st->part_count = static_cast<int>(pieces);
      uint64_t ps = next_pow2((static_cast<uint64_t>(est) / pieces + 1) * 3);
      if (ps < 64) ps = 64;
      if (ps > 4096) ps = 4096;
      st->part_slots = static_cast<int>(ps);
    }
  }
  // More groups than the replicated LDS tables hold, but few enough that one CU's LDS holds an accumulator per group
  // (4 + 8 NS bytes each) next to the tile buffers: the group directory replaces the partition pass / the hash ranges.
  st->dir_gids = 0;
  if ((st->lds_ranges > 1 || st->part_count > 1) && dir_enabled()) {
    DevConfig dev = st->dev;
    const size_t tile = static_cast<size_t>(plan_tile(dev, st->used_columns, kDirBlock, true));
    plan_interpreter(dev, kDirBlock);
    // (a little head-room: the estimate is an estimate; groups beyond the accumulators take the global path, and the
    // table's growth re-derives the geometry when the estimate was far off)
    const size_t want = (static_cast<size_t>(est) + static_cast<size_t>(est) / 16 + 63) / 64 * 64;
    const size_t per_gid = 8 * static_cast<size_t>(dir_plane_sums(st->dev)) + 4;
    for (int nbuf = 2; nbuf >= 1 && st->dir_gids == 0; --nbuf) {
      const size_t fixed = nbuf * tile + dev.temps_bytes + kWave * per_gid + 16;
      if (fixed + want * per_gid <= 160 * 1024) {
        // all the accumulators the CU has room for (capped: the flush walks them): a key box (DirView::bounds) may span
        // more cells than there are groups
        const size_t fit = (160 * 1024 - fixed) / per_gid / 64 * 64;
        st->dir_gids = static_cast<int>(std::min<size_t>(fit, std::max<size_t>(want, 16384)));
        st->dir_nbuf = nbuf;
      }
    }
  }
}

// (Re)creates the directory for st->dir_gids.  Only between launches: caller holds table_mutex exclusively (or is the
// constructor) and the device is idle.  gids start over — they only have to be stable within the directory's life.
static hipError_t ensure_directory(qsx_agg_state *st) {
  if (st->dir_gids == 0) return hipSuccess;
  const unsigned long long want_cap = next_pow2(static_cast<uint64_t>(st->dir_gids) * 4);
  hipError_t err = hipSuccess;
  if (st->dir_ngids == nullptr) err = device_malloc(reinterpret_cast<void **>(&st->dir_ngids), kDirControlBytes);
  const int entry_words = st->dev.wide_words != 0 ? 4 : 2;
  if (err == hipSuccess && (st->dir_cap != want_cap || st->dir_entry_words != entry_words)) {
    (void)device_free(st->dir_entries);
    st->dir_entries = nullptr;
    st->dir_cap = want_cap;
    st->dir_entry_words = entry_words;
    err = device_malloc(reinterpret_cast<void **>(&st->dir_entries), want_cap * 8 * entry_words);
  }
  if (err == hipSuccess && st->dir_codes_len < st->dir_gids) {
    (void)device_free(st->dir_codes);
    st->dir_codes = nullptr;
    st->dir_codes_len = st->dir_gids;
    err = device_malloc(reinterpret_cast<void **>(&st->dir_codes), sizeof(unsigned long long) * st->dir_gids);
    if (err == hipSuccess && st->dev.wide_words != 0) {
      (void)device_free(st->dir_words);
      st->dir_words = nullptr;
      err = device_malloc(reinterpret_cast<void **>(&st->dir_words), sizeof(unsigned long long) * kMaxKeyWords * st->dir_gids);
    }
  }
  if (err == hipSuccess) err = hipMemset(st->dir_entries, 0xFF, st->dir_cap * 8 * entry_words);
  if (err == hipSuccess) err = hipMemset(st->dir_ngids, 0, kDirControlBytes);
  return err;
}

static hipError_t init_hash_image(qsx_agg_state *st, unsigned long long *image, unsigned long long cap, hipStream_t s) {
  hipError_t err = hipMemsetAsync(image, 0xFF, sizeof(unsigned long long) * (cap + 1), s);
  if (err == hipSuccess) err = hipMemsetAsync(image + (cap + 1), 0, sizeof(unsigned long long) * (cap + 1) * st->num_cols, s);
  if (err == hipSuccess && st->has_min_max) {
    hipLaunchKernelGGL(fill_identity_kernel, dim3(grid_for(static_cast<long long>(cap + 1), kABlock * 4)), dim3(kABlock), 0, s,
                       image + (cap + 1), static_cast<long long>(cap + 1), static_cast<long long>(cap + 1), st->num_cols, st->col_kinds);
    err = hipGetLastError();
  }
  return err;
}

static int init_log(qsx_agg_state *st, unsigned long long *log, unsigned long long count, const unsigned int *count_dev, hipStream_t s) {
  // (count on the device: normally 0 — a small grid that strides over whatever it finds)
  const unsigned long long words = count_dev != nullptr ? 256ull * kABlock * 8 : count * (st->num_cols + 1);
  hipLaunchKernelGGL(init_log_kernel, dim3(grid_for(static_cast<int64_t>(words), kABlock * 8)), dim3(kABlock), 0, s, log, st->num_cols + 1,
                     count, count_dev, kLogRecords, st->col_kinds);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

// Grow the table until it holds its groups at load <= 1/4 and the spill log is empty.  Caller holds table_mutex
// exclusively.  Counterpart of PackedPayloadHashTable::resize (storage/PackedPayloadHashTable.cpp:232-288) and
// ThreadPrivateCompactKeyHashTable::resize (.cpp:159-201), which double in place under the table's lock.
static int grow_and_drain(qsx_agg_state *st) {
  QSX_HIP_TRY(hipDeviceSynchronize());   // drain in-flight updates on every stream
  for (int round = 0; round < 40; ++round) {
    unsigned long long control[4];
    QSX_HIP_TRY(hipMemcpy(control, st->control, sizeof(control), hipMemcpyDeviceToHost));
    const unsigned long long groups = control[0];
    const unsigned long long spilled = control[3] & 0xFFFFFFFFull;
    if (static_cast<int>(control[1] & 0xFFFFFFFFu) != 0) return QSX_ERR_TOO_MANY_GROUPS;   // the log itself overflowed: rows were lost
    if (spilled == 0 && groups * 4 <= st->cap) break;
    unsigned long long new_cap = st->cap * 4;
    while (new_cap < 16 * groups) new_cap <<= 1;
    const size_t new_bytes = sizeof(unsigned long long) * (new_cap + 1) * (st->num_cols + 1);
    unsigned long long *bigger = nullptr, *old_log = nullptr;
    QSX_HIP_TRY(device_malloc(reinterpret_cast<void **>(&bigger), new_bytes));
    hipError_t err = init_hash_image(st, bigger, new_cap, nullptr);
    if (err == hipSuccess && spilled != 0) {
      // the spilled records are re-inserted from a copy: what still finds no slot goes to the (reset) log again
      err = device_malloc(reinterpret_cast<void **>(&old_log), sizeof(unsigned long long) * spilled * (st->num_cols + 1));
      if (err == hipSuccess) err = hipMemcpy(old_log, st->log, sizeof(unsigned long long) * spilled * (st->num_cols + 1), hipMemcpyDeviceToDevice);
    }
    if (err != hipSuccess) {
      set_last_error("grow_and_drain", err);
      (void)device_free(bigger);
      (void)device_free(old_log);
      return err == hipErrorOutOfMemory ? QSX_ERR_OUT_OF_MEMORY : QSX_ERR_HIP;
    }
    unsigned long long *old_image = st->image;
    const unsigned long long old_cap = st->cap;
    if (spilled != 0) {
      int rc = init_log(st, st->log, spilled, nullptr, nullptr);
      if (rc != QSX_OK) return rc;
    }
    QSX_HIP_TRY(hipMemset(st->control, 0, 4 * sizeof(unsigned long long)));   // the merge below recounts the groups
    st->image = bigger;
    st->cap = new_cap;
    st->image_bytes = new_bytes;
    const HashTableView g = st->hash_view();
    hipLaunchKernelGGL(merge_hash_kernel, dim3(grid_for(old_cap + 1, kABlock)), dim3(kABlock), 0, nullptr, old_image, old_cap,
                       st->num_cols, st->col_kinds, g);
    QSX_CHECK_LAUNCH();
    if (spilled != 0) {
      hipLaunchKernelGGL(drain_log_kernel, dim3(grid_for(static_cast<int64_t>(spilled), kABlock)), dim3(kABlock), 0, nullptr, old_log,
                         static_cast<unsigned int>(spilled), st->num_cols + 1, st->num_cols, st->col_kinds, g);
      QSX_CHECK_LAUNCH();
    }
    QSX_HIP_TRY(hipDeviceSynchronize());
    (void)device_free(old_image);
    (void)device_free(old_log);
  }
  unsigned long long groups = 0;
  QSX_HIP_TRY(hipMemcpy(&groups, st->control, sizeof(groups), hipMemcpyDeviceToHost));
  st->published[0] = groups;
  st->published[1] = 0;
  // The estimate was off: give the update kernel the LDS geometry of the group count actually seen (a table sized for 6
  // groups sends nearly every row of a 10 k-group input down the per-row global path).
  if (static_cast<int64_t>(groups) > 2 * st->geometry_est) {
    derive_geometry(st, static_cast<int64_t>(groups) * 2);
    QSX_HIP_TRY(ensure_directory(st));
    std::lock_guard<std::mutex> lock(st->jit_mutex);   // run-time shapes carry the geometry as constants: ask again
    for (int v = 0; v < kJitVariants; ++v) {
      st->jit_request[v] = nullptr;
      st->jit[v] = nullptr;
      st->jit_tried[v] = false;
    }
  }
  return QSX_OK;
}

// Before an update launch: grow when the control words published behind the previous launches say the table is past
// load 1/4 or rows went to the spill log.  Never waits for the device unless it grows.
static int maybe_grow(qsx_agg_state *st) {
  if (!st->growable) return QSX_OK;
  const unsigned long long groups = __atomic_load_n(&st->published[0], __ATOMIC_ACQUIRE);
  const unsigned long long spilled = __atomic_load_n(&st->published[1], __ATOMIC_ACQUIRE);
  {
    std::shared_lock<std::shared_mutex> lock(st->table_mutex);
    if (spilled == 0 && groups * 4 <= st->cap) return QSX_OK;
  }
  std::unique_lock<std::shared_mutex> lock(st->table_mutex);
  return grow_and_drain(st);
}

// Behind an update launch (same stream): refresh the published control words.
static int publish_control(qsx_agg_state *st, hipStream_t s) {
  if (!st->growable) return QSX_OK;
  hipLaunchKernelGGL(publish_control_kernel, dim3(1), dim3(64), 0, s, st->control, st->published_dev, st->publish_seq.fetch_add(1) + 1);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

// In front of everything that reads the state as a whole (group count, finalize, export, merge source): wait for the
// stream, fold spilled rows back in, report lost rows.
static int settle(qsx_agg_state *st, hipStream_t stream) {
  unsigned long long control[4];
  QSX_HIP_TRY(hipMemcpyAsync(control, st->control, sizeof(control), hipMemcpyDeviceToHost, stream));
  QSX_HIP_TRY(hipStreamSynchronize(stream));
  if (static_cast<int>(control[1] & 0xFFFFFFFFu) != 0) {
    return st->dense ? QSX_ERR_INVALID_ARGUMENT : QSX_ERR_TOO_MANY_GROUPS;
  }
  if (st->growable && ((control[3] & 0xFFFFFFFFull) != 0 || control[0] * 4 > st->cap)) {
    std::unique_lock<std::shared_mutex> lock(st->table_mutex);
    return grow_and_drain(st);
  }
  return QSX_OK;
}

// (update_run_end_to_end, below)
constexpr int kConcatChunk = 16 * 1024;
template <typename V>
__global__ __launch_bounds__(256) void concat_segments_kernel(const long long *__restrict__ segments, int first_segment) {
  const long long *seg = segments + 3 * static_cast<size_t>(first_segment + blockIdx.y);
  const V *src = as_global(reinterpret_cast<const V *>(seg[0]));
  V *dst = as_global(reinterpret_cast<V *>(seg[1]));
  const long long n = seg[2];
  for (long long chunk = blockIdx.x; chunk * kConcatChunk < n; chunk += gridDim.x) {
    const long long base = chunk * kConcatChunk;
#pragma unroll 4
    for (int i = threadIdx.x; i < kConcatChunk; i += 256) {
      if (base + i < n) store_global_nt(load_global_nt(&src[base + i]), &dst[base + i]);
    }
  }
}

extern "C" {

int qsx_agg_state_create(const qsx_agg_config_t *config, qsx_agg_state_t **out) {
  QSX_REQUIRE_DEVICE();
  if (config == nullptr || out == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  qsx_agg_state *st = new qsx_agg_state();
  st->config = *config;
  int rc = translate_config(*config, st);
  if (rc != QSX_OK) { delete st; return rc; }
  if (!st->dense) st->shape = find_shape(*config);
  if (!st->dense && st->shape == nullptr) st->family = find_family(*config, st->dev, st->num_sums, &st->family_num_columns, st->family_cols);
  if (st->dense) {
    st->exist_words = (config->num_entries + 63) / 64;
    st->image_bytes = sizeof(unsigned long long) * (st->exist_words + static_cast<size_t>(st->num_cols) * config->num_entries);
    st->max_tiles = (st->exist_words + kDenseTileWords - 1) / kDenseTileWords + 1;
  } else {
    const int64_t est = config->strategy == QSX_AGG_SINGLE_STATE ? 1 : (config->est_groups < 1 ? 1 : config->est_groups);
    // generous head-room: the estimate comes from the optimizer and a table cannot be grown in the middle of a
    // kernel — groups beyond it land in the spill log and the table grows before the next launch (maybe_grow / settle)
    st->cap = next_pow2(static_cast<uint64_t>(est) * 8 + 1024);
    st->image_bytes = sizeof(unsigned long long) * (st->cap + 1) * (st->num_cols + 1);
    derive_geometry(st, est);
    st->growable = config->strategy != QSX_AGG_SINGLE_STATE;
  }
  hipError_t err = device_malloc(reinterpret_cast<void **>(&st->image), st->image_bytes);
  if (err == hipSuccess) err = device_malloc(reinterpret_cast<void **>(&st->control), 8 * sizeof(unsigned long long));   // [4]: wide-key collision flag
  if (err == hipSuccess && !st->dense) err = ensure_directory(st);
  if (err == hipSuccess && st->growable) {
    err = device_malloc(reinterpret_cast<void **>(&st->log), sizeof(unsigned long long) * kLogRecords * (st->num_cols + 1));
    if (err == hipSuccess && (st->published = published_slots().take()) == nullptr) err = hipErrorOutOfMemory;
    if (err == hipSuccess) {
      std::memset(st->published, 0, 4 * sizeof(unsigned long long));
      err = hipHostGetDevicePointer(reinterpret_cast<void **>(&st->published_dev), st->published, 0);
    }
    if (err == hipSuccess && init_log(st, st->log, kLogRecords, nullptr, nullptr) != QSX_OK) err = hipErrorUnknown;
  }
  if (err == hipSuccess) err = hipMemset(st->control, 0, 8 * sizeof(unsigned long long));
  if (err == hipSuccess) {
    // InitializeAggregation: zero the state (CollisionFreeVectorTable.hpp:136-143)
    if (st->dense) {
      err = hipMemset(st->image, 0, st->image_bytes);
    } else {
      err = hipMemset(st->image, 0xFF, sizeof(unsigned long long) * (st->cap + 1));
      if (err == hipSuccess) {
        err = hipMemset(st->image + (st->cap + 1), 0, sizeof(unsigned long long) * (st->cap + 1) * st->num_cols);
      }
    }
  }
  if (err == hipSuccess && fill_identities(st, nullptr) != QSX_OK) err = hipErrorUnknown;
  if (err == hipSuccess) err = hipDeviceSynchronize();
  if (err != hipSuccess) {
    set_last_error("qsx_agg_state_create", err);
    (void)device_free(st->image); (void)device_free(st->control); (void)device_free(st->tile_counts); (void)device_free(st->tile_offsets);
    (void)device_free(st->log);
    (void)device_free(st->dir_entries); (void)device_free(st->dir_codes); (void)device_free(st->dir_words); (void)device_free(st->dir_ngids);
    published_slots().give(st->published);
    delete st;
    return err == hipErrorOutOfMemory ? QSX_ERR_OUT_OF_MEMORY : QSX_ERR_HIP;
  }
  *out = st;
  return QSX_OK;
}

int qsx_agg_state_destroy(qsx_agg_state_t *st) {
  if (st == nullptr) return QSX_OK;
  (void)synchronize_owner_device(st->image);   // the device the state lives on, whichever one the caller is on
  (void)device_free_idle(st->image);
  (void)device_free_idle(st->control);
  (void)device_free_idle(st->tile_counts);
  (void)device_free_idle(st->tile_offsets);
  (void)device_free_idle(st->log);
  (void)device_free_idle(st->dir_entries);
  (void)device_free_idle(st->dir_codes);
  (void)device_free_idle(st->dir_words);
  (void)device_free_idle(st->dir_ngids);
  published_slots().give(st->published);
  delete st;
  return QSX_OK;
}

int qsx_agg_state_clear(qsx_agg_state_t *st, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (st == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  hipStream_t s = as_stream(stream);
  std::unique_lock<std::shared_mutex> lock(st->table_mutex);
  if (st->growable) {
    // records a previous run left in the spill log go back to their identities (before the count is zeroed below)
    int rc = init_log(st, st->log, 0, reinterpret_cast<const unsigned int *>(st->control + 3), s);
    if (rc != QSX_OK) return rc;
    __atomic_store_n(&st->published[0], 0ull, __ATOMIC_RELEASE);
    __atomic_store_n(&st->published[1], 0ull, __ATOMIC_RELEASE);
  }
  if (st->dir_gids != 0) {
    QSX_HIP_TRY(hipMemsetAsync(st->dir_entries, 0xFF, st->dir_cap * 8 * st->dir_entry_words, s));
    QSX_HIP_TRY(hipMemsetAsync(st->dir_ngids, 0, 16, s));
  }
  if (st->dense) {
    QSX_HIP_TRY(hipMemsetAsync(st->image, 0, st->image_bytes, s));
  } else {
    QSX_HIP_TRY(hipMemsetAsync(st->image, 0xFF, sizeof(unsigned long long) * (st->cap + 1), s));
    QSX_HIP_TRY(hipMemsetAsync(st->image + (st->cap + 1), 0,
                               sizeof(unsigned long long) * (st->cap + 1) * st->num_cols, s));
  }
  QSX_HIP_TRY(hipMemsetAsync(st->control, 0, 8 * sizeof(unsigned long long), s));
  return fill_identities(st, s);
}

// Dense states between one LDS and eight: key-range families read the input `ranges` times (~0.45 ms per 100 M rows and
// read) and never touch a global atomic per row; the per-row path costs 2 atomics per RUN of equal adjacent keys at 24 G/s
// (8.4 ms per 100 M runs) but reads once.  Random keys: families win by 2.6-10x; lineitem clustered on l_orderkey (runs of
// ~4): the per-row path wins.  One sample of the first rows, once per state (one small synchronous read-back).
static void decide_dense_families(qsx_agg_state *st, const void *key_col, int64_t rows, hipStream_t s) {
  if (st->dense_families.load() != 0) return;
  DevConfig dev = st->dev;
  plan_tile(dev, st->used_columns, kDirBlock, false);
  const DenseLdsGeometry geo = dense_lds_geometry(st->config.num_entries, st->num_sums, static_cast<size_t>(dev.tile_bytes), true);
  if (geo.entries == 0 || geo.ranges == 1) {
    st->dense_families.store(1);   // (nothing to decide: one LDS holds it, or not even eight do)
    return;
  }
  const int key_col_index = st->dev.key_column[0];
  const int sample = static_cast<int>(std::min<int64_t>(rows, 256 * 1024));
  if (key_col == nullptr || st->dev.code_width[key_col_index] != 0 || sample < 4096) {
    if (sample >= 4096 || key_col == nullptr) st->dense_families.store(geo.ranges <= 4 ? 1 : 2);   // (codes: no sample — a middle course)
    return;                                                                                        // (a small first input: ask again)
  }
  CallScratch scratch(s);
  if (scratch.reserve(CallScratch::padded(sizeof(unsigned int))) != QSX_OK) return;
  unsigned int *counter = static_cast<unsigned int *>(scratch.take(sizeof(unsigned int)));
  unsigned int equal = 0;
  if (hipMemsetAsync(counter, 0, sizeof(unsigned int), s) != hipSuccess) return;
  if (st->dev.column_type[key_col_index] == QSX_LONG) {
    hipLaunchKernelGGL(adjacent_equal_kernel<long long>, dim3(1), dim3(1024), 0, s, static_cast<const long long *>(key_col), sample, counter);
  } else {
    hipLaunchKernelGGL(adjacent_equal_kernel<int32_t>, dim3(1), dim3(1024), 0, s, static_cast<const int32_t *>(key_col), sample, counter);
  }
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&equal, counter, sizeof(equal), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  const double runs_per_row = 1.0 - static_cast<double>(equal) / (sample - 1);        // 1 / average run length
  const double per_row_ms = std::max(1.1, 8.4 * runs_per_row), families_ms = 0.45 * geo.ranges;   // per 100 M rows
  st->dense_families.store(families_ms < per_row_ms ? 1 : 2);
}

// One launch of the update kernel over n rows with the given LDS table geometry: AOT plan shape, then the
// run-time one, then the interpreter (CAPACITY — the tile does not fit LDS next to the group tables — and
// compile failures fall through).
// block_run != nullptr: the rows are a run of blocks described by that device table (agg_common.hpp BlockRunView; cols /
// filter_dev then only say which columns and whether a filter exist); it travels in the kernels' `pieces` argument.
static int update_slice(qsx_agg_state *st, const void *const *cols, const void *const *dicts, int64_t n,
                        const uint64_t *filter_dev, int slots, int ranges, const long long *pieces, hipStream_t s,
                        const uint64_t *const *nulls = nullptr, const long long *block_run = nullptr, int64_t first_block_rows = 0) {
  const bool partitioned = pieces != nullptr;
  const bool runs = block_run != nullptr;
  if (runs) pieces = block_run;
  if (st->dense && st->dense_families.load() == 0) {
    // (a run of blocks: cols are the first block's stripes)
    decide_dense_families(st, st->dev.num_keys > 0 ? cols[st->dev.key_column[0]] : nullptr, runs ? first_block_rows : n, s);
  }
  DevConfig dc = st->dev;
  for (int i = 0; i < st->config.num_columns; ++i) {
    dc.cols[i] = cols[i];
    dc.dicts[i] = (dicts != nullptr && dc.code_width[i] != 0) ? dicts[i] : nullptr;
  }
  for (int sl = 0; sl < dc.num_null_cols; ++sl) {
    dc.nulls[sl] = nulls != nullptr ? reinterpret_cast<const unsigned long long *>(nulls[dc.null_column[sl]]) : nullptr;
  }
  const bool aot = !st->dense && st->shape != nullptr && filter_dev == nullptr && dc.num_null_cols == 0;
  // (a run of blocks: agg_update_blocks numbered the table canonically for exactly these states — family_serves_runs; a state with
  // a predicate of its own gets it as a filter from a K1 pass here, over one stripe per column and outside the partitioned path)
  const bool family = !aot && !st->dense && st->family != nullptr && dc.num_null_cols == 0 &&
                      (runs ? family_serves_runs(st) : (dc.num_pred == 0 || pieces == nullptr));
  int variant = 0;
  // (states over nullable columns: the run-time shapes take the null bitmaps of a call behind a pointer — a run of blocks
  // has one set per block, which the run table does not carry: those stay with the interpreter, block by block)
  const JitKernel *jk = (aot || family || (dc.num_null_cols != 0 && runs))
                            ? nullptr
                            : state_jit_kernel(st, filter_dev != nullptr, partitioned, slots, ranges, n, &variant, false, runs);
  if (jk != nullptr) {
    int rc = launch_jit(st, jk, variant, cols, dc.dicts, n, filter_dev, slots, ranges, pieces, s, &dc);
    if (rc == QSX_OK && hipGetLastError() == hipSuccess) return QSX_OK;
    // the specialised kernel could not be launched: keep going with the interpreter from now on
    std::lock_guard<std::mutex> lock(st->jit_mutex);
    st->jit[variant] = nullptr;
  }
  if (st->dense) {
    const DenseView d = st->dense_view();
    int rc = QSX_OK;
    QSX_DISPATCH_NS(st->num_sums, rc = launch_dense, dc, st->used_columns, n, filter_dev, d, s, block_run, st->dense_families.load() == 1);
    if (rc != QSX_OK) return rc;
  } else {
    const HashTableView g = st->hash_view();
    int rc = QSX_OK;
    if (aot) {
      rc = st->shape->launch(cols, st->config.num_columns, n, g, slots, ranges, pieces, s, runs);
    } else if (family) {
      const void *canonical[QSX_MAX_COLUMNS] = {};
      for (int i = 0; i < st->family_num_columns; ++i) canonical[i] = cols[st->family_cols[i]];
      const uint64_t *family_filter = filter_dev;
      CallScratch pred_scratch(s);
      if (!runs && dc.num_pred != 0) {   // the state's predicate -> this call's filter: the tuned K1 kernels, chained through the bitmap
        const size_t words = static_cast<size_t>((n + 63) / 64) + 1;
        rc = pred_scratch.reserve(CallScratch::padded(words * 8));
        if (rc != QSX_OK) return rc;
        uint64_t *bitmap = static_cast<uint64_t *>(pred_scratch.take(words * 8));
        for (int p = 0; p < dc.num_pred; ++p) {
          const unsigned long long literal = dc.pred[p].literal;   // (raw bits typed like the column: what qsx_select_cmp reads behind the pointer)
          rc = qsx_select_cmp(dc.column_type[dc.pred[p].column], cols[dc.pred[p].column], n, dc.pred[p].op, &literal, p == 0 ? filter_dev : bitmap, bitmap,
                              nullptr, reinterpret_cast<qsx_stream_t>(s));
          if (rc != QSX_OK) return rc;
        }
        family_filter = bitmap;
      }
      rc = st->family->launch(canonical, st->family_num_columns, n, family_filter, g, slots, ranges, pieces, s, runs);
      if (rc == QSX_OK) g_family_launches.fetch_add(1, std::memory_order_relaxed);
    } else {
      QSX_DISPATCH_NS(st->num_sums, rc = launch_hash, dc, st->used_columns, n, filter_dev, g, slots, ranges, pieces, s, runs);
    }
    if (rc != QSX_OK) return rc;
  }
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

// ---- aggregates factored through the dictionary codes (agg_factored.hpp) ------------------------------------------------
static std::atomic<long long> g_factored_launches{0};
// Test hook (not part of include/qsx.h): update calls this process has issued through the factored kernels.
extern "C" long long qsx_debug_agg_factored_launches(void) { return g_factored_launches.load(std::memory_order_relaxed); }
// Test hook (not part of include/qsx.h; needs no GPU): how factored_analyse() sees a configuration — out[0] = factors (0 / 1),
// out[1..3] = cell columns, histogram columns, carriers; out[4 + j] = histogram column of sum j or -1.
extern "C" int qsx_debug_agg_factored_plan(const qsx_agg_config_t *config, int32_t *out, int out_len) {
  if (config == nullptr || out == nullptr || out_len < 4 + kMaxSums) return QSX_ERR_INVALID_ARGUMENT;
  const Translated t = translate(*config);
  if (t.status != QSX_OK) return t.status;
  const FactoredStatic f = factored_analyse(t.dev, t.dense);
  out[0] = f.ok ? 1 : 0;
  out[1] = f.ncell;
  out[2] = f.nhist;
  out[3] = f.ncar;
  for (int j = 0; j < kMaxSums; ++j) out[4 + j] = j < t.num_sums ? f.sum_hist[j] : -1;
  return QSX_OK;
}
static bool factored_enabled() {
  const char *e = getenv("QSX_AGG_FACTORED");
  return e == nullptr || e[0] != '0';
}
static long long factored_min_rows() {
  const char *e = getenv("QSX_AGG_FACTORED_MIN_ROWS");
  return e != nullptr ? atoll(e) : 256ll * 1024;
}
// The plan of a call through the factored kernels: everything but the coefficient tables' addresses.
struct FactoredPlan {
  FactoredArgs a{};
  FactoredCoefArgs ca{};
  FactoredDirectArgs da{};
  size_t tile_bytes = 0, table_bytes = 0;
  bool direct = false;   // one of the direct-load kernel's signatures (stripe alignment is the caller's to check)
  int keyw = 0;
};
// cols / dicts: the stripes of the call (a run of blocks: of its first block); entries[column]: the radix of every dictionary
// column (a run of blocks: the largest dictionary of the run).  QSX_ERR_UNSUPPORTED: not a call for these kernels.
static int factored_plan(const qsx_agg_state *st, const void *const *cols, const void *const *dicts, const int32_t *entries, bool has_filter,
                         FactoredPlan *plan) {
  const FactoredStatic &f = st->factored;
  const DevConfig &d = st->dev;
  const int S = st->lds_slots;
  if (S < 1 || S > 64 || (S & (S - 1)) != 0 || st->lds_ranges != 1) return QSX_ERR_UNSUPPORTED;
  FactoredArgs &a = plan->a;
  FactoredCoefArgs &ca = plan->ca;
  size_t tile = 0;
  auto stage = [&](int col) {
    const int q = a.nstaged++;
    a.col[q] = cols[col];
    a.width[q] = d.code_width[col] != 0 ? d.code_width[col] : d.column_width[col];
    a.off[q] = static_cast<int>(tile);
    tile += align16(static_cast<size_t>(kFacTileRows) * a.width[q]);
    return q;
  };
  for (int col = 0; col < d.num_columns; ++col) {
    if (((st->used_columns >> col) & 1u) != 0 && cols[col] == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  }
  a.nkeys = d.num_keys;
  for (int k = 0; k < d.num_keys; ++k) {
    a.key_slot[k] = stage(d.key_column[k]);
    a.key_shift[k] = d.key_shift[k];
  }
  long long cells = 1;
  a.ncell = ca.ncell = f.ncell;
  for (int q = 0; q < f.ncell; ++q) {
    const int col = f.cell_col[q], radix = entries[col];
    if (dicts[col] == nullptr || radix < 1 || radix > kFacMaxDict) return QSX_ERR_UNSUPPORTED;
    a.cell_slot[q] = stage(col);
    a.cell_stride[q] = ca.cell_stride[q] = static_cast<int>(cells);
    a.cell_radix[q] = ca.cell_radix[q] = radix;
    ca.cell_col[q] = col;
    cells *= radix;
    if (cells > kFacMaxCells) return QSX_ERR_UNSUPPORTED;
  }
  a.cells = ca.cells = static_cast<int>(cells);
  a.nhist = ca.nhist = f.nhist;
  for (int h = 0; h < f.nhist; ++h) {
    const int col = f.hist_col[h], size = entries[col];
    if (dicts[col] == nullptr || size < 1 || size > kFacMaxDict) return QSX_ERR_UNSUPPORTED;
    a.hist_slot[h] = stage(col);
    a.hist_off[h] = a.hist_words;
    a.hist_size[h] = ca.hist_size[h] = size;
    ca.hist_col[h] = col;
    a.hist_words += size;
  }
  a.ncar = ca.ncar = f.ncar;
  for (int k = 0; k < f.ncar; ++k) {
    a.car_slot[k] = stage(f.car_col[k]);
    a.car_type[k] = d.column_type[f.car_col[k]];
    a.car_int[k] = f.car_int[k];
    ca.car_col[k] = f.car_col[k];
  }
  a.filter_off = static_cast<int>(tile);
  if (has_filter) tile += align16(kFacTileRows / 64 * 8);
  a.tile_bytes = static_cast<int>(tile);
  a.S = S;
  a.nsums = ca.nsums = d.num_sums;
  for (int j = 0; j < d.num_sums; ++j) {
    a.sum_kind[j] = d.sums[j].kind;
    a.sum_hist[j] = ca.sum_hist[j] = f.sum_hist[j];
    a.sum_car_int[j] = f.sum_car_int[j];
  }
  plan->tile_bytes = tile;
  plan->table_bytes = align16(static_cast<size_t>(S) * 8 + static_cast<size_t>(a.ncar) * S * cells * 8 + static_cast<size_t>(S) * cells * 4 +
                              static_cast<size_t>(S) * a.hist_words * 4);
  // the direct-load kernel's signatures (agg_factored_direct_kernel)
  bool direct = a.nkeys >= 1 && a.nkeys <= 2 && a.ncell >= 1 && a.ncell <= 2 && a.nhist <= 1 && a.ncar <= 1;
  const int keyw = a.width[a.key_slot[0]];
  direct = direct && (keyw == 1 || keyw == 4);
  for (int k = 0; k < a.nkeys && direct; ++k) direct = a.width[a.key_slot[k]] == keyw;
  for (int q = 0; q < a.ncell && direct; ++q) direct = a.width[a.cell_slot[q]] == 1;
  for (int h = 0; h < a.nhist && direct; ++h) direct = a.width[a.hist_slot[h]] == 1;
  if (a.ncar == 1) direct = direct && a.car_type[0] == QSX_DOUBLE && a.car_int[0] == 0;
  plan->direct = direct && plan->table_bytes <= 60 * 1024;
  plan->keyw = keyw;
  FactoredDirectArgs &da = plan->da;
  if (plan->direct) {
    for (int k = 0; k < a.nkeys; ++k) {
      da.key[k] = a.col[a.key_slot[k]];
      da.key_shift[k] = a.key_shift[k];
    }
    for (int q = 0; q < a.ncell; ++q) {
      da.cellc[q] = static_cast<const unsigned char *>(a.col[a.cell_slot[q]]);
      da.cell_stride[q] = a.cell_stride[q];
      da.cell_radix[q] = a.cell_radix[q];
    }
    if (a.nhist == 1) {
      da.histc = static_cast<const unsigned char *>(a.col[a.hist_slot[0]]);
      da.hist_size = a.hist_size[0];
    }
    if (a.ncar == 1) da.carrier = static_cast<const double *>(a.col[a.car_slot[0]]);
    da.S = S;
    da.cells = a.cells;
    da.hist_words = a.hist_words;
  }
  return QSX_OK;
}
static int factored_workgroups_per_cu(size_t lds_bytes) {
  int per_cu = static_cast<int>((160 * 1024) / (lds_bytes + 512));
  per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);   // (measured on Q1, clear included: 2 / 3 / 4 / 5 / 6 workgroups per CU 1.51 / 1.43 / 1.45 / 1.49 / 1.44 ms per 600 M rows)
  if (const char *e = getenv("QSX_AGG_FACTORED_BLOCKS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
  return per_cu;
}
// The stripes a direct-load kernel reads of one block.  The kernel reads 8 rows per thread with one 8- or 16-byte load per
// stripe at whatever address the stripe starts: a reference block image keeps its stripes at offsets that are multiples of
// the block's tuple capacity (storage/CompressedColumnStoreTupleStorageSubBlock.cpp:71-160), aligned to nothing.  gfx950 under
// this stack serves unaligned global loads (tools/unaligned_probe.py: a DOUBLE stripe 1 byte off, code stripes 3 or 5
// bytes off — same groups, 0.34-0.36 instead of 0.33 ms per 100 M Q1 rows).  QSX_AGG_FACTORED_ALIGNED_ONLY=1 restores the
// 16-byte requirement (misaligned blocks then take the decoding kernels).
static bool factored_direct_aligned(const FactoredPlan &plan, const qsx_agg_state *st, const void *const *cols) {
  const FactoredStatic &f = st->factored;
  static const bool aligned_only = [] { const char *e = getenv("QSX_AGG_FACTORED_ALIGNED_ONLY"); return e != nullptr && e[0] == '1'; }();
  auto aligned16 = [](const void *p) { return !aligned_only || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  bool ok = true;
  for (int k = 0; k < st->dev.num_keys; ++k) ok = ok && aligned16(cols[st->dev.key_column[k]]);
  for (int q = 0; q < f.ncell; ++q) ok = ok && aligned16(cols[f.cell_col[q]]);
  for (int h = 0; h < f.nhist; ++h) ok = ok && aligned16(cols[f.hist_col[h]]);
  for (int k = 0; k < f.ncar; ++k) ok = ok && aligned16(cols[f.car_col[k]]);
  (void)plan;
  return ok;
}

// How a factored state's predicate becomes the call's filter: 1 = every term is on a plain stripe -> the tuned K1 kernels
// (qsx_select_cmp / _blocks, chained through the bitmap: ~6 TB/s of the terms' columns); 2 = some term is on a compressed
// attribute -> factored_predicate_kernel, which compares dictionary VALUES (540 G rows/s: slower than what the factored kernels
// save, so only with QSX_AGG_FACTORED_CODED_PREDICATES=1 — the tests keep it exact); 0 = that case by default: the decoding
// kernels keep the state.
static int factored_predicate_route(const DevConfig &d) {
  bool all_plain = true;
  for (int p = 0; p < d.num_pred; ++p) all_plain = all_plain && d.code_width[d.pred[p].column] == 0;
  if (all_plain) return 1;
  const char *e = getenv("QSX_AGG_FACTORED_CODED_PREDICATES");
  return e != nullptr && e[0] == '1' ? 2 : 0;
}

// The static part of the predicate pass's arguments (FactoredPredArgs): the state's terms, the type / width / coding of their columns.
static FactoredPredArgs factored_predicate_terms(const DevConfig &d) {
  FactoredPredArgs pa{};
  pa.num_pred = d.num_pred;
  for (int p = 0; p < d.num_pred; ++p) {
    const int col = d.pred[p].column;
    pa.pred[p] = d.pred[p];
    pa.type[p] = d.column_type[col];
    pa.width[p] = d.code_width[col] != 0 ? d.code_width[col] : d.column_width[col];
    pa.coded[p] = d.code_width[col] != 0 ? 1 : 0;
  }
  return pa;
}

static bool factored_generic_enabled() {
  const char *e = getenv("QSX_AGG_FACTORED_GENERIC");
  return e != nullptr && e[0] == '1';
}

// QSX_OK = the call was issued through the factored kernels; QSX_ERR_UNSUPPORTED = not this call (dictionary sizes unknown or
// too large, the cells do not fit LDS): the caller goes on with the decoding kernels; anything else is an error.
static int update_factored(qsx_agg_state *st, const void *const *cols, const void *const *dicts, const int32_t *entries, int64_t n,
                           const uint64_t *filter_dev, hipStream_t s) {
  const FactoredStatic &f = st->factored;
  if (!f.ok || !factored_enabled() || dicts == nullptr || entries == nullptr || n < factored_min_rows()) return QSX_ERR_UNSUPPORTED;
  const DevConfig &d = st->dev;
  const int pred_route = f.prepass ? factored_predicate_route(d) : 0;
  if (f.prepass && pred_route == 0) return QSX_ERR_UNSUPPORTED;
  FactoredPlan plan;
  int rc = factored_plan(st, cols, dicts, entries, filter_dev != nullptr || f.prepass, &plan);
  if (rc != QSX_OK) return rc;
  FactoredArgs &a = plan.a;
  FactoredCoefArgs &ca = plan.ca;
  const long long cells = a.cells;
  const size_t lds_bytes = plan.table_bytes + plan.tile_bytes;
  const bool direct = plan.direct && factored_direct_aligned(plan, st, cols) && factored_direct_signature(a, plan.keyw);
  // Who serves the call is decided HERE, before the predicate pass, the coefficient launch and the scratch (a state
  // whose plan is no direct-load signature paid for all three on every update and then handed the call to the decoding kernels,
  // which evaluate the predicate again).  The staged kernel needs at least two workgroups per CU, or its tile copies run under nothing.
  if (!direct && (!factored_generic_enabled() || lds_bytes > 64 * 1024)) return QSX_ERR_UNSUPPORTED;
  // coefficients: this call's dictionaries through the state's expression program
  CallScratch scratch(s);
  const size_t coef_bytes = static_cast<size_t>(d.num_sums) * (1 + a.ncar) * cells * 8, hcoef_bytes = static_cast<size_t>(d.num_sums) * kFacMaxDict * 8;
  const long long pred_tiles = (n + 1023) / 1024;
  const size_t pred_bytes = f.prepass ? static_cast<size_t>(pred_tiles) * 16 * 8 : 0;
  rc = scratch.reserve(CallScratch::padded(coef_bytes) + CallScratch::padded(hcoef_bytes) + CallScratch::padded(pred_bytes));
  if (rc != QSX_OK) return rc;
  ca.coef = static_cast<unsigned long long *>(scratch.take(coef_bytes));
  ca.hcoef = static_cast<unsigned long long *>(scratch.take(hcoef_bytes));
  if (f.prepass) {   // the state's predicate -> this call's filter
    uint64_t *bitmap = static_cast<uint64_t *>(scratch.take(pred_bytes));
    if (pred_route == 1) {
      for (int p = 0; p < d.num_pred; ++p) {
        const unsigned long long literal = d.pred[p].literal;   // (raw bits typed like the column: what qsx_select_cmp reads behind the pointer)
        rc = qsx_select_cmp(d.column_type[d.pred[p].column], cols[d.pred[p].column], n, d.pred[p].op, &literal, p == 0 ? filter_dev : bitmap, bitmap,
                            nullptr, reinterpret_cast<qsx_stream_t>(s));
        if (rc != QSX_OK) return rc;
      }
    } else {
      FactoredPredArgs pa = factored_predicate_terms(d);
      for (int p = 0; p < d.num_pred; ++p) {
        pa.col[p] = cols[d.pred[p].column];
        pa.dict[p] = d.code_width[d.pred[p].column] != 0 ? dicts[d.pred[p].column] : nullptr;
      }
      pa.filter_in = reinterpret_cast<const unsigned long long *>(filter_dev);
      pa.n = n;
      pa.out = reinterpret_cast<unsigned long long *>(bitmap);
      rc = launch_factored_predicate(pa, pred_tiles, s);
      if (rc != QSX_OK) return rc;
    }
    filter_dev = bitmap;
  }
  a.coef = ca.coef;
  a.hcoef = ca.hcoef;
  DevConfig dc = d;
  for (int i = 0; i < d.num_columns; ++i) dc.dicts[i] = d.code_width[i] != 0 ? dicts[i] : nullptr;
  rc = launch_factored_coef(dc, ca, s);
  if (rc != QSX_OK) return rc;
  const HashTableView g = st->hash_view();
  // ---- the common signatures: rows by direct loads, no staging (agg_factored_direct_kernel) ----
  if (direct) {
    const size_t direct_lds = plan.table_bytes;
    // the flush and the spill path read the whole plan: behind a pointer (a kernarg segment beyond 512 bytes has cost this
    // code base a factor before, DESIGN.md "Kernel arguments")
    const FactoredArgs *a_dev = static_cast<const FactoredArgs *>(staged_device_buffer(s, sizeof(FactoredArgs)));
    if (a_dev == nullptr) return QSX_ERR_OUT_OF_MEMORY;
    rc = staged_upload(s, &a, sizeof(FactoredArgs));
    if (rc != QSX_OK) return rc;
    const int per_cu = factored_workgroups_per_cu(direct_lds);
    const int64_t tiles = (n + kFacDirectTile - 1) / kFacDirectTile;
    const int grid = static_cast<int>(tiles < static_cast<int64_t>(per_cu) * kCUs ? tiles : static_cast<int64_t>(per_cu) * kCUs);
    if (!launch_factored_direct(a, a_dev, plan.da, plan.keyw, direct_lds, grid, n, filter_dev, g, s)) return QSX_ERR_UNSUPPORTED;   // (cannot happen: the signature was checked above)
    QSX_CHECK_LAUNCH();
    g_factored_launches.fetch_add(1, std::memory_order_relaxed);
    return QSX_OK;
  }
  // ---- any other signature: the staged kernel.  Slower than the decoding plan shapes as it stands (4.5 against 2.1 ms per
  // 600 M rows of Q1: its waves sit out every tile's copy and read their descriptors at run time), so it answers only when
  // asked (QSX_AGG_FACTORED_GENERIC=1, checked before any work of the call: the tests keep it exact for the day it is made fast). ----
  int per_cu = static_cast<int>((160 * 1024) / (lds_bytes + 512));
  per_cu = per_cu > 8 ? 8 : per_cu;
  if (const char *e = getenv("QSX_AGG_FACTORED_BLOCKS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
  rc = launch_factored_staged(a, lds_bytes, per_cu, n, filter_dev, g, s);
  if (rc != QSX_OK) return rc;
  g_factored_launches.fetch_add(1, std::memory_order_relaxed);
  return QSX_OK;
}

static long long partition_min_rows() {
  const char *e = getenv("QSX_AGG_PARTITION_MIN_ROWS");
  return e != nullptr ? atoll(e) : 4ll * 1024 * 1024;
}

// Mid-size group counts: partition the used columns on the key code (pieces aligned to 16 rows so that the update
// kernel's 16-byte DMA works on every column of every piece), then ONE launch in which workgroup family p aggregates
// piece p with a small LDS table.  No host synchronisation: the piece boundaries stay on the device.
static int update_partitioned(qsx_agg_state *st, const void *const *cols, int64_t n, hipStream_t s) {
  const int P = st->part_count;
  const int ncols = st->config.num_columns;
  constexpr int kAlignRows = 16;
  const int64_t padded = n + static_cast<int64_t>(kAlignRows) * P;
  // the scratch of this call: piece table, K9 workspace, one partitioned copy of every used column
  const size_t ws_bytes = partition_workspace_bytes(n, P);
  size_t total = CallScratch::padded(sizeof(int64_t) * 2 * P) + CallScratch::padded(ws_bytes);
  for (int c = 0; c < ncols; ++c) {
    if ((st->used_columns >> c) & 1u) total += CallScratch::padded(static_cast<size_t>(padded) * st->dev.column_width[c] + 16);
  }
  CallScratch scratch(s);
  int rc = scratch.reserve(total);
  if (rc != QSX_OK) return rc;
  int64_t *pieces = static_cast<int64_t *>(scratch.take(sizeof(int64_t) * 2 * P));
  void *ws = scratch.take(ws_bytes);
  const void *src[QSX_MAX_COLUMNS];
  void *dst[QSX_MAX_COLUMNS];
  void *part_cols[QSX_MAX_COLUMNS];
  int32_t widths[QSX_MAX_COLUMNS];
  int moved = 0;
  for (int c = 0; c < ncols; ++c) {
    part_cols[c] = nullptr;
    if (!((st->used_columns >> c) & 1u)) continue;
    part_cols[c] = scratch.take(static_cast<size_t>(padded) * st->dev.column_width[c] + 16);
    if (part_cols[c] == nullptr) return QSX_ERR_OUT_OF_MEMORY;
    src[moved] = cols[c];
    dst[moved] = part_cols[c];
    widths[moved] = st->dev.column_width[c];
    ++moved;
  }
  // the routing key is the packed key code, computed from the key columns inside K9 (never materialised)
  const void *key_cols[QSX_MAX_KEYS];
  for (int k = 0; k < st->dev.num_keys; ++k) key_cols[k] = cols[st->dev.key_column[k]];
  rc = partition_scatter_packed_keys(st->dev.num_keys, key_cols, st->dev.key_width, st->dev.key_shift, n, P, moved, src, widths,
                                     dst, pieces, ws, ws_bytes, s, kAlignRows);
  if (rc != QSX_OK) return rc;
  // n only sizes the grid here (an upper bound of every piece); the kernel reads its piece from `pieces`
  return update_slice(st, part_cols, nullptr, n, nullptr, st->part_slots, P, reinterpret_cast<const long long *>(pieces), s);
}

// More groups than one partition pass brings into LDS (agg_pieces.hpp): two K9 passes on digits of the mixing hash, then 4096
// pieces of disjoint groups, each through a workgroup's LDS table.  For the plans the consumer serves (two_level_plan).
// Per 100 M rows of an INT key and a DOUBLE, random keys, one pass / two levels: 10^5 groups 4.1 / 3.0 ms, 3 x 10^5 5.9 / 3.0,
// 10^6 9.6 / 3.0, 3 x 10^6 13.2 / 3.4, 10^7 20.9 / 4.9 (tools/agg_large_groups.py; profiles/README.md round 6).
static long long two_level_min_groups() {
  const char *e = getenv("QSX_AGG_TWO_LEVEL_MIN_GROUPS");   // 0 = never
  return e != nullptr ? atoll(e) : 100000ll;
}
// Clustered keys up to this many groups stay on the one-pass path (decide_two_level): 2.8 / 3.0 / 3.4 ms at 10^5 / 3 x 10^5 /
// 10^6 groups against 3.5 through two levels (a wave whose 64 rows are one partition's takes its places in the scatter with one
// LDS add — partition.hip — or the gap was a millisecond); 4.9 against 4.3 at 3 x 10^6, 10.8 against 4.9 at 10^7.
constexpr long long kTwoLevelWhateverTheOrder = 2000000;
// The rows of the call in progress have been through the state's predicate already (update_filtered_end_to_end behind the K1
// prepass of agg_update): the consumer of the pieces, which evaluates none, may serve them.
static thread_local bool tl_predicate_applied = false;
static bool two_level_plan(const qsx_agg_state *st) {
  const DevConfig &d = st->dev;
  if (st->dense || d.wide_words != 0 || (d.num_pred != 0 && !tl_predicate_applied) || d.num_null_cols != 0 || st->has_coded_columns || st->has_date_key) {
    return false;
  }
  if (d.num_keys < 1) return false;
  for (int j = 0; j < st->num_sums; ++j) {
    const DevSum &sum = d.sums[j];
    if (sum.count_valid != 0 || sum.null_mask != 0) return false;
    if (sum.arg.kind == QSX_OPD_TEMP) {
      // an expression (DOUBLE arithmetic): its values become a stripe in front of the passes (update_two_level)
      if (sum.is_int != 0 || sum.kind == kAccSumI64 || sum.arg.index < 0 || sum.arg.index >= QSX_MAX_TEMPS) return false;
      continue;
    }
    if (sum.arg.kind != QSX_OPD_COLUMN) return false;
    const int type = d.column_type[sum.arg.index];
    const bool integer = type == QSX_INT || type == QSX_LONG;
    const bool min_max = sum.kind == kAccMinI64 || sum.kind == kAccMaxI64;
    if (!((sum.kind == kAccSumF64 && type == QSX_DOUBLE) || (sum.kind == kAccSumI64 && integer) ||
          (min_max && (integer ? sum.is_int != 0 : (type == QSX_DOUBLE && sum.is_int == 0))))) {
      return false;
    }
  }
  return true;
}
// Slots of a piece's LDS table for `est` groups in all (load <= 1/3 when it fits), and whether a piece's groups fit it at all —
// a table that is full makes every further row of the piece walk it before it takes the global path.
static int two_level_slots(const qsx_agg_state *st, bool *fits) {
  const uint64_t per_piece = static_cast<uint64_t>(st->geometry_est) / kNumPieces + 1;
  uint64_t slots = next_pow2(per_piece * 3);
  if (slots < 64) slots = 64;
  while (slots > 64 && slots * 8 * (static_cast<uint64_t>(st->num_sums) + 2) > 128 * 1024) slots >>= 1;
  *fits = per_piece * 10 <= slots * 7;
  return static_cast<int>(slots);
}
// Clustered keys (lineitem on l_orderkey) are the one-pass path's best case — its LDS table works as a write-combining cache —
// and random keys its worst: decided once per state from how often the leading key repeats among the first 256 K rows it sees
// (1 = two levels, 2 = one pass; kTwoLevelWhateverTheOrder for the numbers).
static int decide_two_level(qsx_agg_state *st, const void *const *cols, int64_t n, hipStream_t s) {
  int mode = st->two_level_mode.load();
  if (mode != 0) return mode;
  const int col = st->dev.key_column[0], width = st->dev.key_width[0];
  const int sample = static_cast<int>(std::min<int64_t>(n, 256 * 1024));
  mode = 1;
  static const bool sampling = []() { const char *e = std::getenv("QSX_AGG_TWO_LEVEL_SAMPLE"); return e == nullptr || std::atoi(e) != 0; }();
  if (sampling && (width == 4 || width == 8) && cols[col] != nullptr && sample >= 4096) {
    CallScratch scratch(s);
    unsigned int equal = 0;
    if (scratch.reserve(CallScratch::padded(sizeof(unsigned int))) == QSX_OK) {
      unsigned int *counter = static_cast<unsigned int *>(scratch.take(sizeof(unsigned int)));
      bool ok = hipMemsetAsync(counter, 0, sizeof(unsigned int), s) == hipSuccess;
      if (ok && width == 8) hipLaunchKernelGGL(adjacent_equal_kernel<long long>, dim3(1), dim3(1024), 0, s, static_cast<const long long *>(cols[col]), sample, counter);
      if (ok && width == 4) hipLaunchKernelGGL(adjacent_equal_kernel<int32_t>, dim3(1), dim3(1024), 0, s, static_cast<const int32_t *>(cols[col]), sample, counter);
      ok = ok && hipGetLastError() == hipSuccess && hipMemcpyAsync(&equal, counter, sizeof(equal), hipMemcpyDeviceToHost, s) == hipSuccess &&
           hipStreamSynchronize(s) == hipSuccess;
      if (ok && static_cast<double>(equal) / (sample - 1) > 0.5) mode = 2;   // runs of two and more rows on average
      if (!ok) (void)hipGetLastError();
    }
  }
  st->two_level_mode.store(mode);
  return mode;
}
static std::atomic<long long> g_two_level_updates{0};
// Test hook (not part of include/qsx.h): update calls this process has served through the two-level partitioned aggregation.
extern "C" long long qsx_debug_agg_two_level_updates(void) { return g_two_level_updates.load(std::memory_order_relaxed); }

static int update_two_level(qsx_agg_state *st, const void *const *cols, int64_t n, hipStream_t s) {
  const int ncols = st->config.num_columns;
  const DevConfig &d = st->dev;
  // What travels through the two passes: the key columns, the columns the aggregates read directly, and — for an aggregate over
  // an expression — ONE stripe of the expression's values, computed in front of the passes by the K11 evaluator
  // (qsx_eval_expression: the same program, the same arithmetic as the fused form) instead of its operand columns.
  struct Moved {
    const void *src;
    int width;
    void *first, *second;
  };
  Moved items[QSX_MAX_COLUMNS + kMaxSums] = {};
  int moved = 0, item_of_column[QSX_MAX_COLUMNS], item_of_temp[QSX_MAX_TEMPS], item_of_sum[kMaxSums];
  for (int &v : item_of_column) v = -1;
  for (int &v : item_of_temp) v = -1;
  auto need_column = [&](int c) {
    if (item_of_column[c] < 0) {
      items[moved] = Moved{cols[c], d.column_width[c], nullptr, nullptr};
      item_of_column[c] = moved++;
    }
    return item_of_column[c];
  };
  for (int k = 0; k < d.num_keys; ++k) (void)need_column(d.key_column[k]);
  int num_temps = 0;
  for (int j = 0; j < st->num_sums; ++j) {
    const DevOperand &arg = d.sums[j].arg;
    if (arg.kind == QSX_OPD_COLUMN) {
      item_of_sum[j] = need_column(arg.index);
    } else {   // QSX_OPD_TEMP (two_level_plan)
      if (item_of_temp[arg.index] < 0) {
        items[moved] = Moved{nullptr, 8, nullptr, nullptr};
        item_of_temp[arg.index] = moved++;
        ++num_temps;
      }
      item_of_sum[j] = item_of_temp[arg.index];
    }
  }
  if (moved > QSX_MAX_COLUMNS) return QSX_ERR_UNSUPPORTED;
  const size_t ws_bytes = partition_workspace_bytes(n, kWave);
  size_t total = CallScratch::padded(sizeof(int64_t) * (kWave + 1)) + CallScratch::padded(sizeof(long long) * (kNumPieces + 1)) + CallScratch::padded(ws_bytes) +
                 static_cast<size_t>(num_temps) * CallScratch::padded(static_cast<size_t>(n) * 8 + 16);
  for (int i = 0; i < moved; ++i) total += 2 * CallScratch::padded(static_cast<size_t>(n) * items[i].width + 16);
  CallScratch scratch(s);
  int rc = scratch.reserve(total);
  if (rc != QSX_OK) return rc;
  int64_t *offsets = static_cast<int64_t *>(scratch.take(sizeof(int64_t) * (kWave + 1)));
  long long *bounds = static_cast<long long *>(scratch.take(sizeof(long long) * (kNumPieces + 1)));
  void *ws = scratch.take(ws_bytes);
  if (num_temps != 0) {
    // (columns the program cannot name — a CHAR key — stand in as INT stripes it never reads)
    const void *eval_cols[QSX_MAX_COLUMNS];
    int32_t eval_types[QSX_MAX_COLUMNS];
    const void *any = cols[d.key_column[0]];
    for (int c = 0; c < ncols; ++c) {
      const int type = st->config.column_type[c];
      const bool numeric = type == QSX_INT || type == QSX_LONG || type == QSX_FLOAT || type == QSX_DOUBLE;
      eval_cols[c] = numeric && cols[c] != nullptr ? cols[c] : any;
      eval_types[c] = numeric && cols[c] != nullptr ? type : QSX_INT;
    }
    for (int t = 0; t < QSX_MAX_TEMPS; ++t) {
      if (item_of_temp[t] < 0) continue;
      double *values = static_cast<double *>(scratch.take(static_cast<size_t>(n) * 8 + 16));
      if (values == nullptr) return QSX_ERR_OUT_OF_MEMORY;
      qsx_operand_t result{};
      result.kind = QSX_OPD_TEMP;
      result.index = t;
      rc = qsx_eval_expression(ncols, eval_cols, eval_types, st->config.num_instrs, st->config.instrs, st->config.consts, result, n, values,
                               reinterpret_cast<qsx_stream_t>(s));
      if (rc != QSX_OK) return rc;
      items[item_of_temp[t]].src = values;
    }
  }
  const void *src[QSX_MAX_COLUMNS], *first_const[QSX_MAX_COLUMNS];
  void *first[QSX_MAX_COLUMNS], *second[QSX_MAX_COLUMNS];
  int32_t widths[QSX_MAX_COLUMNS];
  for (int i = 0; i < moved; ++i) {
    const size_t bytes = static_cast<size_t>(n) * items[i].width + 16;
    items[i].first = scratch.take(bytes);
    items[i].second = scratch.take(bytes);
    if (items[i].first == nullptr || items[i].second == nullptr) return QSX_ERR_OUT_OF_MEMORY;
    src[i] = items[i].src;
    first[i] = items[i].first;
    first_const[i] = items[i].first;
    second[i] = items[i].second;
    widths[i] = items[i].width;
  }
  // the routing key is the packed key code, computed from the key columns inside K9 (never materialised): the low digit of
  // the top 12 hash bits first (any order), then the high digit keeping that order
  const void *key_cols[QSX_MAX_KEYS];
  for (int k = 0; k < d.num_keys; ++k) key_cols[k] = cols[d.key_column[k]];
  rc = partition_scatter_packed_digit(d.num_keys, key_cols, d.key_width, d.key_shift, n, 64 - kPieceBits, false, moved, src, widths, first, offsets,
                                      ws, ws_bytes, s);
  if (rc != QSX_OK) return rc;
  for (int k = 0; k < d.num_keys; ++k) key_cols[k] = items[item_of_column[d.key_column[k]]].first;
  rc = partition_scatter_packed_digit(d.num_keys, key_cols, d.key_width, d.key_shift, n, 64 - kPieceBits + 6, true, moved, first_const, widths, second,
                                      offsets, ws, ws_bytes, s);
  if (rc != QSX_OK) return rc;
  PieceArgs a{};
  a.num_keys = d.num_keys;
  for (int k = 0; k < d.num_keys; ++k) {
    a.key_col[k] = items[item_of_column[d.key_column[k]]].second;
    a.key_width[k] = d.key_width[k];
    a.key_shift[k] = d.key_shift[k];
  }
  a.num_sums = st->num_sums;
  for (int j = 0; j < st->num_sums; ++j) {
    a.sum_col[j] = items[item_of_sum[j]].second;
    a.sum_type[j] = d.sums[j].arg.kind == QSX_OPD_COLUMN ? d.column_type[d.sums[j].arg.index] : QSX_DOUBLE;
    a.sum_kind[j] = d.sums[j].kind;
  }
  a.bounds = bounds;
  a.n = n;
  bool fits = false;
  a.S = two_level_slots(st, &fits);
  rc = launch_agg_pieces(a, st->hash_view(), s);
  if (rc == QSX_OK) g_two_level_updates.fetch_add(1, std::memory_order_relaxed);
  return rc;
}

// Mid-size group count: group directory + one accumulator per group in LDS (derive_geometry).  Two launches
// (agg_common.hpp): the distinct key codes of these rows enter the directory, then the rows are aggregated.
// block_run != nullptr: the rows are a run of blocks (cols = the first block's stripes: only which ones exist matters).
static int update_directory(qsx_agg_state *st, const void *const *cols, const void *const *dicts, int64_t n, const uint64_t *filter_dev,
                            const uint64_t *const *nulls, const long long *block_run, int bounds_slot, hipStream_t s) {
  const bool runs = block_run != nullptr;
  const DirView dir = st->dir_view(bounds_slot);
  DevConfig dc = st->dev;
  for (int i = 0; i < st->config.num_columns; ++i) {
    dc.cols[i] = cols[i];
    dc.dicts[i] = (dicts != nullptr && dc.code_width[i] != 0) ? dicts[i] : nullptr;
  }
  for (int sl = 0; sl < dc.num_null_cols; ++sl) {
    dc.nulls[sl] = nulls != nullptr ? reinterpret_cast<const unsigned long long *>(nulls[dc.null_column[sl]]) : nullptr;
  }
  unsigned key_columns = 0;
  for (int k = 0; k < dc.num_keys; ++k) key_columns |= 1u << dc.key_column[k];
  for (int p = 0; p < dc.num_pred; ++p) key_columns |= 1u << dc.pred[p].column;
  int rc = launch_dir_build(dc, key_columns, n, filter_dev, dir, st->dir_gids, st->dir_calls.fetch_add(1), s, block_run);
  if (rc != QSX_OK) return rc;
  QSX_CHECK_LAUNCH();
  int variant = 0;
  const JitKernel *jk = nullptr;
  if (st->shape != nullptr && filter_dev == nullptr && dc.num_null_cols == 0) {
    st->rows_seen.fetch_add(n);
    rc = st->shape->launch_dir(cols, st->config.num_columns, n, st->hash_view(), dir, st->dir_gids, st->dir_nbuf, s, block_run);
  } else if ((dc.num_null_cols == 0 || !runs) &&
             (jk = state_jit_kernel(st, filter_dev != nullptr, false, st->dir_gids, 1, n, &variant, true, runs)) != nullptr &&
             st->jit_geometry[variant].dir_gids == st->dir_gids &&
             launch_jit_dir(st, jk, variant, cols, dc.dicts, n, filter_dev, dir, s, block_run, &dc) == QSX_OK) {
    rc = QSX_OK;   // run-time plan shape of the directory kernel
  } else {
    if (jk != nullptr) {   // the specialised kernel could not be launched: the interpreter from now on
      (void)hipGetLastError();
      std::lock_guard<std::mutex> jit_lock(st->jit_mutex);
      st->jit[variant] = nullptr;
    } else if (dc.num_null_cols != 0) {
      st->rows_seen.fetch_add(n);
    }
    QSX_DISPATCH_NS(st->num_sums, rc = launch_dir, dc, st->used_columns, n, filter_dev, st->hash_view(), dir, st->dir_gids, st->dir_nbuf, s,
                    block_run);
  }
  if (rc == QSX_OK) QSX_CHECK_LAUNCH();
  return rc;
}

static int agg_update(qsx_agg_state_t *st, const void *const *cols, const void *const *dicts, int64_t n,
                      const uint64_t *filter_dev, qsx_stream_t stream, const uint64_t *const *nulls = nullptr);
static std::atomic<long long> g_filtered_compactions{0};
// Test hook (not part of include/qsx.h): filtered calls whose surviving rows were compacted for the partition passes.
extern "C" long long qsx_debug_agg_filtered_compactions(void) { return g_filtered_compactions.load(std::memory_order_relaxed); }
// A filter bitmap in front of a state that partitions its input (a predicate's or LIP filter's TupleIdSequence over a
// group-by with many groups): the partition passes take whole columns, so a filtered call went through the tile kernels —
// NS + 1 global atomics per SURVIVING row.  The survivors are counted first (one pass over n / 8 bytes and a wait), then the
// used columns are compacted under the filter (K2) and the survivors take the stripe form — per 100 M rows at 10^6 groups
// (tools/agg_filtered_groups.py): selectivity 0.9 9.1 -> 3.2 ms, 0.5 6.2 -> 2.0, 0.1 5.9 -> 0.9.  Returns QSX_ERR_UNSUPPORTED
// when the call should go on as it came.
static bool filtered_call_compacts(const qsx_agg_state *st, int64_t n) {
  static const bool enabled = []() { const char *e = getenv("QSX_AGG_FILTER_COMPACT"); return e == nullptr || atoi(e) != 0; }();
  return enabled && !st->dense && st->part_count > 1 && !st->has_coded_columns && st->dev.num_null_cols == 0 && !st->has_date_key && st->dir_gids == 0 &&
         !st->factored.ok && n >= 2 * partition_min_rows();
}
static int update_filtered_end_to_end(qsx_agg_state_t *st, const void *const *cols, int64_t n, const uint64_t *filter_dev, qsx_stream_t stream,
                                      bool predicate_applied = false) {
  if (!filtered_call_compacts(st, n)) return QSX_ERR_UNSUPPORTED;
  hipStream_t s = as_stream(stream);
  const int ncols = st->config.num_columns;
  const void *used_cols[QSX_MAX_COLUMNS];
  void *out_cols[QSX_MAX_COLUMNS];
  int32_t widths[QSX_MAX_COLUMNS];
  int column_of[QSX_MAX_COLUMNS];
  int used = 0;
  size_t bytes = CallScratch::padded(16) + CallScratch::padded(qsx_compact_workspace_bytes(n) + 16);
  for (int c = 0; c < ncols; ++c) {
    if (!((st->used_columns >> c) & 1u)) continue;
    if (cols[c] == nullptr) return QSX_ERR_INVALID_ARGUMENT;
    used_cols[used] = cols[c];
    widths[used] = st->dev.column_width[c];
    column_of[used] = c;
    bytes += CallScratch::padded(static_cast<size_t>(n) * st->dev.column_width[c] + 16);
    ++used;
  }
  if (used == 0) return QSX_ERR_UNSUPPORTED;
  CallScratch scratch(s);
  int rc = scratch.reserve(bytes);
  if (rc != QSX_OK) return rc;
  int64_t *count_dev = static_cast<int64_t *>(scratch.take(16));
  const size_t ws_bytes = qsx_compact_workspace_bytes(n) + 16;
  void *ws = scratch.take(ws_bytes);
  rc = qsx_bitmap_count(filter_dev, n, count_dev, stream);
  if (rc != QSX_OK) return rc;
  int64_t survivors = 0;
  QSX_HIP_TRY(hipMemcpyAsync(&survivors, count_dev, sizeof(survivors), hipMemcpyDeviceToHost, s));
  QSX_HIP_TRY(hipStreamSynchronize(s));
  if (survivors == 0) return QSX_OK;
  // (few survivors too: the tile kernels read every row's columns to find them — 2.1 ms per 100 M rows at a selectivity of 0.01
  // against 0.5 for compaction + the update of 1 M rows)
  const void *stripes[QSX_MAX_COLUMNS] = {};
  for (int i = 0; i < used; ++i) {
    out_cols[i] = scratch.take(static_cast<size_t>(n) * widths[i] + 16);
    if (out_cols[i] == nullptr) return QSX_ERR_OUT_OF_MEMORY;
    stripes[column_of[i]] = out_cols[i];
  }
  rc = qsx_compact_gather(used, used_cols, widths, filter_dev, n, out_cols, count_dev, ws, ws_bytes, stream);
  if (rc != QSX_OK) return rc;
  g_filtered_compactions.fetch_add(1, std::memory_order_relaxed);
  const bool outer = tl_predicate_applied;
  tl_predicate_applied = predicate_applied;
  rc = agg_update(st, stripes, nullptr, survivors, nullptr, stream, nullptr);
  tl_predicate_applied = outer;
  return rc;
}
// The state's own predicate (plain INT / LONG / FLOAT / DOUBLE columns against literals) as a TupleIdSequence: a K1 pass per term,
// chained through the bitmap (and behind the call's filter) — what update_slice does in front of the family's kernels.
static int predicate_bitmap(const qsx_agg_state *st, const void *const *cols, int64_t n, const uint64_t *filter_dev, uint64_t *bitmap, qsx_stream_t stream) {
  const DevConfig &d = st->dev;
  for (int p = 0; p < d.num_pred; ++p) {
    const unsigned long long literal = d.pred[p].literal;
    const int rc = qsx_select_cmp(d.column_type[d.pred[p].column], cols[d.pred[p].column], n, d.pred[p].op, &literal, p == 0 ? filter_dev : bitmap, bitmap,
                                  nullptr, stream);
    if (rc != QSX_OK) return rc;
  }
  return QSX_OK;
}
static bool predicate_is_plain(const qsx_agg_state *st, const void *const *cols) {
  const DevConfig &d = st->dev;
  for (int p = 0; p < d.num_pred; ++p) {
    const int c = d.pred[p].column, type = d.column_type[c];
    if (cols[c] == nullptr || d.code_width[c] != 0 || (type != QSX_INT && type != QSX_LONG && type != QSX_FLOAT && type != QSX_DOUBLE)) return false;
  }
  return d.num_pred != 0;
}

static int agg_update(qsx_agg_state_t *st, const void *const *cols, const void *const *dicts, int64_t n,
                      const uint64_t *filter_dev, qsx_stream_t stream, const uint64_t *const *nulls) {
  QSX_REQUIRE_DEVICE();
  if (st == nullptr || n < 0 || (n > 0 && st->config.num_columns > 0 && cols == nullptr)) return QSX_ERR_INVALID_ARGUMENT;
  if (n == 0) return QSX_OK;
  hipStream_t s = as_stream(stream);
  if (dicts == nullptr && nulls == nullptr && !tl_predicate_applied && filtered_call_compacts(st, n) && predicate_is_plain(st, cols)) {
    // a state that filters inside its kernels AND partitions its input: the predicate becomes a bitmap first, the survivors are
    // compacted and — past their predicate — may take the two partition passes
    CallScratch pred_scratch(s);
    const size_t words = static_cast<size_t>((n + 63) / 64) + 1;
    int rc_pred = pred_scratch.reserve(CallScratch::padded(words * 8));
    if (rc_pred != QSX_OK) return rc_pred;
    uint64_t *bitmap = static_cast<uint64_t *>(pred_scratch.take(words * 8));
    rc_pred = predicate_bitmap(st, cols, n, filter_dev, bitmap, stream);
    if (rc_pred != QSX_OK) return rc_pred;
    rc_pred = update_filtered_end_to_end(st, cols, n, bitmap, stream, true);
    if (rc_pred != QSX_ERR_UNSUPPORTED) return rc_pred;
  } else if (filter_dev != nullptr && dicts == nullptr && nulls == nullptr) {
    const int rc_filtered = update_filtered_end_to_end(st, cols, n, filter_dev, stream, tl_predicate_applied);
    if (rc_filtered != QSX_ERR_UNSUPPORTED) return rc_filtered;
  }
  int rc = maybe_grow(st);
  if (rc != QSX_OK) return rc;
  std::shared_lock<std::shared_mutex> lock(st->table_mutex);
  // (the partitioned path scatters value columns: states over compressed attributes take the tile path)
  const int bounds_slot = !st->dense && st->dir_gids != 0 ? st->dir_bounds_slot(s) : -1;
  // a state over dictionary-coded attributes whose aggregates factor through the codes (agg_factored.hpp), and a call that
  // brought the dictionaries' sizes (qsx_agg_update_coded_sized)
  if (st->factored.ok && nulls == nullptr && bounds_slot < 0) {
    rc = update_factored(st, cols, dicts, tl_dictionary_entries, n, filter_dev, s);
    if (rc == QSX_OK) return publish_control(st, s);
    if (rc != QSX_ERR_UNSUPPORTED) return rc;
  }
  if (bounds_slot >= 0) {
    rc = update_directory(st, cols, dicts, n, filter_dev, nulls, nullptr, bounds_slot, s);
  } else
  // (and so do states over nullable columns: a null bitmap cannot be scattered like a value column)
  // (and states with a DATE key: K9 packs the key columns itself and would take the DateLit padding bytes along.  A wide
  // key is fine: K9 ORs the components at their in-word shifts — equal keys still meet in one piece, which is all the
  // pieces have to guarantee; the per-piece tables are keyed by the hashed code like everything else)
  if (!st->dense && st->part_count > 1 && filter_dev == nullptr && !st->has_coded_columns && st->dev.num_null_cols == 0 &&
      !st->has_date_key && n >= partition_min_rows()) {
    const long long two_level_from = two_level_min_groups();
    bool pieces_fit = false;
    if (two_level_from > 0 && st->geometry_est >= two_level_from && n >= 2 * partition_min_rows() && two_level_plan(st) &&
        (two_level_slots(st, &pieces_fit), pieces_fit) && (st->geometry_est >= kTwoLevelWhateverTheOrder || decide_two_level(st, cols, n, s) == 1)) {
      // the passes ping-pong the used columns through scratch (2 x their bytes): a call whose columns would take more than
      // 8 GiB of it goes slice by slice — every slice flushes its groups into the state like a call of its own
      size_t row_bytes = 0;
      for (int c = 0; c < st->config.num_columns; ++c) row_bytes += ((st->used_columns >> c) & 1u) ? st->dev.column_width[c] : 0;
      static const long long forced = []() { const char *e = getenv("QSX_AGG_TWO_LEVEL_SLICE_ROWS"); return e != nullptr ? atoll(e) : 0ll; }();
      const int64_t slice = forced > 0 ? forced : static_cast<int64_t>((size_t(8) << 30) / (2 * (row_bytes ? row_bytes : 1)));
      rc = QSX_OK;
      for (int64_t at = 0; at < n && rc == QSX_OK; at += slice) {
        const void *from[QSX_MAX_COLUMNS];
        for (int c = 0; c < st->config.num_columns; ++c) {
          from[c] = cols[c] != nullptr ? static_cast<const char *>(cols[c]) + static_cast<size_t>(at) * st->dev.column_width[c] : nullptr;
        }
        rc = update_two_level(st, at == 0 ? cols : from, std::min<int64_t>(slice, n - at), s);
      }
    } else {
      rc = update_partitioned(st, cols, n, s);
    }
  } else {
    rc = update_slice(st, cols, dicts, n, filter_dev, st->lds_slots, st->lds_ranges, nullptr, s, nulls);
  }
  if (rc != QSX_OK) return rc;
  return publish_control(st, s);
}

int qsx_agg_update(qsx_agg_state_t *st, const void *const *cols, int64_t n, const uint64_t *filter_dev,
                   qsx_stream_t stream) {
  // a state declared over compressed attributes needs the dictionaries / code stripes: qsx_agg_update_coded
  if (st != nullptr && st->has_coded_columns) return QSX_ERR_INVALID_ARGUMENT;
  return agg_update(st, cols, nullptr, n, filter_dev, stream);
}

int qsx_agg_update_nullable(qsx_agg_state_t *st, const void *const *cols, const uint64_t *const *null_bitmaps_dev, int64_t n,
                            const uint64_t *filter_dev, qsx_stream_t stream) {
  if (st != nullptr && st->has_coded_columns) return QSX_ERR_INVALID_ARGUMENT;
  if (st != nullptr && null_bitmaps_dev != nullptr) {
    for (int i = 0; i < st->config.num_columns; ++i) {
      if (null_bitmaps_dev[i] != nullptr && st->config.column_nullable[i] == 0) return QSX_ERR_INVALID_ARGUMENT;
    }
  }
  return agg_update(st, cols, nullptr, n, filter_dev, stream, null_bitmaps_dev);
}

// A run of blocks for a state whose groups need the partition passes (update_partitioned / update_two_level scatter whole
// columns): the run's used columns are laid end to end in scratch of the call first — one launch, 2 x the columns' bytes, 0.5 ms
// per 100 M rows of a 4-byte key and an 8-byte value — and take the stripe form from there.  Without it every row of a run paid
// NS + 1 global atomics: the operators hand the state runs of 1 M-row blocks (AggregationOperator::setBlocksPerWorkOrder), never
// a stripe long enough for the partition passes.
// segments: {source, destination, elements} per (block, used column); blockIdx.y = segment, blockIdx.x = its 16 Ki-element chunks.
static bool run_takes_partition_passes(const qsx_agg_state *st, int64_t total, bool any_filter, bool coded) {
  if (st->dense || st->part_count <= 1 || any_filter || coded || st->has_coded_columns || st->dev.num_null_cols != 0 || st->has_date_key) return false;
  if (st->dir_gids != 0 || st->factored.ok) return false;
  static const bool enabled = []() { const char *e = getenv("QSX_AGG_RUN_PARTITIONED"); return e == nullptr || atoi(e) != 0; }();
  return enabled && total >= partition_min_rows();
}
static std::atomic<long long> g_run_concats{0};
// Test hook (not part of include/qsx.h): runs of blocks that were laid end to end for the partition passes.
extern "C" long long qsx_debug_agg_run_concats(void) { return g_run_concats.load(std::memory_order_relaxed); }
static int update_run_end_to_end(qsx_agg_state_t *st, int num_blocks, const int64_t *block_rows, const void *const *block_cols, int64_t total,
                                 qsx_stream_t stream) {
  hipStream_t s = as_stream(stream);
  const int ncols = st->config.num_columns;
  size_t bytes = 0;
  int used = 0;
  for (int c = 0; c < ncols; ++c) {
    if ((st->used_columns >> c) & 1u) {
      bytes += CallScratch::padded(static_cast<size_t>(total) * st->dev.column_width[c] + 16);
      ++used;
    }
  }
  std::vector<long long> segments;   // grouped by element width: one launch per width that occurs
  int first_of_width[5] = {0, 0, 0, 0, 0};
  const int widths[4] = {1, 2, 4, 8};
  CallScratch scratch(s);
  int rc = scratch.reserve(bytes);
  if (rc != QSX_OK) return rc;
  const void *stripes[QSX_MAX_COLUMNS] = {};
  char *stripe_of[QSX_MAX_COLUMNS] = {};
  for (int c = 0; c < ncols; ++c) {
    if (!((st->used_columns >> c) & 1u)) continue;
    stripe_of[c] = static_cast<char *>(scratch.take(static_cast<size_t>(total) * st->dev.column_width[c] + 16));
    if (stripe_of[c] == nullptr) return QSX_ERR_OUT_OF_MEMORY;
    stripes[c] = stripe_of[c];
  }
  long long widest_segment = 0;
  for (int wi = 0; wi < 4; ++wi) {
    first_of_width[wi] = static_cast<int>(segments.size() / 3);
    for (int c = 0; c < ncols; ++c) {
      if (!((st->used_columns >> c) & 1u) || st->dev.column_width[c] != widths[wi]) continue;
      int64_t at = 0;
      for (int b = 0; b < num_blocks; ++b) {
        if (block_rows[b] == 0) continue;
        const void *src = block_cols[static_cast<size_t>(b) * ncols + c];
        if (src == nullptr) return QSX_ERR_INVALID_ARGUMENT;
        segments.push_back(static_cast<long long>(reinterpret_cast<uintptr_t>(src)));
        segments.push_back(static_cast<long long>(reinterpret_cast<uintptr_t>(stripe_of[c] + static_cast<size_t>(at) * widths[wi])));
        segments.push_back(block_rows[b]);
        widest_segment = std::max<long long>(widest_segment, block_rows[b]);
        at += block_rows[b];
      }
    }
  }
  first_of_width[4] = static_cast<int>(segments.size() / 3);
  const size_t table_bytes = segments.size() * sizeof(long long);
  const long long *segments_dev = static_cast<const long long *>(staged_device_buffer(s, table_bytes));
  if (segments_dev == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  rc = staged_upload(s, segments.data(), table_bytes);
  if (rc != QSX_OK) return rc;
  const unsigned chunks = static_cast<unsigned>(std::min<long long>((widest_segment + kConcatChunk - 1) / kConcatChunk, 64));
  for (int wi = 0; wi < 4; ++wi) {
    for (int first = first_of_width[wi]; first < first_of_width[wi + 1]; first += 65535) {
      const unsigned count = static_cast<unsigned>(std::min(first_of_width[wi + 1] - first, 65535));
      const dim3 grid(chunks, count);
      switch (widths[wi]) {
        case 1: hipLaunchKernelGGL(concat_segments_kernel<uint8_t>, grid, dim3(256), 0, s, segments_dev, first); break;
        case 2: hipLaunchKernelGGL(concat_segments_kernel<uint16_t>, grid, dim3(256), 0, s, segments_dev, first); break;
        case 4: hipLaunchKernelGGL(concat_segments_kernel<uint32_t>, grid, dim3(256), 0, s, segments_dev, first); break;
        default: hipLaunchKernelGGL(concat_segments_kernel<unsigned long long>, grid, dim3(256), 0, s, segments_dev, first); break;
      }
      QSX_CHECK_LAUNCH();
    }
  }
  g_run_concats.fetch_add(1, std::memory_order_relaxed);
  (void)used;
  return agg_update(st, stripes, nullptr, total, nullptr, stream, nullptr);
}

// The same for a run whose blocks all come with a filter (an AggregationWorkOrder under a predicate or LIP filter over a run): the
// survivors of the whole run compacted into one stripe per used column (K2 over the run), then the stripe form.
static int update_filtered_run_end_to_end(qsx_agg_state_t *st, int num_blocks, const int64_t *block_rows, const void *const *block_cols,
                                          const uint64_t *const *block_filters, int64_t total, qsx_stream_t stream) {
  hipStream_t s = as_stream(stream);
  const int ncols = st->config.num_columns;
  int32_t widths[QSX_MAX_COLUMNS];
  int column_of[QSX_MAX_COLUMNS];
  int used = 0;
  size_t bytes = CallScratch::padded(16);
  for (int c = 0; c < ncols; ++c) {
    if (!((st->used_columns >> c) & 1u)) continue;
    widths[used] = st->dev.column_width[c];
    column_of[used] = c;
    bytes += CallScratch::padded(static_cast<size_t>(total) * st->dev.column_width[c] + 16);
    ++used;
  }
  if (used == 0) return QSX_ERR_UNSUPPORTED;
  std::vector<int64_t> rows;
  std::vector<const void *> cols;
  std::vector<const uint64_t *> filters;
  for (int b = 0; b < num_blocks; ++b) {
    if (block_rows[b] == 0) continue;
    rows.push_back(block_rows[b]);
    filters.push_back(block_filters[b]);
    for (int i = 0; i < used; ++i) {
      const void *p = block_cols[static_cast<size_t>(b) * ncols + column_of[i]];
      if (p == nullptr) return QSX_ERR_INVALID_ARGUMENT;
      cols.push_back(p);
    }
  }
  const int64_t nb = static_cast<int64_t>(rows.size());
  const size_t ws_bytes = qsx_compact_blocks_workspace_bytes(nb, rows.data()) + 16;
  bytes += CallScratch::padded(ws_bytes);
  CallScratch scratch(s);
  int rc = scratch.reserve(bytes);
  if (rc != QSX_OK) return rc;
  int64_t *count_dev = static_cast<int64_t *>(scratch.take(16));
  void *ws = scratch.take(ws_bytes);
  void *out_cols[QSX_MAX_COLUMNS];
  const void *stripes[QSX_MAX_COLUMNS] = {};
  for (int i = 0; i < used; ++i) {
    out_cols[i] = scratch.take(static_cast<size_t>(total) * widths[i] + 16);
    if (out_cols[i] == nullptr) return QSX_ERR_OUT_OF_MEMORY;
    stripes[column_of[i]] = out_cols[i];
  }
  rc = qsx_compact_gather_blocks(used, widths, nb, rows.data(), cols.data(), filters.data(), nullptr, out_cols, nullptr, count_dev, ws, ws_bytes, stream);
  if (rc != QSX_OK) return rc;
  int64_t survivors = 0;
  QSX_HIP_TRY(hipMemcpyAsync(&survivors, count_dev, sizeof(survivors), hipMemcpyDeviceToHost, s));
  QSX_HIP_TRY(hipStreamSynchronize(s));
  g_filtered_compactions.fetch_add(1, std::memory_order_relaxed);
  if (survivors == 0) return QSX_OK;
  return agg_update(st, stripes, nullptr, survivors, nullptr, stream, nullptr);
}

// block_dicts: the dictionaries of a state over compressed attributes, [block * num_columns + column] (nullptr otherwise).
static int agg_update_blocks(qsx_agg_state_t *st, int num_blocks, const int64_t *block_rows, const void *const *block_cols,
                             const void *const *block_dicts, const uint64_t *const *block_filters, qsx_stream_t stream,
                             const int32_t *block_entries = nullptr) {
  QSX_REQUIRE_DEVICE();
  if (st == nullptr || num_blocks < 0 || (num_blocks > 0 && (block_rows == nullptr || block_cols == nullptr))) return QSX_ERR_INVALID_ARGUMENT;
  // (nullable inputs carry per-block null bitmaps: one call per block, qsx_agg_update_nullable)
  if (st->dev.num_null_cols != 0) return QSX_ERR_UNSUPPORTED;
  if (st->has_coded_columns != (block_dicts != nullptr)) return st->has_coded_columns ? QSX_ERR_INVALID_ARGUMENT : QSX_ERR_UNSUPPORTED;
  const int ncols = st->config.num_columns;
  // the table (agg_common.hpp): header, first_tile for 1024- and 512-row tiles, rows, column pointers, filters
  std::vector<long long> table(kBlockRunHeaderWords);
  std::vector<long long> tiles1024, tiles512, rows, filters;
  std::vector<long long> cols, dicts;
  {   // (a run of a few thousand blocks: the table is ~35 words per block; growing the vectors word by word showed in the call's time)
    const size_t nb_max = static_cast<size_t>(num_blocks);
    tiles1024.reserve(nb_max + 1);
    tiles512.reserve(nb_max + 1);
    rows.reserve(nb_max);
    filters.reserve(nb_max);
    cols.reserve(nb_max * QSX_MAX_COLUMNS);
    if (block_dicts != nullptr) dicts.reserve(nb_max * QSX_MAX_COLUMNS);
  }
  int64_t total = 0;
  bool any_filter = false;
  const void *first_cols[QSX_MAX_COLUMNS] = {};
  const uint64_t *first_filter = nullptr;
  const bool canonical_table = block_dicts == nullptr && family_serves_runs(st);   // (the same test the update makes: one decision)
  for (int b = 0; b < num_blocks; ++b) {
    if (block_rows[b] < 0) return QSX_ERR_INVALID_ARGUMENT;
    if (block_rows[b] == 0) continue;
    if (rows.empty()) {
      for (int c = 0; c < ncols; ++c) first_cols[c] = block_cols[static_cast<size_t>(b) * ncols + c];
    }
    tiles1024.push_back(tiles1024.empty() ? 0 : tiles1024.back() + (rows.back() + 1023) / 1024);
    tiles512.push_back(tiles512.empty() ? 0 : tiles512.back() + (rows.back() + 511) / 512);
    rows.push_back(block_rows[b]);
    for (int c = 0; c < QSX_MAX_COLUMNS; ++c) {
      // (a state served by the AOT family: the table's column c is the family's canonical column c, agg_family.hpp)
      const int from = canonical_table ? (c < st->family_num_columns ? st->family_cols[c] : -1) : (c < ncols ? c : -1);
      const void *p = from >= 0 ? block_cols[static_cast<size_t>(b) * ncols + from] : nullptr;
      if (from >= 0 && ((st->used_columns >> from) & 1u) != 0 && p == nullptr) return QSX_ERR_INVALID_ARGUMENT;
      cols.push_back(static_cast<long long>(reinterpret_cast<uintptr_t>(p)));
      if (block_dicts != nullptr) {
        const void *d = c < ncols && st->dev.code_width[c] != 0 ? block_dicts[static_cast<size_t>(b) * ncols + c] : nullptr;
        dicts.push_back(static_cast<long long>(reinterpret_cast<uintptr_t>(d)));
      }
    }
    const uint64_t *f = block_filters != nullptr ? block_filters[b] : nullptr;
    if (f != nullptr && first_filter == nullptr) first_filter = f;
    any_filter = any_filter || f != nullptr;
    filters.push_back(static_cast<long long>(reinterpret_cast<uintptr_t>(f)));
    total += block_rows[b];
  }
  if (rows.empty()) return QSX_OK;
  if (run_takes_partition_passes(st, total, any_filter, block_dicts != nullptr)) {
    return update_run_end_to_end(st, num_blocks, block_rows, block_cols, total, stream);
  }
  if (any_filter && total >= 2 * partition_min_rows() && run_takes_partition_passes(st, total, false, block_dicts != nullptr)) {
    bool every_block_filtered = true;
    for (int b = 0; b < num_blocks; ++b) every_block_filtered = every_block_filtered && (block_rows[b] == 0 || block_filters[b] != nullptr);
    static const bool compact = []() { const char *e = getenv("QSX_AGG_FILTER_COMPACT"); return e == nullptr || atoi(e) != 0; }();
    if (every_block_filtered && compact) return update_filtered_run_end_to_end(st, num_blocks, block_rows, block_cols, block_filters, total, stream);
  }
  tiles1024.push_back(tiles1024.back() + (rows.back() + 1023) / 1024);
  tiles512.push_back(tiles512.back() + (rows.back() + 511) / 512);
  const size_t nb = rows.size();
  hipStream_t s = as_stream(stream);
  int rc = maybe_grow(st);
  if (rc != QSX_OK) return rc;
  std::shared_lock<std::shared_mutex> lock(st->table_mutex);
  const int bounds_slot = !st->dense && st->dir_gids != 0 ? st->dir_bounds_slot(s) : -1;
  // ---- aggregates factored through the dictionary codes (agg_factored.hpp), block by block: the caller brought every block's
  // dictionary sizes, the plan is one of the direct-load kernel's signatures and every stripe is 16-byte aligned ----
  FactoredPlan fplan;
  bool factored = false;
  std::vector<long long> fac_first_tile;
  std::vector<int32_t> fac_entries;
  const int pred_route = st->factored.ok && st->factored.prepass ? factored_predicate_route(st->dev) : 0;
  if (block_entries != nullptr && block_dicts != nullptr && st->factored.ok && bounds_slot < 0 && factored_enabled() && total >= factored_min_rows() &&
      !(st->factored.prepass && pred_route == 0)) {
    const FactoredStatic &f = st->factored;
    int32_t radix[QSX_MAX_COLUMNS] = {};
    const void *first_dicts[QSX_MAX_COLUMNS] = {};
    fac_entries.assign(nb * QSX_MAX_COLUMNS, 0);
    factored = true;
    size_t at = 0;
    for (int b = 0; b < num_blocks && factored; ++b) {
      if (block_rows[b] == 0) continue;
      const void *const *bc = block_cols + static_cast<size_t>(b) * ncols;
      auto dictionary_column = [&](int col) {
        const int32_t e = block_entries[static_cast<size_t>(b) * ncols + col];
        const void *dict = block_dicts[static_cast<size_t>(b) * ncols + col];
        if (e < 0) rc = QSX_ERR_INVALID_ARGUMENT;
        if (dict == nullptr || e < 1 || e > kFacMaxDict) factored = false;
        radix[col] = e > radix[col] ? e : radix[col];
        if (first_dicts[col] == nullptr) first_dicts[col] = dict;
        fac_entries[at * QSX_MAX_COLUMNS + col] = e;
      };
      for (int q = 0; q < f.ncell; ++q) dictionary_column(f.cell_col[q]);
      for (int h = 0; h < f.nhist; ++h) dictionary_column(f.hist_col[h]);
      factored = factored && factored_direct_aligned(fplan, st, bc);
      const uint64_t *bf = block_filters != nullptr ? block_filters[b] : nullptr;
      factored = factored && (reinterpret_cast<uintptr_t>(bf) & 7) == 0;
      ++at;
    }
    if (rc != QSX_OK) return rc;
    if (factored) {
      rc = factored_plan(st, first_cols, first_dicts, radix, any_filter || f.prepass, &fplan);
      if (rc != QSX_OK && rc != QSX_ERR_UNSUPPORTED) return rc;
      factored = rc == QSX_OK && fplan.direct;
    }
    if (factored) {
      fac_first_tile.push_back(0);
      for (size_t b = 0; b < nb; ++b) fac_first_tile.push_back(fac_first_tile.back() + (rows[b] + kFacDirectTile - 1) / kFacDirectTile);
    }
  }
  // the coefficient tables of every block
  CallScratch fac_scratch(s);
  const size_t coef_words = factored ? static_cast<size_t>(st->dev.num_sums) * (1 + fplan.a.ncar) * fplan.a.cells : 0,
               hcoef_words = factored ? static_cast<size_t>(st->dev.num_sums) * kFacMaxDict : 0;
  if (factored && nb * (coef_words + hcoef_words) * 8 > (size_t(256) << 20)) factored = false;   // (a run of very many blocks over very many cells)
  // the state's predicate becomes every block's filter bitmap first (FactoredPredArgs): 16 words per 1024-row tile of the run
  const bool prepass = factored && st->factored.prepass;
  const size_t pred_bytes = prepass ? static_cast<size_t>(tiles1024.back()) * 16 * 8 : 0;
  unsigned long long *pred_bitmaps = nullptr;
  if (factored) {
    rc = fac_scratch.reserve(CallScratch::padded(nb * coef_words * 8) + CallScratch::padded(nb * hcoef_words * 8) + CallScratch::padded(pred_bytes));
    if (rc != QSX_OK) return rc;
    fplan.ca.coef = static_cast<unsigned long long *>(fac_scratch.take(nb * coef_words * 8));
    fplan.ca.hcoef = static_cast<unsigned long long *>(fac_scratch.take(nb * hcoef_words * 8));
    if (prepass) pred_bitmaps = static_cast<unsigned long long *>(fac_scratch.take(pred_bytes));
    fplan.a.coef = fplan.ca.coef;      // (block 0's: the kernel takes every block's from FactoredRunArgs)
    fplan.a.hcoef = fplan.ca.hcoef;
  }
  const size_t off1024 = kBlockRunHeaderWords, off512 = off1024 + nb + 1, off_rows = off512 + nb + 1, off_cols = off_rows + nb,
               off_filters = off_cols + nb * QSX_MAX_COLUMNS, off_dicts = off_filters + nb,
               off_fac_tiles = off_dicts + (block_dicts != nullptr ? nb * QSX_MAX_COLUMNS : 0),
               off_fac_entries = off_fac_tiles + (factored ? nb + 1 : 0),
               off_fac_args = off_fac_entries + (factored ? nb * QSX_MAX_COLUMNS / 2 : 0),
               off_fac_infilters = off_fac_args + (factored ? (sizeof(FactoredArgs) + 7) / 8 : 0),
               words = off_fac_infilters + (prepass && any_filter ? nb : 0);
  static_assert(QSX_MAX_COLUMNS % 2 == 0, "two int32 dictionary sizes per table word");
  table.resize(words);
  std::copy(dicts.begin(), dicts.end(), table.begin() + off_dicts);
  std::copy(tiles1024.begin(), tiles1024.end(), table.begin() + off1024);
  std::copy(tiles512.begin(), tiles512.end(), table.begin() + off512);
  std::copy(rows.begin(), rows.end(), table.begin() + off_rows);
  std::copy(cols.begin(), cols.end(), table.begin() + off_cols);
  std::copy(filters.begin(), filters.end(), table.begin() + off_filters);
  if (prepass) {   // the accumulate kernel's filters are the predicate pass's bitmaps; the caller's own go to that pass
    if (any_filter) std::copy(filters.begin(), filters.end(), table.begin() + off_fac_infilters);
    for (size_t b = 0; b < nb; ++b) {
      table[off_filters + b] = static_cast<long long>(reinterpret_cast<uintptr_t>(pred_bitmaps + static_cast<size_t>(tiles1024[b]) * 16));
    }
  }
  if (factored) {
    std::copy(fac_first_tile.begin(), fac_first_tile.end(), table.begin() + off_fac_tiles);
    std::memcpy(&table[off_fac_entries], fac_entries.data(), fac_entries.size() * sizeof(int32_t));
    std::memcpy(&table[off_fac_args], &fplan.a, sizeof(FactoredArgs));
  }
  if (prepass && pred_route == 1) {
    // every term on a plain stripe: the K1 run kernels, chained through the blocks' bitmaps — BEFORE this call's own table goes
    // into the stream's staging buffer (they stage theirs there)
    std::vector<int64_t> rows64(rows.begin(), rows.end());
    std::vector<const void *> term_cols(nb);
    std::vector<const uint64_t *> in_filters(nb);
    std::vector<uint64_t *> out_bitmaps(nb);
    for (size_t b = 0; b < nb; ++b) {
      in_filters[b] = reinterpret_cast<const uint64_t *>(static_cast<uintptr_t>(filters[b]));
      out_bitmaps[b] = reinterpret_cast<uint64_t *>(pred_bitmaps + static_cast<size_t>(tiles1024[b]) * 16);
    }
    for (int p = 0; p < st->dev.num_pred; ++p) {
      const int col = st->dev.pred[p].column;
      for (size_t b = 0; b < nb; ++b) term_cols[b] = reinterpret_cast<const void *>(static_cast<uintptr_t>(cols[b * QSX_MAX_COLUMNS + col]));
      const unsigned long long literal = st->dev.pred[p].literal;
      rc = qsx_select_cmp_blocks(st->dev.column_type[col], static_cast<int64_t>(nb), rows64.data(), term_cols.data(), st->dev.pred[p].op, &literal,
                                 p == 0 ? (any_filter ? in_filters.data() : nullptr) : const_cast<const uint64_t *const *>(out_bitmaps.data()),
                                 out_bitmaps.data(), nullptr, stream);
      if (rc != QSX_OK) return rc;
    }
  }
  long long *dev_table = static_cast<long long *>(staged_device_buffer(s, words * sizeof(long long)));
  if (dev_table == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  table[0] = kBlockRunMagic;
  table[1] = static_cast<long long>(nb);
  table[2] = static_cast<long long>(off1024);
  table[3] = static_cast<long long>(off512);
  table[4] = static_cast<long long>(off_rows);
  table[5] = static_cast<long long>(off_cols);
  table[6] = any_filter || prepass ? static_cast<long long>(off_filters) : 0;
  table[7] = 0;
  table[8] = block_dicts != nullptr ? static_cast<long long>(off_dicts) : 0;
  {   // equal-sized blocks (a relation's blocks all hold the same number of tuples but the last): no search per tile
    const long long per_block = (rows[0] + 1023) / 1024;
    bool uniform = per_block > 0;
    for (size_t b = 0; b + 1 < nb && uniform; ++b) uniform = (rows[b] + 1023) / 1024 == per_block;
    if (uniform && nb > 1 && (rows[nb - 1] + 1023) / 1024 > per_block) uniform = false;
    if (uniform) table[7] = per_block;
  }
  rc = staged_upload(s, table.data(), words * sizeof(long long));
  if (rc != QSX_OK) return rc;
  if (factored) {
    // (one staged table per call and stream: the run table, the kernel's tile index, the dictionary sizes and the plan travel together)
    fplan.ca.run_dicts = dev_table + off_dicts;
    fplan.ca.run_entries = reinterpret_cast<const int *>(dev_table + off_fac_entries);
    fplan.ca.coef_words = static_cast<long long>(coef_words);
    fplan.ca.hcoef_words = static_cast<long long>(hcoef_words);
    rc = launch_factored_coef(st->dev, fplan.ca, s, static_cast<int>(nb));
    if (rc != QSX_OK) return rc;
    if (prepass && pred_route == 2) {
      FactoredPredArgs pa = factored_predicate_terms(st->dev);
      pa.run = dev_table;
      pa.filters_in = any_filter ? dev_table + off_fac_infilters : nullptr;
      pa.out = pred_bitmaps;
      rc = launch_factored_predicate(pa, tiles1024.back(), s);
      if (rc != QSX_OK) return rc;
    }
    const FactoredStatic &f = st->factored;
    FactoredRunArgs ra{};
    ra.run = dev_table;
    ra.first_tile = dev_table + off_fac_tiles;
    ra.total_tiles = fac_first_tile.back();
    ra.num_blocks = static_cast<int>(nb);
    for (int k = 0; k < st->dev.num_keys && k < 2; ++k) ra.key_col[k] = st->dev.key_column[k];
    for (int q = 0; q < f.ncell && q < 2; ++q) ra.cell_col[q] = f.cell_col[q];
    ra.hist_col = f.nhist > 0 ? f.hist_col[0] : 0;
    ra.car_col = f.ncar > 0 ? f.car_col[0] : 0;
    ra.coef = fplan.ca.coef;
    ra.hcoef = fplan.ca.hcoef;
    ra.coef_words = static_cast<long long>(coef_words);
    ra.hcoef_words = static_cast<long long>(hcoef_words);
    const int per_cu = factored_workgroups_per_cu(fplan.table_bytes);
    // (a workgroup flushes its cells once per block it touches: no more workgroups than tiles / 4)
    const int64_t want = ra.total_tiles / 4 > 0 ? ra.total_tiles / 4 : 1;
    const int grid = static_cast<int>(want < static_cast<int64_t>(per_cu) * kCUs ? want : static_cast<int64_t>(per_cu) * kCUs);
    const FactoredArgs *a_dev = reinterpret_cast<const FactoredArgs *>(dev_table + off_fac_args);
    // (kFilter of the kernel = some block of the run has a filter: any non-null pointer says so)
    const uint64_t *filter_flag = prepass ? reinterpret_cast<const uint64_t *>(pred_bitmaps) : (any_filter ? first_filter : nullptr);
    if (launch_factored_direct(fplan.a, a_dev, fplan.da, fplan.keyw, fplan.table_bytes, grid, total, filter_flag, st->hash_view(), s, &ra)) {
      QSX_CHECK_LAUNCH();
      g_factored_launches.fetch_add(1, std::memory_order_relaxed);
      return publish_control(st, s);
    }
  }
  // one launch over the tiles of all blocks (a mid-size group count: the two passes of the group directory; the partition
  // pass wants one stripe per column, its group counts take the hash-range families here)
  if (bounds_slot >= 0) {
    rc = update_directory(st, first_cols, nullptr, total, any_filter ? first_filter : nullptr, nullptr, dev_table, bounds_slot, s);
    if (rc != QSX_OK) return rc;
    return publish_control(st, s);
  }
  rc = update_slice(st, first_cols, nullptr, total, any_filter ? first_filter : nullptr, st->lds_slots, st->lds_ranges, nullptr, s, nullptr,
                    dev_table, rows[0]);
  if (rc != QSX_OK) return rc;
  return publish_control(st, s);
}

int qsx_agg_update_blocks(qsx_agg_state_t *st, int num_blocks, const int64_t *block_rows, const void *const *block_cols,
                          const uint64_t *const *block_filters, qsx_stream_t stream) {
  return agg_update_blocks(st, num_blocks, block_rows, block_cols, nullptr, block_filters, stream);
}

int qsx_agg_update_coded_blocks(qsx_agg_state_t *st, int num_blocks, const int64_t *block_rows, const void *const *block_cols,
                                const void *const *block_dictionaries, const uint64_t *const *block_filters, qsx_stream_t stream) {
  if (num_blocks > 0 && block_dictionaries == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  static const void *const no_dictionaries[1] = {nullptr};
  return agg_update_blocks(st, num_blocks, block_rows, block_cols, num_blocks > 0 ? block_dictionaries : no_dictionaries, block_filters, stream);
}

int qsx_agg_update_coded_blocks_sized(qsx_agg_state_t *st, int num_blocks, const int64_t *block_rows, const void *const *block_cols,
                                      const void *const *block_dictionaries, const int32_t *block_dictionary_entries,
                                      const uint64_t *const *block_filters, qsx_stream_t stream) {
  if (num_blocks > 0 && block_dictionaries == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  static const void *const no_dictionaries[1] = {nullptr};
  return agg_update_blocks(st, num_blocks, block_rows, block_cols, num_blocks > 0 ? block_dictionaries : no_dictionaries, block_filters, stream,
                           num_blocks > 0 ? block_dictionary_entries : nullptr);
}

int qsx_agg_update_coded(qsx_agg_state_t *st, const void *const *cols, const void *const *dictionaries_dev, int64_t n,
                         const uint64_t *filter_dev, qsx_stream_t stream) {
  return agg_update(st, cols, dictionaries_dev, n, filter_dev, stream);
}

int qsx_agg_update_coded_sized(qsx_agg_state_t *st, const void *const *cols, const void *const *dictionaries_dev,
                               const int32_t *dictionary_entries, int64_t n, const uint64_t *filter_dev, qsx_stream_t stream) {
  if (st != nullptr && dictionary_entries != nullptr) {
    for (int i = 0; i < st->config.num_columns; ++i) {
      if (dictionary_entries[i] < 0) return QSX_ERR_INVALID_ARGUMENT;
    }
  }
  tl_dictionary_entries = dictionaries_dev != nullptr ? dictionary_entries : nullptr;
  const int rc = agg_update(st, cols, dictionaries_dev, n, filter_dev, stream);
  tl_dictionary_entries = nullptr;
  return rc;
}

// Test hook (not part of include/qsx.h): where the run-time plan shape of a state stands.
// -2: not requested yet, 0: compiling in the background, 1: in use, -1: given up (off / failed).
int qsx_debug_agg_jit_state(qsx_agg_state_t *st, int with_filter) {
  if (st == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  const int v = with_filter != 0 ? 1 : 0;
  std::lock_guard<std::mutex> lock(st->jit_mutex);
  if (st->jit_tried[v]) return st->jit[v] != nullptr ? 1 : -1;
  if (st->jit_request[v] == nullptr) return -2;
  return jit_request_state(st->jit_request[v], nullptr) == 0 ? 0 : (jit_request_state(st->jit_request[v], nullptr) == 1 ? 1 : -1);
}

int qsx_agg_mark_existence(qsx_agg_state_t *st, int key_type, const void *keys_dev, int64_t n, const uint64_t *filter_dev,
                           qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (st == nullptr || n < 0 || (n > 0 && keys_dev == nullptr)) return QSX_ERR_INVALID_ARGUMENT;
  if (!st->dense || (key_type != QSX_INT && key_type != QSX_LONG)) return QSX_ERR_UNSUPPORTED;
  if (n == 0) return QSX_OK;
  hipStream_t s = as_stream(stream);
  const int grid = grid_for(n, kABlock * 4);
  if (key_type == QSX_INT) {
    hipLaunchKernelGGL(mark_existence_kernel<int32_t>, dim3(grid), dim3(kABlock), 0, s, static_cast<const int32_t *>(keys_dev), n,
                       filter_dev, st->image, static_cast<long long>(st->config.num_entries));
  } else {
    hipLaunchKernelGGL(mark_existence_kernel<long long>, dim3(grid), dim3(kABlock), 0, s, static_cast<const long long *>(keys_dev), n,
                       filter_dev, st->image, static_cast<long long>(st->config.num_entries));
  }
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_agg_state_export_bytes(qsx_agg_state_t *st, size_t *out_bytes, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (st == nullptr || out_bytes == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  // a table that grew has a bigger image: bring it to rest first (spilled rows folded in), then the size is final
  // until the next update
  int rc = settle(st, as_stream(stream));
  if (rc != QSX_OK) return rc;
  std::shared_lock<std::shared_mutex> lock(st->table_mutex);
  *out_bytes = st->image_bytes;
  return QSX_OK;
}

int qsx_agg_state_image_layout(qsx_agg_state_t *st, int *out_dense, int64_t *out_header_words, int64_t *out_words_per_column,
                               int *out_num_columns, int32_t *out_column_kinds, int kinds_capacity) {
  if (st == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  std::shared_lock<std::shared_mutex> lock(st->table_mutex);
  if (out_dense != nullptr) *out_dense = st->dense ? 1 : 0;
  // dense: [existence words][num_cols columns of num_entries words]; hash: [cap + 1 key words][num_cols columns of cap + 1]
  if (out_header_words != nullptr) *out_header_words = st->dense ? st->exist_words : static_cast<int64_t>(st->cap + 1);
  if (out_words_per_column != nullptr) *out_words_per_column = st->dense ? st->config.num_entries : static_cast<int64_t>(st->cap + 1);
  if (out_num_columns != nullptr) *out_num_columns = st->num_cols;
  if (out_column_kinds != nullptr) {
    if (kinds_capacity < st->num_cols) return QSX_ERR_CAPACITY;
    for (int c = 0; c < st->num_cols; ++c) out_column_kinds[c] = st->col_kinds.kind[c];
  }
  return QSX_OK;
}

int qsx_agg_state_export(qsx_agg_state_t *st, void *out_dev, size_t capacity_bytes, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (st == nullptr || out_dev == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  int rc = settle(st, as_stream(stream));
  if (rc != QSX_OK) return rc;
  std::shared_lock<std::shared_mutex> lock(st->table_mutex);
  if (capacity_bytes < st->image_bytes) return QSX_ERR_CAPACITY;
  QSX_HIP_TRY(hipMemcpyAsync(out_dev, st->image, st->image_bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
  return QSX_OK;
}

int qsx_agg_state_import_merge(qsx_agg_state_t *dst, const void *image_dev, size_t image_bytes, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (dst == nullptr || image_dev == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  const unsigned long long *image = static_cast<const unsigned long long *>(image_dev);
  hipStream_t s = as_stream(stream);
  int rc = maybe_grow(dst);
  if (rc != QSX_OK) return rc;
  std::shared_lock<std::shared_mutex> lock(dst->table_mutex);
  if (dst->dense) {
    if (image_bytes != dst->image_bytes) return QSX_ERR_INVALID_ARGUMENT;
    const long long total = dst->exist_words + static_cast<long long>(dst->num_cols) * dst->config.num_entries;
    hipLaunchKernelGGL(merge_dense_kernel, dim3(grid_for(total, kABlock * 4)), dim3(kABlock), 0, s, image,
                       dst->image, dst->exist_words, static_cast<long long>(dst->config.num_entries),
                       dst->num_cols, dst->col_kinds);
  } else {
    // the source table may have grown differently: its capacity follows from the image size
    const size_t row_bytes = sizeof(unsigned long long) * (dst->num_cols + 1);
    if (image_bytes == 0 || image_bytes % row_bytes != 0) return QSX_ERR_INVALID_ARGUMENT;
    const unsigned long long src_cap = image_bytes / row_bytes - 1;
    if (src_cap == 0 || (src_cap & (src_cap - 1)) != 0) return QSX_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(merge_hash_kernel, dim3(grid_for(src_cap + 1, kABlock)), dim3(kABlock), 0, s, image,
                       src_cap, dst->num_cols, dst->col_kinds, dst->hash_view());
  }
  QSX_CHECK_LAUNCH();
  return publish_control(dst, s);
}

int qsx_agg_merge(qsx_agg_state_t *dst, qsx_agg_state_t *src, qsx_stream_t stream) {
  if (dst == nullptr || src == nullptr || dst == src) return QSX_ERR_INVALID_ARGUMENT;
  if (dst->num_cols != src->num_cols || dst->dense != src->dense || (dst->dense && dst->image_bytes != src->image_bytes)) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  int rc = settle(src, as_stream(stream));   // the source's spilled rows are part of what is merged
  if (rc != QSX_OK) return rc;
  std::shared_lock<std::shared_mutex> lock(src->table_mutex);
  return qsx_agg_state_import_merge(dst, src->image, src->image_bytes, stream);
}

// Key range [begin, end) of partition `part` of `parts` of a dense (CollisionFreeVector) state with `entries` keys
// (storage/CollisionFreeVectorTable.hpp:192-208: contiguous ranges of ceil(entries / parts) keys, the last ones short or
// empty), the LSB-first existence words [first_word, last_word) that cover it, and the masks that cut the range out of its
// first and last word.  The ONE statement of this split: qsx_agg_finalize, qsx_agg_reduce_scatter and — through
// qsx_agg_dense_partition_range — quickstep_amd/distributed.py all take it from here.
struct DenseRange {
  long long begin, end, first_word, last_word;
  unsigned long long first_mask, last_mask;
};
static DenseRange dense_partition_range(long long entries, int parts, int part) {
  DenseRange r{};
  const long long length = (entries + parts - 1) / parts;
  r.begin = static_cast<long long>(part) * length < entries ? static_cast<long long>(part) * length : entries;
  r.end = r.begin + length < entries ? r.begin + length : entries;
  if (r.end > r.begin) {
    r.first_word = r.begin / 64;
    r.last_word = (r.end + 63) / 64;
  }
  r.first_mask = ~0ull << (r.begin & 63);
  r.last_mask = (r.end & 63) != 0 ? ~0ull >> (64 - (r.end & 63)) : ~0ull;
  return r;
}

// ---- partial aggregates across GPUs ------------------------------------------------------------------------------------------
// Dense (CollisionFreeVector) states: reduce-scatter.  Rank r ends up holding — and finalizes with partition = r,
// num_partitions = world — the merged groups of key range r (CollisionFreeVectorTable.hpp:192-208's contiguous ranges), and
// nothing else; per rank this moves 1 / world of what an all-reduce moves.  Every state column is reduced the way its
// accumulator combines (f64 +, int64 +, MIN, MAX); the existence bits of the owned range are the OR of what every rank holds
// for it (RCCL has no bitwise reduction: each rank sends every peer the words covering that peer's range).
int qsx_agg_dense_partition_range(int64_t num_entries, int num_partitions, int partition, int64_t *out_begin, int64_t *out_end,
                                   int64_t *out_first_word, int64_t *out_last_word, uint64_t *out_first_mask, uint64_t *out_last_mask) {
  if (num_entries < 0 || num_partitions < 1 || partition < 0 || partition >= num_partitions) return QSX_ERR_INVALID_ARGUMENT;
  const DenseRange r = dense_partition_range(num_entries, num_partitions, partition);
  if (out_begin != nullptr) *out_begin = r.begin;
  if (out_end != nullptr) *out_end = r.end;
  if (out_first_word != nullptr) *out_first_word = r.first_word;
  if (out_last_word != nullptr) *out_last_word = r.last_word;
  if (out_first_mask != nullptr) *out_first_mask = r.first_mask;
  if (out_last_mask != nullptr) *out_last_mask = r.last_mask;
  return QSX_OK;
}

int qsx_agg_reduce_scatter(qsx_comm_t *comm, qsx_agg_state_t *st, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (comm == nullptr || st == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  if (!st->dense) return QSX_ERR_UNSUPPORTED;   // hash states: qsx_agg_allgather_merge
  const int world = comm->world, rank = comm->rank;
  if (world == 1) return QSX_OK;
  const RcclApi *api = rccl();
  if (api == nullptr) return QSX_ERR_COMM;
  hipStream_t s = as_stream(stream);
  const long long entries = st->config.num_entries, exist_words = st->exist_words;
  const long long length = (entries + world - 1) / world, padded = length * world;
  const DenseRange own = dense_partition_range(entries, world, rank);
  const long long begin = own.begin, end = own.end, my_first = own.first_word, my_last = own.last_word;
  const long long my_words = my_last - my_first;
  CallScratch scratch(s);
  const size_t bytes_send = static_cast<size_t>(padded) * 8, bytes_mine = static_cast<size_t>(length) * 8,
               bytes_parts = static_cast<size_t>(my_words > 0 ? my_words : 1) * 8 * world;
  // (a rank that cannot get its scratch must not leave the others inside the first reduce-scatter: agree on it first)
  int rc = comm_agree(comm, scratch.reserve(CallScratch::padded(bytes_send) + CallScratch::padded(bytes_mine) + CallScratch::padded(bytes_parts) +
                                            CallScratch::padded(st->image_bytes)), s);
  if (rc != QSX_OK) return rc;
  unsigned long long *send = static_cast<unsigned long long *>(scratch.take(bytes_send));
  unsigned long long *mine = static_cast<unsigned long long *>(scratch.take(bytes_mine));
  unsigned long long *parts = static_cast<unsigned long long *>(scratch.take(bytes_parts));
  unsigned long long *reduced = static_cast<unsigned long long *>(scratch.take(st->image_bytes));
  std::shared_lock<std::shared_mutex> lock(st->table_mutex);
  // the reduced image: zero / identity everywhere, the owned key range filled in below
  QSX_HIP_TRY(hipMemsetAsync(reduced, 0, st->image_bytes, s));
  if (st->has_min_max) {
    hipLaunchKernelGGL(fill_identity_kernel, dim3(grid_for(entries, kABlock * 4)), dim3(kABlock), 0, s, reduced + exist_words, entries, entries,
                       st->num_cols, st->col_kinds);
    QSX_CHECK_LAUNCH();
  }
  for (int col = 0; col < st->num_cols; ++col) {
    const int kind = st->col_kinds.kind[col];
    const unsigned long long *column = st->image + exist_words + static_cast<long long>(col) * entries;
    QSX_HIP_TRY(hipMemcpyAsync(send, column, static_cast<size_t>(entries) * 8, hipMemcpyDeviceToDevice, s));
    if (padded > entries) {   // the tail of the last range: the accumulator's identity
      if (kind >= kAccMinI64) {
        ColKinds one{};
        one.kind[0] = kind;
        hipLaunchKernelGGL(fill_identity_kernel, dim3(1), dim3(kABlock), 0, s, send + entries, padded - entries, padded - entries, 1, one);
        QSX_CHECK_LAUNCH();
      } else {
        QSX_HIP_TRY(hipMemsetAsync(send + entries, 0, static_cast<size_t>(padded - entries) * 8, s));
      }
    }
    const ncclDataType_t type = kind == kAccSumF64 ? ncclFloat64 : ncclInt64;
    const ncclRedOp_t op = kind == kAccMinI64 ? ncclMin : (kind == kAccMaxI64 ? ncclMax : ncclSum);
    QSX_RCCL_TRY(api->ReduceScatter(send, mine, static_cast<size_t>(length), type, op, comm->comm, s), "ncclReduceScatter");
    if (end > begin) {
      QSX_HIP_TRY(hipMemcpyAsync(reduced + exist_words + static_cast<long long>(col) * entries + begin, mine, static_cast<size_t>(end - begin) * 8,
                                 hipMemcpyDeviceToDevice, s));
    }
  }
  // existence words of every peer's range -> that peer; the owner ORs them
  {
    RcclGroup group(api);
    for (int p = 0; p < world && group.ok(); ++p) {
      const DenseRange theirs = dense_partition_range(entries, world, p);
      const long long first = theirs.first_word, last = theirs.last_word;
      if (last > first) group.add(api->Send(st->image + first, static_cast<size_t>(last - first), ncclUint64, p, comm->comm, s), "ncclSend");
      if (my_words > 0) group.add(api->Recv(parts + static_cast<long long>(p) * my_words, static_cast<size_t>(my_words), ncclUint64, p, comm->comm, s), "ncclRecv");
    }
    rc = group.end();
    if (rc != QSX_OK) return rc;
  }
  if (my_words > 0) {
    // LSB-first existence words: keep bits [begin, end) only
    hipLaunchKernelGGL(or_words_kernel, dim3(grid_for(my_words, 256)), dim3(256), 0, s, parts, world, my_words, own.first_mask, own.last_mask,
                       reduced + my_first);
    QSX_CHECK_LAUNCH();
  }
  // the state becomes the reduced image
  QSX_HIP_TRY(hipMemcpyAsync(st->image, reduced, st->image_bytes, hipMemcpyDeviceToDevice, s));
  return QSX_OK;
}

// Hash-table states (Q1-sized: a few KiB): every rank gathers every image and merges the others' into its own — afterwards
// all ranks hold the whole merged table (the counterpart of merging the thread-private tables at finalize,
// storage/AggregationOperationState.cpp:925-948).  Images may differ in size (a table that grew): sizes first.
int qsx_agg_allgather_merge(qsx_comm_t *comm, qsx_agg_state_t *st, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (comm == nullptr || st == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  if (st->dense) return QSX_ERR_UNSUPPORTED;    // dense states: qsx_agg_reduce_scatter
  const int world = comm->world, rank = comm->rank;
  if (world == 1) return QSX_OK;
  const RcclApi *api = rccl();
  if (api == nullptr) return QSX_ERR_COMM;
  hipStream_t s = as_stream(stream);
  // Every step that can fail on one rank alone (the size query, the allocations, the export) happens BETWEEN collectives
  // and is followed by an agreement (comm_agree: one word per rank through the communicator's own buffer): either all
  // ranks go on to the next collective or all of them give the call up — none is left waiting for a peer that returned.
  size_t my_bytes = 0;
  int rc = comm_agree(comm, qsx_agg_state_export_bytes(st, &my_bytes, stream), s);
  if (rc != QSX_OK) return rc;
  // sizes: the agreement's buffer carries them (one word per rank, nothing allocated)
  std::vector<unsigned long long> sizes(static_cast<size_t>(world));
  comm->status_host[world] = static_cast<long long>(my_bytes);
  int status = QSX_OK;
  if (hipMemcpyAsync(comm->status_dev + world, comm->status_host + world, 8, hipMemcpyHostToDevice, s) != hipSuccess) status = QSX_ERR_HIP;
  const int gathered_sizes = rccl_status(api->AllGather(comm->status_dev + world, comm->status_dev, 1, ncclInt64, comm->comm, s), "ncclAllGather(sizes)");
  if (gathered_sizes != QSX_OK) return gathered_sizes;
  if (hipMemcpyAsync(comm->status_host, comm->status_dev, 8 * static_cast<size_t>(world), hipMemcpyDeviceToHost, s) != hipSuccess) status = QSX_ERR_HIP;
  const int waited = comm_wait(comm, s);
  if (waited != QSX_OK) return waited;
  size_t pad = 0;
  for (int r = 0; r < world; ++r) {
    sizes[static_cast<size_t>(r)] = static_cast<unsigned long long>(comm->status_host[r]);
    pad = sizes[static_cast<size_t>(r)] > pad ? static_cast<size_t>(sizes[static_cast<size_t>(r)]) : pad;
  }
  unsigned char *gathered = nullptr, *padded_image = nullptr;
  if (status == QSX_OK && device_malloc(&gathered, pad * world) != hipSuccess) status = QSX_ERR_OUT_OF_MEMORY;
  if (status == QSX_OK && device_malloc(&padded_image, pad) != hipSuccess) status = QSX_ERR_OUT_OF_MEMORY;
  if (status == QSX_OK) status = qsx_agg_state_export(st, padded_image, my_bytes, stream);
  status = comm_agree(comm, status, s);
  if (status == QSX_OK) status = rccl_status(api->AllGather(padded_image, gathered, pad, ncclUint8, comm->comm, s), "ncclAllGather");
  if (status == QSX_OK) status = comm_wait(comm, s);
  // (from here on a failure is this rank's alone and no collective follows it)
  for (int r = 0; r < world && status == QSX_OK; ++r) {
    if (r != rank) status = qsx_agg_state_import_merge(st, gathered + static_cast<size_t>(r) * pad, static_cast<size_t>(sizes[static_cast<size_t>(r)]), stream);
  }
  if (hipStreamSynchronize(s) != hipSuccess && status == QSX_OK) status = QSX_ERR_HIP;
  if (gathered != nullptr) (void)device_free(gathered);
  if (padded_image != nullptr) (void)device_free(padded_image);
  return status;
}

int qsx_agg_num_groups(qsx_agg_state_t *st, int64_t *out_groups, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (st == nullptr || out_groups == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  hipStream_t s = as_stream(stream);
  int rc = settle(st, s);
  if (rc != QSX_OK) return rc;
  std::shared_lock<std::shared_mutex> lock(st->table_mutex);
  if (st->config.strategy == QSX_AGG_SINGLE_STATE) {
    *out_groups = 1;
    return QSX_OK;
  }
  unsigned long long v = 0;
  if (st->dense) {
    // the counter belongs to this call: the partitions of a state are finalized by concurrent work orders, each asking
    // for the group count first — a counter in the state (zeroed, added to, read) gave two of them each other's halves
    CallScratch scratch(s);
    const int rc_scratch = scratch.reserve(CallScratch::padded(sizeof(unsigned long long)));
    if (rc_scratch != QSX_OK) return rc_scratch;
    unsigned long long *counter = static_cast<unsigned long long *>(scratch.take(sizeof(unsigned long long)));
    QSX_HIP_TRY(hipMemsetAsync(counter, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(popcount_words_kernel, dim3(grid_for(st->exist_words, kABlock * 32) < 2 * kCUs ? grid_for(st->exist_words, kABlock * 32) : 2 * kCUs), dim3(kABlock), 0, s,
                       st->image, st->exist_words, counter);
    QSX_CHECK_LAUNCH();
    QSX_HIP_TRY(hipMemcpyAsync(&v, counter, sizeof(v), hipMemcpyDeviceToHost, s));
    QSX_HIP_TRY(hipStreamSynchronize(s));
    *out_groups = static_cast<int64_t>(v);
  } else {
    QSX_HIP_TRY(hipMemcpyAsync(&v, st->control, sizeof(v), hipMemcpyDeviceToHost, s));
    QSX_HIP_TRY(hipStreamSynchronize(s));
    *out_groups = static_cast<int64_t>(v) + 1;  // + the sentinel slot
  }
  return QSX_OK;
}

int qsx_eval_expression(int num_columns, const void *const *cols, const int32_t *types, int num_instrs,
                        const qsx_expr_instr_t *instrs, const double *consts, qsx_operand_t result, int64_t n, double *out_dev,
                        qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (num_columns < 0 || num_columns > QSX_MAX_COLUMNS || num_instrs < 0 || num_instrs > QSX_MAX_INSTRS || n < 0 ||
      (num_columns > 0 && (cols == nullptr || types == nullptr)) || (num_instrs > 0 && instrs == nullptr) ||
      (n > 0 && out_dev == nullptr)) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  ExprProgram prog{};
  prog.num_instrs = num_instrs;
  for (int c = 0; c < num_columns; ++c) {
    if (types[c] != QSX_INT && types[c] != QSX_LONG && types[c] != QSX_FLOAT && types[c] != QSX_DOUBLE) return QSX_ERR_UNSUPPORTED;
    if (n > 0 && cols[c] == nullptr) return QSX_ERR_INVALID_ARGUMENT;
    prog.cols[c] = cols[c];
    prog.types[c] = types[c];
  }
  int defined = 0;
  auto valid = [&](const qsx_operand_t &o) {
    switch (o.kind) {
      case QSX_OPD_COLUMN: return o.index >= 0 && o.index < num_columns;
      case QSX_OPD_CONST: return o.index >= 0 && o.index < QSX_MAX_CONSTS && consts != nullptr;
      case QSX_OPD_TEMP: return o.index >= 0 && o.index < QSX_MAX_TEMPS && ((defined >> o.index) & 1) != 0;
      default: return false;
    }
  };
  for (int k = 0; k < num_instrs; ++k) {
    const qsx_expr_instr_t &in = instrs[k];
    if (in.op < QSX_EX_ADD || in.op > QSX_EX_DIV || in.dst < 0 || in.dst >= QSX_MAX_TEMPS || !valid(in.a) || !valid(in.b)) {
      return QSX_ERR_INVALID_ARGUMENT;
    }
    prog.instrs[k].op = in.op;
    prog.instrs[k].dst = in.dst;
    prog.instrs[k].a = DevOperand{in.a.kind, in.a.index};
    prog.instrs[k].b = DevOperand{in.b.kind, in.b.index};
    defined |= 1 << in.dst;
  }
  if (!valid(result)) return QSX_ERR_INVALID_ARGUMENT;
  prog.result = DevOperand{result.kind, result.index};
  for (int k = 0; k < QSX_MAX_CONSTS; ++k) prog.consts[k] = consts != nullptr ? consts[k] : 0.0;
  if (n == 0) return QSX_OK;
  hipStream_t s = as_stream(stream);
  // the program travels through a device slot, not the kernarg segment (DESIGN.md "Kernel arguments")
  ExprProgram *slot = device_slot<ExprProgram>(s);
  if (slot == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  hipLaunchKernelGGL(store_struct_kernel<ExprProgram>, dim3(1), dim3(64), 0, s, prog, slot);
  QSX_CHECK_LAUNCH();
  hipLaunchKernelGGL(eval_expression_kernel, dim3(grid_for(n, kABlock * kExprRows)), dim3(kABlock), 0, s, slot, n, out_dev);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_eval_expression_long(int num_columns, const void *const *cols, const int32_t *types, int num_instrs,
                             const qsx_expr_instr_t *instrs, const int64_t *consts, qsx_operand_t result, int64_t n, int out_width,
                             void *out_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (num_columns < 0 || num_columns > QSX_MAX_COLUMNS || num_instrs < 0 || num_instrs > QSX_MAX_INSTRS || n < 0 ||
      (num_columns > 0 && (cols == nullptr || types == nullptr)) || (num_instrs > 0 && instrs == nullptr) ||
      (n > 0 && out_dev == nullptr) || (out_width != 4 && out_width != 8)) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  IntExprProgram prog{};
  prog.num_instrs = num_instrs;
  for (int c = 0; c < num_columns; ++c) {
    if (types[c] != QSX_INT && types[c] != QSX_LONG) return QSX_ERR_UNSUPPORTED;   // FLOAT / DOUBLE operands: qsx_eval_expression
    if (n > 0 && cols[c] == nullptr) return QSX_ERR_INVALID_ARGUMENT;
    prog.cols[c] = cols[c];
    prog.types[c] = types[c];
  }
  int defined = 0;
  bool temp_is_int[QSX_MAX_TEMPS] = {};
  auto valid = [&](const qsx_operand_t &o) {
    switch (o.kind) {
      case QSX_OPD_COLUMN: return o.index >= 0 && o.index < num_columns;
      case QSX_OPD_CONST: return o.index >= 0 && o.index < QSX_MAX_CONSTS && consts != nullptr;
      case QSX_OPD_TEMP: return o.index >= 0 && o.index < QSX_MAX_TEMPS && ((defined >> o.index) & 1) != 0;
      default: return false;
    }
  };
  // INT unless a LONG is involved (a constant counts as INT when it fits 32 bits)
  auto is_int = [&](const qsx_operand_t &o) {
    switch (o.kind) {
      case QSX_OPD_COLUMN: return types[o.index] == QSX_INT;
      case QSX_OPD_CONST: return consts[o.index] >= INT32_MIN && consts[o.index] <= INT32_MAX;
      default: return temp_is_int[o.index];
    }
  };
  for (int k = 0; k < num_instrs; ++k) {
    const qsx_expr_instr_t &in = instrs[k];
    if (in.op < QSX_EX_ADD || in.op > QSX_EX_DIV || in.dst < 0 || in.dst >= QSX_MAX_TEMPS || !valid(in.a) || !valid(in.b)) {
      return QSX_ERR_INVALID_ARGUMENT;
    }
    prog.instrs[k].op = in.op;
    prog.instrs[k].dst = in.dst;
    prog.instrs[k].a = DevOperand{in.a.kind, in.a.index};
    prog.instrs[k].b = DevOperand{in.b.kind, in.b.index};
    prog.narrow[k] = is_int(in.a) && is_int(in.b) ? 1 : 0;
    temp_is_int[in.dst] = prog.narrow[k] != 0;
    defined |= 1 << in.dst;
  }
  if (!valid(result)) return QSX_ERR_INVALID_ARGUMENT;
  prog.result = DevOperand{result.kind, result.index};
  for (int k = 0; k < QSX_MAX_CONSTS; ++k) prog.consts[k] = consts != nullptr ? consts[k] : 0;
  if (n == 0) return QSX_OK;
  hipStream_t s = as_stream(stream);
  IntExprProgram *slot = device_slot<IntExprProgram>(s);
  if (slot == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  hipLaunchKernelGGL(store_struct_kernel<IntExprProgram>, dim3(1), dim3(64), 0, s, prog, slot);
  QSX_CHECK_LAUNCH();
  hipLaunchKernelGGL(eval_expression_long_kernel, dim3(grid_for(n, kABlock * 4)), dim3(kABlock), 0, s, slot, n, out_width, out_dev);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_agg_finalize(qsx_agg_state_t *st, int partition, int num_partitions, void *const *out_key_cols,
                     void *const *out_val_cols, uint8_t *const *out_null_cols, int64_t capacity,
                     int64_t *out_groups_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (st == nullptr || out_groups_dev == nullptr || num_partitions < 1 || partition < 0 ||
      partition >= num_partitions || capacity < 0 || out_val_cols == nullptr) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t s = as_stream(stream);
  int rc = settle(st, s);
  if (rc != QSX_OK) return rc;
  std::shared_lock<std::shared_mutex> lock(st->table_mutex);
  FinalizeDesc f = st->fin;
  for (int k = 0; k < f.num_keys; ++k) {
    if (out_key_cols == nullptr || out_key_cols[k] == nullptr) return QSX_ERR_INVALID_ARGUMENT;
    f.out_keys[k] = out_key_cols[k];
  }
  for (int a = 0; a < f.num_aggs; ++a) {
    f.out_vals[a] = out_val_cols[a];
    f.out_nulls[a] = out_null_cols != nullptr ? out_null_cols[a] : nullptr;
  }
  unsigned long long *out_groups = reinterpret_cast<unsigned long long *>(out_groups_dev);
  QSX_HIP_TRY(hipMemsetAsync(out_groups_dev, 0, sizeof(int64_t), s));
  f.collision = reinterpret_cast<int *>(st->control + 4);
  switch (st->config.strategy) {
    case QSX_AGG_SINGLE_STATE:
      if (partition != 0) return QSX_OK;
      if (capacity < 1) return QSX_ERR_CAPACITY;
      hipLaunchKernelGGL(finalize_single_kernel, dim3(1), dim3(64), 0, s, st->hash_view(), f, out_groups);
      break;
    case QSX_AGG_COMPACT_KEY:
      // compact-key tables are finalized in one piece (AggregationOperationState.cpp:925-948)
      if (partition != 0) return QSX_OK;
      hipLaunchKernelGGL(finalize_hash_kernel, dim3(grid_for(st->cap + 1, kABlock)), dim3(kABlock), 0, s,
                         st->hash_view(), f, 0, 1, 0, static_cast<long long>(capacity), out_groups);
      break;
    case QSX_AGG_GENERIC:
      hipLaunchKernelGGL(finalize_hash_kernel, dim3(grid_for(st->cap + 1, kABlock)), dim3(kABlock), 0, s,
                         st->hash_view(), f, partition, num_partitions, 1, static_cast<long long>(capacity),
                         out_groups);
      break;
    default: {
      const DenseRange range = dense_partition_range(st->config.num_entries, num_partitions, partition);
      const long long begin = range.begin, end = range.end;
      if (begin >= end) return QSX_OK;
      const long long first_word = range.first_word;
      const long long num_words = range.last_word - first_word;
      const long long num_tiles = (num_words + kDenseTileWords - 1) / kDenseTileWords;
      const DenseView d = st->dense_view();
      // the tile counts / offsets of THIS call: the partitions of a state are finalized by concurrent work orders
      // (FinalizeAggregationOperator makes one per partition), each on its own stream — scratch owned by the state was a race
      // between them (two partitions interleaving their scans: groups with their sums missing, seen once the host layer's
      // worker threads stopped being created per query and really ran the two work orders at the same time)
      CallScratch scratch(s);
      const size_t counts_bytes = sizeof(int32_t) * static_cast<size_t>(num_tiles + 1), offsets_bytes = sizeof(int64_t) * static_cast<size_t>(num_tiles + 2);
      const int rc_scratch = scratch.reserve(CallScratch::padded(counts_bytes) + CallScratch::padded(offsets_bytes));
      if (rc_scratch != QSX_OK) return rc_scratch;
      int32_t *tile_counts = static_cast<int32_t *>(scratch.take(counts_bytes));
      int64_t *tile_offsets = static_cast<int64_t *>(scratch.take(offsets_bytes));
      hipLaunchKernelGGL(dense_tile_count_kernel, dim3(grid_for(num_tiles, kABlock / kWave)), dim3(kABlock), 0, s,
                         d.exist, first_word, num_words, begin, end, num_tiles, tile_counts);
      QSX_CHECK_LAUNCH();
      hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, tile_counts,
                         static_cast<int64_t>(num_tiles), tile_offsets, out_groups_dev);
      QSX_CHECK_LAUNCH();
      hipLaunchKernelGGL(finalize_dense_kernel, dim3(grid_for(num_tiles, kABlock / kWave)), dim3(kABlock), 0, s,
                         d, f, first_word, num_words, begin, end, num_tiles, tile_offsets,
                         static_cast<long long>(capacity));
      break;
    }
  }
  QSX_CHECK_LAUNCH();
  if (f.wide_words != 0) {
    hipLaunchKernelGGL(report_collision_kernel, dim3(1), dim3(64), 0, s, f.collision, out_groups);
    QSX_CHECK_LAUNCH();
  }
  return QSX_OK;
}

}  // extern "C"
