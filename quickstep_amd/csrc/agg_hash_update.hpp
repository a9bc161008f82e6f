// agg_hash_update.hpp — the hash-strategy aggregation update kernel (K6/K8 with
// fused K11 expressions and the state's predicate): SINGLE_STATE, COMPACT_KEY
// and GENERIC all run through it.
//
// Reference loops replaced (paths in the Quickstep tree):
//   storage/ThreadPrivateCompactKeyHashTable.cpp:203-304 (+ .hpp:125-169),
//   storage/PackedPayloadHashTable.hpp:838-909,
//   storage/AggregationOperationState.cpp:476-519 (single state),
//   expressions/scalar/ScalarBinaryExpression.cpp:100-195 (temp vectors, fused here).
//
// Structure of one 256-thread workgroup (TR = 256 * V rows per tile):
//   1. STAGE   every referenced column stripe of the tile is copied HBM -> LDS by
//              the DMA path (global_load_lds_dwordx4: 1 KiB per wave instruction,
//              no VGPR round trip), double-buffered: the DMA of tile i+1 runs
//              while tile i is computed.  All HBM reads of the kernel are these
//              wide coalesced copies; the interpreter below only touches LDS, so
//              its dynamic column indexing costs address arithmetic, not scratch.
//   2. COMPUTE thread t owns rows t, t+256, ... of the tile.  Predicate, key
//              packing and the expression program are evaluated V rows at a time
//              per interpreted instruction (the program is wave-uniform: scalar
//              branches, amortised over 64 * V rows).
//   3. ACCUMULATE
//              * the first R = 4 distinct groups the workgroup meets are "register
//                groups": their codes sit in 4 LDS tag words and every thread keeps
//                private partial sums for them in VGPRs (predicated adds, no
//                atomics) — the TPC-H Q1 regime;
//              * further groups: workgroup-private open-addressing table in LDS
//                (ds_cmpst_b64 claim, ds_add_u64 / ds_add_f64);
//              * groups that do not fit LDS: global table, 64-bit global atomics.
//   4. FLUSH   registers -> LDS (wave reduction) -> global table: one global atomic
//              per group per accumulator per workgroup.
#ifndef QSX_CSRC_AGG_HASH_UPDATE_HPP_
#define QSX_CSRC_AGG_HASH_UPDATE_HPP_

#include "agg_common.hpp"

namespace qsx {

using lds_ptr_t = __attribute__((address_space(3))) void *;
using glb_ptr_t = const __attribute__((address_space(1))) void *;

// One wave instruction: lane l copies 16 bytes from its own global address to
// (wave-uniform LDS base) + 16 * l.
__device__ __forceinline__ void dma16(const char *global_lane_addr, char *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_ptr_t)global_lane_addr, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

// Copy `count` elements of `width` bytes (1/2/4/8) with ordinary loads; used
// for stripes that are not 16-byte aligned and for sub-16-byte tails.
__device__ __forceinline__ void copy_elements_to_lds(const char *src, char *dst, int count, int width) {
  for (int i = threadIdx.x; i < count; i += kABlock) {
    switch (width) {
      case 1: reinterpret_cast<uint8_t *>(dst)[i] = reinterpret_cast<const uint8_t *>(src)[i]; break;
      case 2: reinterpret_cast<uint16_t *>(dst)[i] = reinterpret_cast<const uint16_t *>(src)[i]; break;
      case 4: reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[i]; break;
      default: reinterpret_cast<uint64_t *>(dst)[i] = reinterpret_cast<const uint64_t *>(src)[i]; break;
    }
  }
}

// Issue the HBM -> LDS copy of one tile (rows [row0, row0 + rows)).
__device__ __forceinline__ void stage_tile(const DevConfig &c, const uint64_t *filter, char *tile, int64_t row0,
                                           int rows) {
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  for (int col = 0; col < c.num_columns; ++col) {
    const int off = c.lds_off[col];
    if (off < 0) continue;  // column not referenced by keys / predicate / expressions
    const int w = c.column_width[col];
    const char *src = static_cast<const char *>(c.cols[col]) + row0 * w;
    char *dst = tile + off;
    const int bytes = rows * w;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
      const int full = bytes & ~15;
      const int chunks = (full + 1023) >> 10;
      for (int k = wave; k < chunks; k += kABlock / kWave) {
        const int o = (k << 10) + (lane << 4);
        if (o < full) dma16(src + o, dst + (k << 10));
      }
      if (full != bytes) copy_elements_to_lds(src + full, dst + full, (bytes - full) / w, w);
    } else {
      copy_elements_to_lds(src, dst, rows, w);
    }
  }
  if (c.filter_lds_off >= 0) {
    copy_elements_to_lds(reinterpret_cast<const char *>(filter + (row0 >> 6)), tile + c.filter_lds_off,
                         (rows + 63) >> 6, 8);
  }
}

// ---- typed reads of a staged column ---------------------------------------------
__device__ __forceinline__ double tile_double(const DevConfig &c, const char *tile, int col, int r) {
  const char *p = tile + c.lds_off[col];
  switch (c.column_type[col]) {
    case QSX_INT: return static_cast<double>(reinterpret_cast<const int32_t *>(p)[r]);
    case QSX_LONG: return static_cast<double>(reinterpret_cast<const int64_t *>(p)[r]);
    case QSX_FLOAT: return static_cast<double>(reinterpret_cast<const float *>(p)[r]);
    default: return reinterpret_cast<const double *>(p)[r];
  }
}
__device__ __forceinline__ long long tile_int(const DevConfig &c, const char *tile, int col, int r) {
  const char *p = tile + c.lds_off[col];
  if (c.column_type[col] == QSX_INT) return reinterpret_cast<const int32_t *>(p)[r];
  return reinterpret_cast<const long long *>(p)[r];
}

template <int V>
struct Temps {
  double t[QSX_MAX_TEMPS][V];
};

template <int V>
__device__ __forceinline__ void temps_get(const Temps<V> &s, int i, double (&out)[V]) {
// The distinct asm comment per case keeps SimplifyCFG from sinking the V loads
// of all cases into one dynamically indexed load, which would force the temps
// out of VGPRs into scratch.
#define QSX_TG(k) \
  case k:         \
    _Pragma("unroll") for (int v = 0; v < V; ++v) { out[v] = s.t[k][v]; asm volatile("; temp get " #k : "+v"(out[v])); } \
    break;
  switch (i) {  // wave-uniform index: scalar branches, the temps stay in VGPRs
    QSX_TG(0) QSX_TG(1) QSX_TG(2) QSX_TG(3) QSX_TG(4) QSX_TG(5) QSX_TG(6)
    default:
#pragma unroll
      for (int v = 0; v < V; ++v) { out[v] = s.t[7][v]; asm volatile("; temp get 7" : "+v"(out[v])); }
      break;
  }
#undef QSX_TG
}
template <int V>
__device__ __forceinline__ void temps_set(Temps<V> &s, int i, const double (&in)[V]) {
#define QSX_TS(k) \
  case k:         \
    _Pragma("unroll") for (int v = 0; v < V; ++v) { s.t[k][v] = in[v]; asm volatile("; temp set " #k : "+v"(s.t[k][v])); } \
    break;
  switch (i) {
    QSX_TS(0) QSX_TS(1) QSX_TS(2) QSX_TS(3) QSX_TS(4) QSX_TS(5) QSX_TS(6)
    default:
#pragma unroll
      for (int v = 0; v < V; ++v) { s.t[7][v] = in[v]; asm volatile("; temp set 7" : "+v"(s.t[7][v])); }
      break;
  }
#undef QSX_TS
}

template <int V>
__device__ __forceinline__ void operand_vec(const DevConfig &c, const DevOperand &o, const Temps<V> &s,
                                            const char *tile, double (&out)[V]) {
  switch (o.kind) {
    case QSX_OPD_COLUMN:
#pragma unroll
      for (int v = 0; v < V; ++v) out[v] = tile_double(c, tile, o.index, threadIdx.x + v * kABlock);
      break;
    case QSX_OPD_CONST:
#pragma unroll
      for (int v = 0; v < V; ++v) out[v] = c.consts[o.index];
      break;
    default:
      temps_get<V>(s, o.index, out);
      break;
  }
}

template <int V>
__device__ __forceinline__ void predicate_vec(const DevConfig &c, const char *tile, bool (&live)[V]) {
  for (int p = 0; p < c.num_pred; ++p) {
    const DevPred term = c.pred[p];
    const char *base = tile + c.lds_off[term.column];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int r = threadIdx.x + v * kABlock;
      bool ok;
      switch (c.column_type[term.column]) {
        case QSX_INT:
          ok = compare_op<int32_t>(reinterpret_cast<const int32_t *>(base)[r], term.op, static_cast<int32_t>(term.literal));
          break;
        case QSX_LONG:
          ok = compare_op<int64_t>(reinterpret_cast<const int64_t *>(base)[r], term.op, static_cast<int64_t>(term.literal));
          break;
        case QSX_FLOAT:
          ok = compare_op<float>(reinterpret_cast<const float *>(base)[r], term.op,
                                 __uint_as_float(static_cast<uint32_t>(term.literal)));
          break;
        default:
          ok = compare_op<double>(reinterpret_cast<const double *>(base)[r], term.op,
                                  __longlong_as_double(static_cast<long long>(term.literal)));
          break;
      }
      live[v] = live[v] && ok;
    }
  }
}

// Compact key codes of V rows (ThreadPrivateCompactKeyHashTable.cpp:216-232).
template <int V>
__device__ __forceinline__ void key_codes_vec(const DevConfig &c, const char *tile, unsigned long long (&code)[V]) {
#pragma unroll
  for (int v = 0; v < V; ++v) code[v] = 0;
  for (int k = 0; k < c.num_keys; ++k) {
    const char *base = tile + c.lds_off[c.key_column[k]];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int r = threadIdx.x + v * kABlock;
      unsigned long long x;
      switch (c.key_width[k]) {
        case 1: x = reinterpret_cast<const uint8_t *>(base)[r]; break;
        case 2: x = reinterpret_cast<const uint16_t *>(base)[r]; break;
        case 4: x = reinterpret_cast<const uint32_t *>(base)[r]; break;
        default: x = reinterpret_cast<const unsigned long long *>(base)[r]; break;
      }
      code[v] |= x << c.key_shift[k];
    }
  }
}

// Where a row that is not in a register group accumulates.
enum : int { kDestNone = 0, kDestLds = 1, kDestGlobal = 2 };

// Decide where one live row accumulates: a register group (sel >= 0), the
// workgroup's LDS table or the global table; the row count of non-register
// groups is bumped right here.
__device__ __forceinline__ void classify_row(bool live, unsigned long long code,
                                             const unsigned long long (&tag)[kRegGroups],
                                             unsigned long long *l_tags, unsigned long long *l_keys,
                                             unsigned long long *l_state, int S, const HashTableView &g,
                                             int &sel, int &dest, long long &slot) {
  sel = -1;
  dest = kDestNone;
  slot = 0;
  if (!live) return;
  if (code != kEmptyCode) {
#pragma unroll
    for (int r = 0; r < kRegGroups; ++r) {
      if (tag[r] == code) sel = r;
    }
    if (sel < 0) {
      // Not in the snapshot: claim a free tag (first rows of a workgroup only).
#pragma unroll
      for (int r = 0; r < kRegGroups; ++r) {
        if (sel < 0) {
          unsigned long long k = __hip_atomic_load(&l_tags[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (k == kEmptyCode) k = atomicCAS(&l_tags[r], kEmptyCode, code);
          if (k == kEmptyCode || k == code) sel = r;
        }
      }
    }
  }
  if (sel >= 0) return;
  const int s = code == kEmptyCode ? -1 : lds_find_or_insert(l_keys, S, code);
  if (s >= 0) {
    dest = kDestLds;
    slot = s;
    atomicAdd(&l_state[s], 1ull);
  } else {
    const unsigned long long gs = global_find_or_insert(g, code);
    if (gs != ~0ull) {
      dest = kDestGlobal;
      slot = static_cast<long long>(gs);
      global_add(g, 0, gs, 1ull, 1);
    }
  }
}

// Dynamic LDS: tile[2][tile_bytes] | tags[R] | rstate[R][NS+1] | l_keys[S] | l_state[NS+1][S]
template <int NS, int V>
__global__ __launch_bounds__(kABlock) void agg_hash_update_kernel(DevConfig c, int64_t n,
                                                                 const uint64_t *__restrict__ filter,
                                                                 HashTableView g, int S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int TR = kABlock * V;
  char *tiles = reinterpret_cast<char *>(smem_raw);
  unsigned long long *l_tags = reinterpret_cast<unsigned long long *>(smem_raw + 2 * c.tile_bytes);
  unsigned long long *l_rstate = l_tags + kRegGroups;               // [R][NS + 1]
  unsigned long long *l_keys = l_rstate + kRegGroups * (NS + 1);    // [S]
  unsigned long long *l_state = l_keys + S;                         // [NS + 1][S]

  for (int i = threadIdx.x; i < kRegGroups; i += kABlock) l_tags[i] = kEmptyCode;
  for (int i = threadIdx.x; i < kRegGroups * (NS + 1); i += kABlock) l_rstate[i] = 0;
  for (int i = threadIdx.x; i < S; i += kABlock) l_keys[i] = kEmptyCode;
  for (int i = threadIdx.x; i < (NS + 1) * S; i += kABlock) l_state[i] = 0;

  unsigned long long racc[kRegGroups][NS > 0 ? NS : 1];
  unsigned int rcnt[kRegGroups];
#pragma unroll
  for (int r = 0; r < kRegGroups; ++r) {
    rcnt[r] = 0;
#pragma unroll
    for (int j = 0; j < NS; ++j) racc[r][j] = 0;
  }

  const int64_t num_tiles = (n + TR - 1) / TR;
  int buf = 0;
  if (static_cast<int64_t>(blockIdx.x) < num_tiles) {
    const int64_t row0 = static_cast<int64_t>(blockIdx.x) * TR;
    stage_tile(c, filter, tiles, row0, static_cast<int>(n - row0 < TR ? n - row0 : TR));
  }
  for (int64_t tile_id = blockIdx.x; tile_id < num_tiles; tile_id += gridDim.x) {
    // The tile staged during the previous iteration has landed; every wave is
    // done with the other buffer.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int64_t next = tile_id + gridDim.x;
    if (next < num_tiles) {
      const int64_t row0 = next * TR;
      stage_tile(c, filter, tiles + (buf ^ 1) * c.tile_bytes, row0, static_cast<int>(n - row0 < TR ? n - row0 : TR));
    }
    const char *tile = tiles + buf * c.tile_bytes;
    buf ^= 1;
    const int64_t row0 = tile_id * TR;
    const int rows = static_cast<int>(n - row0 < TR ? n - row0 : TR);

    // ---- which rows are live -------------------------------------------------
    bool live[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int r = threadIdx.x + v * kABlock;
      live[v] = r < rows;
      if (c.filter_lds_off >= 0 && live[v]) {
        const uint64_t word = reinterpret_cast<const uint64_t *>(tile + c.filter_lds_off)[r >> 6];
        live[v] = msb_bit(word, r & 63);
      }
    }
    predicate_vec<V>(c, tile, live);

    // ---- group of every row ----------------------------------------------------
    unsigned long long code[V];
    key_codes_vec<V>(c, tile, code);
    unsigned long long tag[kRegGroups];
#pragma unroll
    for (int r = 0; r < kRegGroups; ++r) {
      tag[r] = __hip_atomic_load(&l_tags[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    int sel[V];     // register group, or -1
    int dest[V];    // for sel < 0: kDestLds / kDestGlobal / kDestNone
    long long slot[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      classify_row(live[v], code[v], tag, l_tags, l_keys, l_state, S, g, sel[v], dest[v], slot[v]);
    }
#pragma unroll
    for (int r = 0; r < kRegGroups; ++r) {
#pragma unroll
      for (int v = 0; v < V; ++v) rcnt[r] += sel[v] == r ? 1u : 0u;
    }

    // ---- expression program ------------------------------------------------------
    Temps<V> temps;
    for (int k = 0; k < c.num_instrs; ++k) {
      const DevInstr in = c.instrs[k];
      double a[V], b[V], res[V];
      operand_vec<V>(c, in.a, temps, tile, a);
      operand_vec<V>(c, in.b, temps, tile, b);
      switch (in.op) {
        case QSX_EX_ADD:
#pragma unroll
          for (int v = 0; v < V; ++v) res[v] = a[v] + b[v];
          break;
        case QSX_EX_SUB:
#pragma unroll
          for (int v = 0; v < V; ++v) res[v] = a[v] - b[v];
          break;
        case QSX_EX_MUL:
#pragma unroll
          for (int v = 0; v < V; ++v) res[v] = a[v] * b[v];
          break;
        default:
#pragma unroll
          for (int v = 0; v < V; ++v) res[v] = a[v] / b[v];
          break;
      }
      temps_set<V>(temps, in.dst, res);
    }

    // ---- accumulate ------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const DevSum s = c.sums[j];
      unsigned long long inc[V];
      if (s.is_int) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
          inc[v] = static_cast<unsigned long long>(tile_int(c, tile, s.arg.index, threadIdx.x + v * kABlock));
        }
#pragma unroll
        for (int r = 0; r < kRegGroups; ++r) {
#pragma unroll
          for (int v = 0; v < V; ++v) racc[r][j] += sel[v] == r ? inc[v] : 0ull;
        }
      } else {
        double x[V];
        operand_vec<V>(c, s.arg, temps, tile, x);
#pragma unroll
        for (int r = 0; r < kRegGroups; ++r) {
          double acc = __longlong_as_double(static_cast<long long>(racc[r][j]));
#pragma unroll
          for (int v = 0; v < V; ++v) acc += sel[v] == r ? x[v] : 0.0;
          racc[r][j] = static_cast<unsigned long long>(__double_as_longlong(acc));
        }
#pragma unroll
        for (int v = 0; v < V; ++v) inc[v] = static_cast<unsigned long long>(__double_as_longlong(x[v]));
      }
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if (dest[v] == kDestLds) {
          lds_add(&l_state[(j + 1) * S + slot[v]], inc[v], s.is_int);
        } else if (dest[v] == kDestGlobal) {
          global_add(g, j + 1, static_cast<unsigned long long>(slot[v]), inc[v], s.is_int);
        }
      }
    }
  }

  // ---- registers -> LDS (wave reduction first: one LDS atomic per wave per word) ----
#pragma unroll
  for (int r = 0; r < kRegGroups; ++r) {
    const unsigned long long cnt = wave_reduce_add(static_cast<unsigned long long>(rcnt[r]));
    if (cnt == 0) continue;  // wave-uniform
    if (lane_id() == 0) atomicAdd(&l_rstate[r * (NS + 1)], cnt);
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      unsigned long long v;
      if (c.sums[j].is_int) {
        v = wave_reduce_add(racc[r][j]);
      } else {
        v = static_cast<unsigned long long>(__double_as_longlong(
            wave_reduce_add(__longlong_as_double(static_cast<long long>(racc[r][j])))));
      }
      if (lane_id() == 0) lds_add(&l_rstate[r * (NS + 1) + j + 1], v, c.sums[j].is_int);
    }
  }
  __syncthreads();

  // ---- LDS -> global table: one atomic per group per accumulator per workgroup ------
  for (int i = threadIdx.x; i < kRegGroups + S; i += kABlock) {
    unsigned long long code;
    const unsigned long long *src;
    int stride;
    if (i < kRegGroups) {
      code = l_tags[i];
      src = &l_rstate[i * (NS + 1)];
      stride = 1;
    } else {
      code = l_keys[i - kRegGroups];
      src = &l_state[i - kRegGroups];
      stride = S;
    }
    const unsigned long long cnt = src[0];
    if (code == kEmptyCode || cnt == 0) continue;
    const unsigned long long gs = global_find_or_insert(g, code);
    if (gs == ~0ull) continue;
    global_add(g, 0, gs, cnt, 1);
    for (int j = 0; j < NS; ++j) global_add(g, j + 1, gs, src[(j + 1) * stride], c.sums[j].is_int);
  }
}

}  // namespace qsx

#endif  // QSX_CSRC_AGG_HASH_UPDATE_HPP_
