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
//              * groups live in a workgroup-private open-addressing table in LDS
//                (ds_cmpst_b64 claim); every accumulator of a group is replicated
//                REP times (REP = 64 for few groups: one bank column per lane), so
//                a row costs one conflict-free ds_add_u64 / ds_add_f64 per
//                accumulator and no VGPR state — the TPC-H Q1 regime (4 groups)
//                and the thousands-of-groups regime run the same code;
//              * groups that do not fit LDS: global table, 64-bit global atomics.
//   4. FLUSH   fold the REP partials -> global table: one global atomic per group
//              per accumulator per workgroup.
#ifndef QSX_CSRC_AGG_HASH_UPDATE_HPP_
#define QSX_CSRC_AGG_HASH_UPDATE_HPP_

#include "agg_common.hpp"
#include "agg_translate.hpp"

namespace qsx {

using lds_ptr_t = __attribute__((address_space(3))) void *;
using glb_ptr_t = const __attribute__((address_space(1))) void *;

// Cache policy of the tile copies (the aux operand: 0 plain, 1 sc0, 2 nt, 16 sc1).  Non-temporal: the tiles are read once,
// and a kernel that only reads gets more out of HBM that way — tools/ubench/read_ceiling.hip: six streams side by side
// 6.8–7.0 TB/s non-temporal against 6.2 plain; the Q1 update over 600 M rows 3.18 against 3.29–3.44 ms on the same box.
#ifndef QSX_DMA_AUX
#define QSX_DMA_AUX 2
#endif
// One wave instruction: lane l copies 16 bytes from its own global address to
// (wave-uniform LDS base) + 16 * l.
// QSX_DMA_ASM=1 (experiment, through QSX_JIT_OPTIONS): the copy as an asm statement instead of the builtin.  hipcc counts the
// builtin's DMA like a store to LDS it cannot tell from any other and drains it (`s_waitcnt vmcnt(0)`) in front of the next
// LDS read and of every barrier, so a tile staged AHEAD of the work on the current one runs under nothing; an asm statement
// is absent from that bookkeeping (the tile loop has its own `s_waitcnt vmcnt(0)` + barrier before a tile is read).
#ifndef QSX_DMA_ASM
#define QSX_DMA_ASM 0
#endif
__device__ __forceinline__ void dma16(const char *global_lane_addr, char *lds_wave_base) {
#if QSX_DMA_ASM
  const unsigned lds_at = __builtin_amdgcn_readfirstlane(
      static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char *)lds_wave_base)));
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(global_lane_addr), "s"(lds_at) : "memory");
#else
  __builtin_amdgcn_global_load_lds((glb_ptr_t)global_lane_addr, (lds_ptr_t)lds_wave_base, 16, 0, QSX_DMA_AUX);
#endif
}

// Loop over a configuration-sized range.  kStatic = the configuration is a compile-time
// constant (AOT plan shape, agg_shapes.hpp): unroll to the ABI maximum so that every
// switch on a configuration field folds away; otherwise a plain run-time loop (interpreter).
template <bool kStatic, int MAX, typename F>
__device__ __forceinline__ void cfg_for(int count, F &&f) {
  if constexpr (kStatic) {
#pragma unroll
    for (int k = 0; k < MAX; ++k) {
      if (k < count) f(k);
    }
  } else {
    for (int k = 0; k < count; ++k) f(k);
  }
}

// Copy `count` elements of `width` bytes (1/2/4/8) with ordinary loads; used
// for stripes that are not 16-byte aligned and for sub-16-byte tails.
template <int BLOCK = kABlock>
__device__ __forceinline__ void copy_elements_to_lds(const char *src_generic, char *dst, int count, int width) {
  const char *src = as_global(src_generic);
  for (int i = threadIdx.x; i < count; i += BLOCK) {
    switch (width) {
      case 1: reinterpret_cast<uint8_t *>(dst)[i] = reinterpret_cast<const uint8_t *>(src)[i]; break;
      case 2: reinterpret_cast<uint16_t *>(dst)[i] = reinterpret_cast<const uint16_t *>(src)[i]; break;
      case 4: reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[i]; break;
      default: reinterpret_cast<uint64_t *>(dst)[i] = reinterpret_cast<const uint64_t *>(src)[i]; break;
    }
  }
}

// Issue the HBM -> LDS copy of one tile (rows [row0, row0 + rows)).
// A block's column base pointers held in registers (run of blocks): loaded together, one tile ahead of their use.  Read one
// by one from the table at stage time they are six dependent vector loads in front of the tile's DMA (Q1: 5.0 instead of
// 3.4 ms per 600 M rows).
struct ColumnBases {
  const void *p[QSX_MAX_COLUMNS];
};
template <bool kStatic, int BLOCK = kABlock, bool kRuns = false>
__device__ __forceinline__ void stage_tile(const DevConfig &c, const void *const *cols, const uint64_t *filter, char *tile,
                                           int64_t row0, int rows, const unsigned long long *const *nulls = nullptr,
                                           const ColumnBases *bases = nullptr) {
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  auto stage_column = [&](int col, const void *base) __attribute__((always_inline)) {
    const int off = c.lds_off[col];
    if (off < 0 && off != kRegDecoded) return;  // column not referenced by keys / predicate / expressions
    // a compressed attribute is staged as its code stripe (decode_tile_codes fills the value slots afterwards, or
    // decode_rows the thread's registers: lds_off = kRegDecoded, no slots)
    const bool coded = c.code_width[col] != 0;
    const int w = coded ? c.code_width[col] : c.column_width[col];
    const char *src = static_cast<const char *>(base) + row0 * w;
    char *dst = tile + (coded ? c.code_off[col] : off);
    const int bytes = rows * w;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
      const int full = bytes & ~15;
      const int chunks = (full + 1023) >> 10;
      for (int k = wave; k < chunks; k += BLOCK / kWave) {
        const int o = (k << 10) + (lane << 4);
        if (o < full) dma16(src + o, dst + (k << 10));
      }
      if (full != bytes) copy_elements_to_lds<BLOCK>(src + full, dst + full, (bytes - full) / w, w);
    } else {
      copy_elements_to_lds<BLOCK>(src, dst, rows, w);
    }
  };
  if constexpr (kRuns && kStatic) {
    // (plan shapes: the referenced columns are known, their bases sit in registers.  The interpreter reads the bases from
    // the table column by column: unrolling its loop over all 16 possible columns costs 200 VGPRs and scratch)
#pragma unroll
    for (int col = 0; col < QSX_MAX_COLUMNS; ++col) {
      if (col < c.num_columns) stage_column(col, bases->p[col]);
    }
  } else {
    cfg_for<kStatic, QSX_MAX_COLUMNS>(c.num_columns, [&](int col) __attribute__((always_inline)) { stage_column(col, cols[col]); });
  }
  if (c.filter_lds_off >= 0) {
    if (!kRuns || filter != nullptr) {
      copy_elements_to_lds<BLOCK>(reinterpret_cast<const char *>(filter + (row0 >> 6)), tile + c.filter_lds_off,
                           (rows + 63) >> 6, 8);
    } else {   // (a block of a run without a filter of its own: every row)
      for (int i = threadIdx.x; i < ((rows + 63) >> 6); i += BLOCK) reinterpret_cast<uint64_t *>(tile + c.filter_lds_off)[i] = ~0ull;
    }
  }
  // null words of the nullable columns the plan reads (zeros for a block without NULLs in that attribute)
  cfg_for<kStatic, QSX_MAX_COLUMNS>(c.num_null_cols, [&](int s) __attribute__((always_inline)) {
    const unsigned long long *src = nulls != nullptr ? nulls[s] : nullptr;
    char *dst = tile + c.null_lds_off[s];
    const int words = (rows + 63) >> 6;
    if (src != nullptr) {
      copy_elements_to_lds<BLOCK>(reinterpret_cast<const char *>(src + (row0 >> 6)), dst, words, 8);
    } else {
      for (int i = threadIdx.x; i < words; i += BLOCK) reinterpret_cast<uint64_t *>(dst)[i] = 0;
    }
  });
}

// Compressed attributes: every thread turns the staged codes of its V rows into values (dictionary entry, or the code
// itself for truncation) in the column's value slots.  Only the thread itself reads those slots afterwards (rows are
// owned per thread), so no barrier follows; with a static configuration the dictionary reads of all coded columns are
// in flight together.
template <bool kStatic, int V, int BLOCK = kABlock>
__device__ __forceinline__ void decode_tile_codes(const DevConfig &c, const void *const *dicts, char *tile, int trow, int rows) {
  cfg_for<kStatic, QSX_MAX_COLUMNS>(c.num_columns, [&](int col) __attribute__((always_inline)) {
    if (c.code_width[col] == 0 || c.lds_off[col] < 0) return;
    const char *codes = tile + c.code_off[col];
    char *slots = tile + c.lds_off[col];
    const void *dict = dicts != nullptr ? as_global(dicts[col]) : nullptr;   // (from a table of a run: tell the compiler it is device memory)
    uint32_t code[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int r = trow + v * BLOCK;
      switch (c.code_width[col]) {
        case 1: code[v] = reinterpret_cast<const uint8_t *>(codes)[r]; break;
        case 2: code[v] = reinterpret_cast<const uint16_t *>(codes)[r]; break;
        default: code[v] = reinterpret_cast<const uint32_t *>(codes)[r]; break;
      }
      if (r >= rows) code[v] = 0;   // stale LDS behind a partial tile must not index the dictionary
    }
    switch (c.column_type[col]) {
      case QSX_INT:
#pragma unroll
        for (int v = 0; v < V; ++v) reinterpret_cast<int32_t *>(slots)[trow + v * BLOCK] = dict != nullptr ? static_cast<const int32_t *>(dict)[code[v]] : static_cast<int32_t>(code[v]);
        break;
      case QSX_FLOAT:
#pragma unroll
        for (int v = 0; v < V; ++v) reinterpret_cast<float *>(slots)[trow + v * BLOCK] = dict != nullptr ? static_cast<const float *>(dict)[code[v]] : static_cast<float>(code[v]);
        break;
      case QSX_LONG:
#pragma unroll
        for (int v = 0; v < V; ++v) reinterpret_cast<long long *>(slots)[trow + v * BLOCK] = dict != nullptr ? static_cast<const long long *>(dict)[code[v]] : static_cast<long long>(code[v]);
        break;
      default:
#pragma unroll
        for (int v = 0; v < V; ++v) reinterpret_cast<double *>(slots)[trow + v * BLOCK] = dict != nullptr ? static_cast<const double *>(dict)[code[v]] : static_cast<double>(code[v]);
        break;
    }
  });
}

// Compressed attributes of a plan shape whose tile holds only the codes (plan_tile reg_decode, lds_off = kRegDecoded): the
// values of the thread's V rows live in registers — the column's own type, zero-extended into a 64-bit container.  With a
// static configuration every index below is a constant after unrolling, and only the columns the plan reads survive.
// Dictionaries of up to kDictLdsEntries entries whose size the caller named (qsx_agg_update_coded_sized) are copied into LDS
// once per workgroup (plan shapes of the hash path, one launch over one set of stripes) and read from there: the three
// 8-byte gathers per row of Q1 over lineitem's codes cost 0.3 ms per 600 M rows through the vector memory path.
constexpr int kDictLdsEntries = 64;
// (the sizes sit right behind the QSX_MAX_COLUMNS dictionary pointers of the call's table: aggregate.hip DictTable)
__device__ __forceinline__ int dict_entries_behind(const void *const *dicts, int col) {
  return reinterpret_cast<const int *>(as_global(dicts) + QSX_MAX_COLUMNS)[col];
}
template <int V>
struct DecodedRows {
  unsigned long long raw[QSX_MAX_COLUMNS][V];
};
// l_dict / dict_in_lds: the LDS copies (kDictLdsEntries 8-byte entries per decoded column, in column order) and which
// decoded columns have one (bit k = the k-th decoded column); nullptr / 0: every dictionary is read from memory.
template <int V, int BLOCK = kABlock>
__device__ __forceinline__ void decode_rows(const DevConfig &c, const void *const *dicts, const char *tile, int trow, int rows,
                                            DecodedRows<V> &dec, const unsigned long long *l_dict = nullptr, unsigned dict_in_lds = 0) {
  int k = -1;   // (a constant per column after unrolling)
#pragma unroll
  for (int col = 0; col < QSX_MAX_COLUMNS; ++col) {
    if (col >= c.num_columns || c.lds_off[col] != kRegDecoded) continue;
    ++k;
    const char *codes = tile + c.code_off[col];
    const void *dict = dicts != nullptr ? as_global(dicts[col]) : nullptr;
    uint32_t code[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int r = trow + v * BLOCK;
      switch (c.code_width[col]) {
        case 1: code[v] = reinterpret_cast<const uint8_t *>(codes)[r]; break;
        case 2: code[v] = reinterpret_cast<const uint16_t *>(codes)[r]; break;
        default: code[v] = reinterpret_cast<const uint32_t *>(codes)[r]; break;
      }
      if (r >= rows) code[v] = 0;   // stale LDS behind a partial tile must not index the dictionary
    }
    const bool narrow = c.column_type[col] == QSX_INT || c.column_type[col] == QSX_FLOAT;
    if (((dict_in_lds >> k) & 1u) != 0) {   // (wave-uniform)
#pragma unroll
      for (int v = 0; v < V; ++v) dec.raw[col][v] = l_dict[k * kDictLdsEntries + (code[v] < kDictLdsEntries ? code[v] : 0u)];
    } else if (dict != nullptr) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        dec.raw[col][v] = narrow ? static_cast<unsigned long long>(static_cast<const uint32_t *>(dict)[code[v]])
                                 : static_cast<const unsigned long long *>(dict)[code[v]];
      }
    } else {   // truncation: value = code (the conversions of decode_tile_codes)
#pragma unroll
      for (int v = 0; v < V; ++v) {
        switch (c.column_type[col]) {
          case QSX_INT: dec.raw[col][v] = code[v]; break;
          case QSX_FLOAT: dec.raw[col][v] = __float_as_uint(static_cast<float>(code[v])); break;
          case QSX_LONG: dec.raw[col][v] = code[v]; break;
          default: dec.raw[col][v] = static_cast<unsigned long long>(__double_as_longlong(static_cast<double>(code[v]))); break;
        }
      }
    }
  }
}

// ---- typed reads of a staged column ---------------------------------------------
// (dec / v: the registers of a plan shape's compressed attributes and which of the thread's rows r is; nullptr elsewhere)
// (kDec, not a null test on dec: a comparison of the register struct's address with the null pointer of another address space
// does not fold, counts as a use of the address and keeps the whole struct in scratch — found in round 4 on a shape with a
// decoded INT column: 520 bytes of scratch per lane, and wrong sums from the hipRTC build of it)
template <int V = 1, bool kDec = false>
__device__ __forceinline__ double tile_double(const DevConfig &c, const char *tile, int col, int r, const DecodedRows<V> *dec = nullptr,
                                              int v = 0) {
  if (kDec && c.lds_off[col] == kRegDecoded) {
    const unsigned long long raw = dec->raw[col][v];
    switch (c.column_type[col]) {
      case QSX_INT: return static_cast<double>(static_cast<int32_t>(raw));
      case QSX_LONG: return static_cast<double>(static_cast<int64_t>(raw));
      case QSX_FLOAT: return static_cast<double>(__uint_as_float(static_cast<uint32_t>(raw)));
      default: return __longlong_as_double(static_cast<long long>(raw));
    }
  }
  const char *p = tile + c.lds_off[col];
  switch (c.column_type[col]) {
    case QSX_INT: return static_cast<double>(reinterpret_cast<const int32_t *>(p)[r]);
    case QSX_LONG: return static_cast<double>(reinterpret_cast<const int64_t *>(p)[r]);
    case QSX_FLOAT: return static_cast<double>(reinterpret_cast<const float *>(p)[r]);
    default: return reinterpret_cast<const double *>(p)[r];
  }
}
template <int V = 1, bool kDec = false>
__device__ __forceinline__ long long tile_int(const DevConfig &c, const char *tile, int col, int r, const DecodedRows<V> *dec = nullptr,
                                              int v = 0) {
  if (kDec && c.lds_off[col] == kRegDecoded) {
    const unsigned long long raw = dec->raw[col][v];
    return c.column_type[col] == QSX_INT ? static_cast<long long>(static_cast<int32_t>(raw)) : static_cast<long long>(raw);
  }
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
    _Pragma("unroll") for (int v = 0; v < V; ++v) { out[v] = s.t[k][v]; asm("; temp get " #k : "+v"(out[v])); } \
    break;
  switch (i) {  // wave-uniform index: scalar branches, the temps stay in VGPRs
    QSX_TG(0) QSX_TG(1) QSX_TG(2) QSX_TG(3) QSX_TG(4) QSX_TG(5) QSX_TG(6)
    default:
#pragma unroll
      for (int v = 0; v < V; ++v) { out[v] = s.t[7][v]; asm("; temp get 7" : "+v"(out[v])); }
      break;
  }
#undef QSX_TG
}
template <int V>
__device__ __forceinline__ void temps_set(Temps<V> &s, int i, const double (&in)[V]) {
#define QSX_TS(k) \
  case k:         \
    _Pragma("unroll") for (int v = 0; v < V; ++v) { s.t[k][v] = in[v]; asm("; temp set " #k : "+v"(s.t[k][v])); } \
    break;
  switch (i) {
    QSX_TS(0) QSX_TS(1) QSX_TS(2) QSX_TS(3) QSX_TS(4) QSX_TS(5) QSX_TS(6)
    default:
#pragma unroll
      for (int v = 0; v < V; ++v) { s.t[7][v] = in[v]; asm("; temp set 7" : "+v"(s.t[7][v])); }
      break;
  }
#undef QSX_TS
}

template <int V, int BLOCK = kABlock, bool kDec = false>
__device__ __forceinline__ void operand_vec(const DevConfig &c, const DevOperand &o, const Temps<V> &s,
                                            const char *tile, int trow, double (&out)[V], const DecodedRows<V> *dec = nullptr) {
  switch (o.kind) {
    case QSX_OPD_COLUMN:
#pragma unroll
      for (int v = 0; v < V; ++v) out[v] = tile_double<V, kDec>(c, tile, o.index, trow + v * BLOCK, dec, v);
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

// Interpreter operand: everything is resolved to (mode, LDS byte offset | immediate).
template <int V, int BLOCK = kABlock>
__device__ __forceinline__ void plan_operand_vec(const PlanOperand &o, const char *tile, const char *temps, int trow,
                                                 double (&out)[V]) {
  switch (o.mode) {
    case kPlanImm:
#pragma unroll
      for (int v = 0; v < V; ++v) out[v] = o.imm;
      break;
    case kPlanTileI32: {
      const int32_t *p = reinterpret_cast<const int32_t *>(tile + o.off);
#pragma unroll
      for (int v = 0; v < V; ++v) out[v] = static_cast<double>(p[trow + v * BLOCK]);
      break;
    }
    case kPlanTileI64: {
      const long long *p = reinterpret_cast<const long long *>(tile + o.off);
#pragma unroll
      for (int v = 0; v < V; ++v) out[v] = static_cast<double>(p[trow + v * BLOCK]);
      break;
    }
    case kPlanTileF32: {
      const float *p = reinterpret_cast<const float *>(tile + o.off);
#pragma unroll
      for (int v = 0; v < V; ++v) out[v] = static_cast<double>(p[trow + v * BLOCK]);
      break;
    }
    default: {  // f64 column of the tile or f64 temp slot: same load, different base
      const double *p = reinterpret_cast<const double *>((o.mode == kPlanTempF64 ? temps : tile) + o.off);
#pragma unroll
      for (int v = 0; v < V; ++v) out[v] = p[trow + v * BLOCK];
      break;
    }
  }
}

template <bool kStatic, int V, int BLOCK = kABlock>
__device__ __forceinline__ void predicate_vec(const DevConfig &c, const char *tile, int trow, bool (&live)[V],
                                              const DecodedRows<V> *dec = nullptr) {
  cfg_for<kStatic, QSX_MAX_PRED_TERMS>(c.num_pred, [&](int p) __attribute__((always_inline)) {
    const DevPred term = c.pred[p];
    if (kStatic && c.lds_off[term.column] == kRegDecoded) {
      // a compressed attribute of a plan shape: the value is in the thread's registers
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const unsigned long long raw = dec->raw[term.column][v];
        bool ok;
        switch (c.column_type[term.column]) {
          case QSX_INT: ok = compare_op<int32_t>(static_cast<int32_t>(raw), term.op, static_cast<int32_t>(term.literal)); break;
          case QSX_LONG: ok = compare_op<int64_t>(static_cast<int64_t>(raw), term.op, static_cast<int64_t>(term.literal)); break;
          case QSX_FLOAT:
            ok = compare_op<float>(__uint_as_float(static_cast<uint32_t>(raw)), term.op, __uint_as_float(static_cast<uint32_t>(term.literal)));
            break;
          case QSX_DATE: ok = compare_op<long long>(date_ordered(raw), term.op, date_ordered(term.literal)); break;
          default:
            ok = compare_op<double>(__longlong_as_double(static_cast<long long>(raw)), term.op,
                                    __longlong_as_double(static_cast<long long>(term.literal)));
            break;
        }
        live[v] = live[v] && ok;
      }
      return;
    }
    const char *base = tile + c.lds_off[term.column];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int r = trow + v * BLOCK;
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
        case QSX_DATE:   // year, month, day (types/DatetimeLit.hpp:65-90)
          ok = compare_op<long long>(date_ordered(reinterpret_cast<const unsigned long long *>(base)[r]), term.op,
                                     date_ordered(term.literal));
          break;
        default:
          ok = compare_op<double>(reinterpret_cast<const double *>(base)[r], term.op,
                                  __longlong_as_double(static_cast<long long>(term.literal)));
          break;
      }
      live[v] = live[v] && ok;
    }
  });
}

// Word w of the packed wide key of tile row r (DevConfig::wide_words).
template <int V = 1, bool kDec = false>
__device__ __forceinline__ unsigned long long key_word_of(const DevConfig &c, const char *tile, int w, int r,
                                                          const DecodedRows<V> *dec = nullptr, int v = 0) {
  unsigned long long word = 0;
#pragma unroll
  for (int k = 0; k < QSX_MAX_KEYS; ++k) {
    if (k < c.num_keys && c.key_word[k] == w) {
      const char *base = tile + c.lds_off[c.key_column[k]];
      unsigned long long x;
      if (kDec && c.lds_off[c.key_column[k]] == kRegDecoded) {
        x = dec->raw[c.key_column[k]][v];   // (zero-extended in its container: the same bits the stripe would hold)
      } else
      switch (c.key_width[k]) {
        case 1: x = reinterpret_cast<const uint8_t *>(base)[r]; break;
        case 2: x = reinterpret_cast<const uint16_t *>(base)[r]; break;
        case 4: x = reinterpret_cast<const uint32_t *>(base)[r]; break;
        default: x = reinterpret_cast<const unsigned long long *>(base)[r]; break;
      }
      if (c.column_type[c.key_column[k]] == QSX_DATE) x &= kDateValueMask;   // the padding bytes of a DateLit are not part of the key
      word |= x << c.key_shift[k];
    }
  }
  return word;
}

// Compact key codes of V rows (ThreadPrivateCompactKeyHashTable.cpp:216-232); for a wide key the 64-bit mixing hash of
// its words.
template <bool kStatic, int V, int BLOCK = kABlock>
__device__ __forceinline__ void key_codes_vec(const DevConfig &c, const char *tile, int trow, unsigned long long (&code)[V],
                                              const DecodedRows<V> *dec = nullptr) {
  if (c.wide_words != 0) {
#pragma unroll
    for (int v = 0; v < V; ++v) {
      unsigned long long words[kMaxKeyWords];
#pragma unroll
      for (int w = 0; w < kMaxKeyWords; ++w) words[w] = w < c.wide_words ? key_word_of<V, kStatic>(c, tile, w, trow + v * BLOCK, dec, v) : 0ull;
      code[v] = wide_key_code(words, c.wide_words, c.wide_hash_mask);
    }
    return;
  }
#pragma unroll
  for (int v = 0; v < V; ++v) code[v] = 0;
  cfg_for<kStatic, QSX_MAX_KEYS>(c.num_keys, [&](int k) __attribute__((always_inline)) {
    const char *base = tile + c.lds_off[c.key_column[k]];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int r = trow + v * BLOCK;
      unsigned long long x;
      if (kStatic && c.lds_off[c.key_column[k]] == kRegDecoded) {
        x = dec->raw[c.key_column[k]][v];
      } else
      switch (c.key_width[k]) {
        case 1: x = reinterpret_cast<const uint8_t *>(base)[r]; break;
        case 2: x = reinterpret_cast<const uint16_t *>(base)[r]; break;
        case 4: x = reinterpret_cast<const uint32_t *>(base)[r]; break;
        default: x = reinterpret_cast<const unsigned long long *>(base)[r]; break;
      }
      if (c.column_type[c.key_column[k]] == QSX_DATE) x &= kDateValueMask;
      code[v] |= x << c.key_shift[k];
    }
  });
}

// k-th group-by key of tile row r as a number: INT / LONG sign-extended, everything else (CHAR bytes, FLOAT / DOUBLE bit
// patterns) as the unsigned field.  Any injective image serves the key box of the group directory (DirView::bounds).
__device__ __forceinline__ long long key_field(const DevConfig &c, const char *tile, int k, int r) {
  const char *base = tile + c.lds_off[c.key_column[k]];
  switch (c.key_width[k]) {
    case 1: return reinterpret_cast<const uint8_t *>(base)[r];
    case 2: return reinterpret_cast<const uint16_t *>(base)[r];
    case 4:
      return c.column_type[c.key_column[k]] == QSX_INT ? static_cast<long long>(reinterpret_cast<const int32_t *>(base)[r])
                                                       : static_cast<long long>(reinterpret_cast<const uint32_t *>(base)[r]);
    default:
      return c.column_type[c.key_column[k]] == QSX_DATE
                 ? static_cast<long long>(reinterpret_cast<const unsigned long long *>(base)[r] & kDateValueMask)
                 : reinterpret_cast<const long long *>(base)[r];
  }
}

// Runs of equal ADJACENT keys inside a wave (clustered / sorted inputs, e.g. lineitem on l_orderkey).  One shuffle (the
// predecessor's key) and one ballot describe them all: a lane heads a run when its key differs from its predecessor's, the
// run a lane belongs to starts at the highest head at or below it (a count of leading zeros on the ballot), it ends where
// the next lane is a head, and the longest run says how many doubling steps the segmented sums need — two for the four
// lineitems of an order, not six.  (Round 4: the six-step max-scan for the run starts, a six-step scan for the row counts
// and six steps per aggregate were 32 ds_bpermute per wave row; the compute side of K7 took 1.17 ms per 200 M rows with
// nothing read from HBM and no atomic issued — tools/agg_dense_exp.sh.)  Equal keys that are not adjacent are never merged.
struct WaveRuns {
  int start;    // lane where the run containing this lane begins
  bool tail;    // this lane is the last of its run (it ends up with the run's sums)
  int steps;    // doubling steps that cover the longest run of the wave (wave-uniform)
};
__device__ __forceinline__ WaveRuns wave_runs(long long key) {
  const int lane = lane_id();
  const long long prev = __shfl_up(key, 1, kWave);
  const unsigned long long heads = __ballot(lane == 0 || prev != key);   // (bit 0 is always set)
  WaveRuns r;
  r.start = 63 - __builtin_clzll(heads & (~0ull >> (63 - lane)));
  r.tail = lane == kWave - 1 || ((heads >> (lane + 1)) & 1ull) != 0;
  const int back = lane - r.start;
  r.steps = 0;
  while (r.steps < 6 && __any(back >= (1 << r.steps))) ++r.steps;
  return r;
}
// The segmented sums add lane (i - off) only while it is still inside the run; the run's last lane ends up with the whole sum.
__device__ __forceinline__ double segmented_run_sum_f64(const WaveRuns &runs, double v) {
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    if (s >= runs.steps) break;   // (wave-uniform)
    const double v2 = __shfl_up(v, 1 << s, kWave);
    if (lane_id() - (1 << s) >= runs.start) v += v2;
  }
  return v;
}
// Same scan with the accumulator's own combine (MIN / MAX / integer SUM); dead lanes carry the identity.
__device__ __forceinline__ unsigned long long segmented_run_combine(const WaveRuns &runs, unsigned long long v, int kind) {
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    if (s >= runs.steps) break;
    const unsigned long long v2 = __shfl_up(v, 1 << s, kWave);
    if (lane_id() - (1 << s) >= runs.start) v = acc_combine(v, v2, kind);
  }
  return v;
}

// Group of one row.  Every row gets an LDS accumulator index so that the
// accumulate loop needs no branches: live rows whose group sits in the
// workgroup's LDS table get that group's slot, all other rows (filtered out, or
// routed to the global table because the LDS table is full / the code equals the
// empty marker) get the trash slot S.  Global-table rows additionally report
// their global slot (>= 0).  The group's row count is bumped right here.
__device__ __forceinline__ void classify_row(bool live, unsigned long long code, unsigned long long *l_keys,
                                             unsigned long long *l_acc, int S, int rep_shift, int lane_col,
                                             const HashTableView &g, int &slot, long long &global_slot) {
  slot = (S << rep_shift) + lane_id();  // trash: one column per lane whatever REP is (no same-address pile-up)
  global_slot = -1;
  if (!live) return;
  const int s = code == kEmptyCode ? -1 : lds_find_or_insert(l_keys, S, code);
  if (s >= 0) {
    slot = (s << rep_shift) + lane_col;  // index inside one accumulator plane
    atomicAdd(&l_acc[slot], 1ull);       // plane 0 = row count
  } else {
    const unsigned long long gs = global_find_or_insert(g, code);
    if (gs != ~0ull) {
      global_slot = static_cast<long long>(gs);
      global_add(g, 0, gs, 1ull, kAccSumI64);
    }
  }
}

// Dynamic LDS: tile[nbuf][tile_bytes] | temps[temps_bytes] | l_keys[S] | l_acc[NS + 1][S * REP + 64]   (last 64 = trash columns)
// temps = the interpreter's expression values, one f64 column of TR rows per live value (a thread only
// ever touches its own rows, so no barrier separates the interpreted instructions).
// l_acc holds, per accumulator plane and group slot, REP = 2^rep_shift partial
// values; a lane adds into column (lane & (REP - 1)), so with REP = 64 every
// lane owns its bank column and a wave's ds_add never conflicts (measured
// ~6 ns per wave instruction per CU, tools/ubench/lds_atomic.hip).
// kDense: COLLISION_FREE sink — the group of a row is its key value, accumulators are the dense
// arrays in HBM (DenseView), adjacent equal keys of a wave are combined before the atomics.
// kDense && kDir: a dense state small enough for ONE workgroup's LDS (S = num_entries accumulators per aggregate, the
// directory kernels' layout and geometry, no DirView): the group number is the key value, rows accumulate with LDS atomics
// and the workgroup adds its totals to the dense arrays at its end.  Random global atomics complete at ~24 G/s whatever
// their scope, target size or slicing by XCD (tools/ubench/atomic_scope.hip) and same-address ones one at a time: a
// 25-entry state took 290 ms per 100 M rows through the per-row atomics.
// kRuns: the rows are a run of blocks (agg_common.hpp BlockRunView in `pieces`; cols / filter unused).  A compile-time
// flavour, not a run-time test: with both sources in one kernel the by-value column table has its address taken and moves
// to scratch (136 bytes per lane, Q1 3.3 -> 4.9 ms per 600 M rows).
// REG > 0 (plan shapes over a table sized for a handful of groups; off by default, QSX_AGG_REG_GROUPS=1): every wave keeps
// the accumulators of the first REG groups it meets in its lanes' REGISTERS — the reference's
// ThreadPrivateCompactKeyHashTable taken literally — and a row of such a group costs exec-masked VALU adds instead of NS + 1
// LDS atomics plus the table lookup.  The LDS table stays what it was for every other row (a wave adopts a key only after a
// row of it found an LDS slot there, and remembers the slot); the lanes add their registers to that slot's planes when the
// workgroup is done with its tiles, in front of the flush.  Measured: the LDS atomics disappear
// (SQ_LDS_ADDR_CONFLICT 9e7 -> 4e5 cycles) but every (row, entry) pair executes its block of adds whenever any lane of the
// wave matches (4x the f64 adds) and the registers cost a workgroup per CU: slower than the atomics it replaces.
template <bool kStatic, bool kDense, int NS, int V, bool kDir = false, int BLOCK = kABlock, bool kDirBuild = false, bool kRuns = false,
          int REG = 0>
__device__ __forceinline__ void agg_hash_update_body(const DevConfig &c, const void *const *cols, const void *const *dicts, int64_t n,
                                                     const uint64_t *__restrict__ filter, const HashTableView &g,
                                                     const DenseView &dense, int S, int rep_shift, int nbuf,
                                                     int ranges, const long long *__restrict__ pieces = nullptr,
                                                     const unsigned long long *const *nulls = nullptr,
                                                     const DirView *dir = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int TR = BLOCK * V;
  char *tiles = reinterpret_cast<char *>(smem_raw);
  char *lds_temps = reinterpret_cast<char *>(smem_raw) + nbuf * c.tile_bytes;
  unsigned long long *l_keys = reinterpret_cast<unsigned long long *>(smem_raw + nbuf * c.tile_bytes + c.temps_bytes);  // [S]
  unsigned long long *l_acc = l_keys + S;                                                           // [NS + 1][S << rep_shift]
  const int plane = (S << rep_shift) + kWave;  // + 64 trash columns
  const int lane_col = lane_id() & ((1 << rep_shift) - 1);
  // kDir (group directory, agg_common.hpp): S = gids with an accumulator here, no replication, no key table — the
  // accumulator of gid g is word g of its plane; the row counts are 32-bit words behind the NS planes
  // (a workgroup sees < 2^32 rows of one group between two flushes: it flushes once, at its end, after < 2^32 rows in all
  // — the launcher bounds the rows per workgroup).
  // Dynamic LDS: tile[nbuf][tile_bytes] | temps | l_sum[NS][S + 64] u64 | l_cnt[S + 64] u32
  // A wide key's hidden MIN / MAX accumulators (the last 2 x wide_words of c.sums) have no planes here: the directory hands
  // out a gid only to rows whose key words all match (DirView::wide_words), the flush writes the words into those columns.
  const int ns_lds = kDir ? NS - 2 * c.wide_words : NS;
  unsigned long long *l_sum = l_keys;
  unsigned int *l_cnt = reinterpret_cast<unsigned int *>(l_keys + static_cast<size_t>(ns_lds) * plane);
  auto acc_plane_of = [&](int j) { return kDir ? l_sum + static_cast<size_t>(j) * plane : l_acc + static_cast<size_t>(j + 1) * plane; };
  // kDirBuild with a wide key: the words of the key that claimed set slot i (behind the S codes)
  unsigned long long *l_set_words = l_keys + S;

  if constexpr (kDirBuild) {
    // build pass of the group directory: only the set of key codes this workgroup sees, S slots
    if (dir->build_step == 1) {
      if (key_box_of(*dir, c.num_keys).usable) return;   // (uniform) groups will be numbered by their place in the key box
      for (int i = threadIdx.x; i < S; i += BLOCK) l_keys[i] = kEmptyCode;
    }
  } else if constexpr (kDir) {
    for (int i = threadIdx.x; i < plane; i += BLOCK) l_cnt[i] = 0;
  } else {
    for (int i = threadIdx.x; i < S; i += BLOCK) l_keys[i] = kEmptyCode;
    for (int i = threadIdx.x; i < plane; i += BLOCK) l_acc[i] = 0;  // row counts
  }
#pragma unroll
  for (int j = 0; j < (kDirBuild ? 0 : NS); ++j) {
    if (j >= ns_lds) break;
    const unsigned long long identity = static_cast<unsigned long long>(acc_identity(c.sums[j].kind));
    unsigned long long *p = acc_plane_of(j);
    for (int i = threadIdx.x; i < plane; i += BLOCK) p[i] = identity;
  }

  // nbuf == 2: the DMA of tile i+1 overlaps the compute of tile i inside the workgroup;
  // nbuf == 1: one buffer, overlap comes from the other workgroups resident on the CU.
  // ranges > 1 (more groups than one LDS table holds): the workgroups split into `ranges`
  // families; family r walks ALL tiles but only aggregates the groups whose hash falls into
  // range r, so each family's groups fit its LDS tables.  Costs `ranges` reads of the input
  // instead of (NS + 1) global atomics per row.
  // pieces != nullptr (partitioned aggregation, aggregate.hip): the input was hash-partitioned on the key code into
  // `ranges` pieces; family r walks only piece r = rows [pieces[r], pieces[r] + pieces[ranges + r]) — its groups are
  // its own by construction, no hash test, every row is read once.
  const int my_range = ranges > 1 ? static_cast<int>(blockIdx.x % ranges) : 0;
  // (build pass of the group directory: every sample_stride-th tile only)
  const int64_t first_tile = kDirBuild ? static_cast<int64_t>(blockIdx.x) * dir->sample_stride + dir->sample_phase
                                       : (ranges > 1 ? blockIdx.x / ranges : blockIdx.x);
  const int64_t tile_step = kDirBuild ? static_cast<int64_t>(gridDim.x) * dir->sample_stride
                                      : (ranges > 1 ? gridDim.x / ranges : gridDim.x);
  // kDirBuild: per-thread bounds of the keys seen (order-preserving unsigned images; DirView::bounds)
  unsigned long long seen_hi[QSX_MAX_KEYS], seen_nlo[QSX_MAX_KEYS];
#pragma unroll
  for (int k = 0; k < QSX_MAX_KEYS; ++k) seen_hi[k] = seen_nlo[k] = 0;
  KeyBox box;
  box.usable = false;
  // (the box is one of key FIELDS: a wide key's groups are numbered by position like a narrow key's)
  if constexpr (kDir && !kDense) box = key_box_of(*dir, c.num_keys);
  // a run of blocks (agg_common.hpp BlockRunView): the tiles of all blocks, each block with its own stripes
  constexpr bool batched = kRuns;
  BlockRunView run{};
  if constexpr (batched) run = block_run_view(pieces, TR);
  const bool by_piece = pieces != nullptr && !batched;
  const int64_t row_begin = by_piece ? pieces[my_range] : 0;
  const int64_t row_end = by_piece ? row_begin + pieces[ranges + my_range] : n;
  const int64_t num_tiles = batched ? run.first_tile[run.num_blocks] : (row_end - row_begin + TR - 1) / TR;
  auto tile_row0 = [&](int64_t t) { return row_begin + t * TR; };
  auto tile_rows = [&](int64_t t) { return static_cast<int>(row_end - tile_row0(t) < TR ? row_end - tile_row0(t) : TR); };
  // where tile t's rows are: stripes, filter, first row inside the stripes, row count
  struct TileSource {
    const void *const *cols;
    ColumnBases bases;       // kRuns only
    const uint64_t *filter;
    const void *const *dicts;   // kRuns only: the block's dictionaries
    int64_t row0;
    int rows;
  };
  auto locate = [&](int64_t t) {
    TileSource src;
    if constexpr (batched) {
      const long long b = block_of_tile(run, t);
      src.cols = run.cols + b * QSX_MAX_COLUMNS;
      if constexpr (kStatic) {
#pragma unroll
        for (int col = 0; col < QSX_MAX_COLUMNS; ++col) {
          src.bases.p[col] = (col < c.num_columns && (c.lds_off[col] >= 0 || c.lds_off[col] == kRegDecoded)) ? src.cols[col] : nullptr;
        }
      }
      src.filter = run.filters != nullptr ? reinterpret_cast<const uint64_t *>(run.filters[b]) : nullptr;
      src.dicts = run.dicts != nullptr ? run.dicts + b * QSX_MAX_COLUMNS : nullptr;
      src.row0 = (t - run.first_tile[b]) * TR;
      const long long left = run.rows[b] - src.row0;
      src.rows = static_cast<int>(left < TR ? left : TR);
    } else {
      src.cols = cols;
      src.filter = filter;
      src.dicts = dicts;
      src.row0 = tile_row0(t);
      src.rows = tile_rows(t);
    }
    return src;
  };
  int buf = 0;
  // (tables at their largest size only: a table sized for its estimate — Q1's 16 slots — never comes under pressure, and
  // with S a compile-time constant of the fixed-geometry shapes the whole mechanism folds away there)
  const bool can_flush = !kDense && !kDir && !kDirBuild && S >= 512;
  bool flush_enabled = true;
  // kHashCtlWords words behind the accumulator planes (dynamic LDS, counted by the launchers — a static __shared__ or a
  // __syncthreads_or would add static LDS the occupancy calculations do not see): [0..2] pressure flags, written during tile
  // i into slot i % 3, read by everyone at the barrier of tile i + 1, cleared by thread 0 one tile before their next use
  // (three slots: no reader and no writer of a slot can meet its clearing); [3], [4] statistics of the flush in progress
  unsigned long long *l_ctl = l_acc + static_cast<size_t>(NS + 1) * plane;
  unsigned long long *s_flush_stat = l_ctl + 3;
  if (can_flush && threadIdx.x < kHashCtlWords) l_ctl[threadIdx.x] = 0;
  // small dictionaries of known size -> LDS, behind the control words (the first barrier of the tile loop publishes them)
  unsigned long long *l_dict = l_ctl + kHashCtlWords;
  unsigned dict_in_lds = 0;
  if constexpr (kStatic && !kDense && !kDir && !kDirBuild && !kRuns) {
    if (dicts != nullptr) {
      int k = -1;
#pragma unroll
      for (int col = 0; col < QSX_MAX_COLUMNS; ++col) {
        if (col >= c.num_columns || c.lds_off[col] != kRegDecoded) continue;
        ++k;
        const void *dict = as_global(dicts[col]);
        const int entries = __builtin_amdgcn_readfirstlane(dict_entries_behind(dicts, col));
        if (dict == nullptr || entries <= 0 || entries > kDictLdsEntries) continue;
        dict_in_lds |= 1u << k;
        const bool narrow = c.column_type[col] == QSX_INT || c.column_type[col] == QSX_FLOAT;
        // (the slots behind `entries` are zeroed: decode_rows clamps a code to the LDS copy's size, not to `entries` — a code
        // beyond the dictionary, e.g. a reference block's NULL code = num_codes, then decodes to 0 whatever the LDS held)
        for (int i = threadIdx.x; i < kDictLdsEntries; i += BLOCK) {
          unsigned long long word = 0;
          if (i < entries) {
            word = narrow ? static_cast<unsigned long long>(static_cast<const uint32_t *>(dict)[i]) : static_cast<const unsigned long long *>(dict)[i];
          }
          l_dict[k * kDictLdsEntries + i] = word;
        }
      }
    }
  }
  int tile_count = 0;
  static_assert(REG == 0 || (kStatic && !kDense && !kDir && !kDirBuild), "register groups: plan shapes of the hash path only");
  // register groups (REG > 0): key code and LDS table slot per entry (wave-uniform), row count and accumulators per lane
  constexpr int kRegN = REG > 0 ? REG : 1;
  unsigned long long reg_key[kRegN];
  int reg_slot[kRegN];
  unsigned int reg_cnt[kRegN];
  unsigned long long reg_acc[kRegN][NS > 0 ? NS : 1];
  int reg_used = 0;
  if constexpr (REG > 0) {
#pragma unroll
    for (int e = 0; e < REG; ++e) {
      reg_key[e] = kEmptyCode;
      reg_slot[e] = 0;
      reg_cnt[e] = 0;
#pragma unroll
      for (int j = 0; j < NS; ++j) reg_acc[e][j] = static_cast<unsigned long long>(acc_identity(c.sums[j].kind));
    }
  }
  // LDS -> global table: fold the REP partials, one global atomic per group per accumulator per workgroup
  auto flush_table = [&](bool with_stats) {
    const int rep = 1 << rep_shift;
    unsigned long long occupied = 0, absorbed = 0;
    for (int sl = threadIdx.x; sl < S; sl += BLOCK) {
      const unsigned long long code = l_keys[sl];
      if (code == kEmptyCode) continue;
      unsigned long long cnt = 0;
      for (int r = 0; r < rep; ++r) cnt += l_acc[(sl << rep_shift) + r];
      if (cnt == 0) continue;
      ++occupied;
      absorbed += cnt;
      const unsigned long long gs = global_find_or_insert(g, code);
      if (gs == ~0ull) continue;
      global_add(g, 0, gs, cnt, kAccSumI64);
      for (int j = 0; j < NS; ++j) {
        const unsigned long long *p = l_acc + (j + 1) * plane + (sl << rep_shift);
        const int kind = c.sums[j].kind;
        unsigned long long v = static_cast<unsigned long long>(acc_identity(kind));
        for (int r = 0; r < rep; ++r) v = acc_combine(v, p[r], kind);
        global_add(g, j + 1, gs, v, kind);
      }
    }
    if (with_stats) {
      occupied = wave_reduce_add(occupied);
      absorbed = wave_reduce_add(absorbed);
      if (lane_id() == 0 && occupied != 0) {
        atomicAdd(&s_flush_stat[0], occupied);
        atomicAdd(&s_flush_stat[1], absorbed);
      }
    }
  };
  // (directory mode with one tile buffer stages the next tile as soon as the current one has been read, see the loop's end)
  const bool staged_ahead = nbuf == 2 || kDir;
  if (staged_ahead && first_tile < num_tiles) {
    const TileSource src = locate(first_tile);
    stage_tile<kStatic, BLOCK, kRuns>(c, kRuns ? src.cols : cols, kRuns ? src.filter : filter, tiles, src.row0, src.rows, nulls, &src.bases);
  }
  // (a run of blocks: the next tile's source is looked up while the current tile is computed — the table reads are a
  // dependent chain of scalar loads that would otherwise sit in front of every stage)
  TileSource carried{};
  if (first_tile < num_tiles) carried = locate(first_tile);
  for (int64_t tile_id = first_tile; tile_id < num_tiles; tile_id += tile_step, ++tile_count) {
    const TileSource here = carried;
    if (!staged_ahead) {
      __syncthreads();  // every wave is done reading the previous tile
#ifdef QSX_EXP_STAGE_ONCE   // (experiment: every tile works on the first one's bytes — the compute side without HBM traffic)
      if (tile_count == 0)
#endif
      stage_tile<kStatic, BLOCK, kRuns>(c, kRuns ? here.cols : cols, kRuns ? here.filter : filter, tiles, here.row0, here.rows, nulls, &here.bases);
    }
    // The tile has landed (nbuf == 2: it was staged during the previous iteration and
    // every wave is done with the other buffer).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (can_flush) {
      // Pressure on the LDS table (rows of the previous tile found no slot and paid global atomics): write the table out
      // and start over.  The table then works as a write-combining cache: with clustered keys (lineitem on l_orderkey) a
      // flushed group never comes back and every group costs one global update instead of one per row.  A flush that
      // absorbed fewer than two rows per group (keys in random order, far more groups than slots) switches the
      // mechanism off for this workgroup: it would only move the atomics around.
      __syncthreads();
      const bool pressure = tile_count > 0 && l_ctl[(tile_count - 1) % 3] != 0;
      if (threadIdx.x == 0) l_ctl[(tile_count + 1) % 3] = 0;
      if (pressure && flush_enabled) {
        flush_table(true);
        __syncthreads();
        const unsigned long long occupied = s_flush_stat[0], absorbed = s_flush_stat[1];
        flush_enabled = absorbed >= 2 * occupied;
        __syncthreads();
        if (threadIdx.x == 0) s_flush_stat[0] = s_flush_stat[1] = 0;
        for (int i = threadIdx.x; i < S; i += BLOCK) l_keys[i] = kEmptyCode;
        for (int i = threadIdx.x; i < plane; i += BLOCK) l_acc[i] = 0;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          const unsigned long long identity = static_cast<unsigned long long>(acc_identity(c.sums[j].kind));
          for (int i = threadIdx.x; i < plane; i += BLOCK) l_acc[(j + 1) * plane + i] = identity;
        }
        __syncthreads();
      }
    } else {
      __syncthreads();
    }
    const int64_t next = tile_id + tile_step;
    if (next < num_tiles) carried = locate(next);
    if (nbuf == 2 && next < num_tiles) {
#ifndef QSX_EXP_STAGE_ONCE
      stage_tile<kStatic, BLOCK, kRuns>(c, kRuns ? carried.cols : cols, kRuns ? carried.filter : filter, tiles + (buf ^ 1) * c.tile_bytes,
                                        carried.row0, carried.rows, nulls, &carried.bases);
#endif
    }
    char *tile = tiles + buf * c.tile_bytes;
    if (nbuf == 2) buf ^= 1;
    const int rows = here.rows;

    const int trow = threadIdx.x;  // this thread's first row of the tile
#ifdef QSX_EXP_NO_COMPUTE   // (experiment through QSX_JIT_OPTIONS, tools/agg_coded_exp.sh: the tile pipeline alone — staging, waits, barriers)
    if constexpr (kStatic && !kDir && !kDirBuild) continue;   // (hash path and the dense per-row path)
    if constexpr (kStatic && kDir) {                          // (directory kernels: their one buffer is staged behind the reads)
      if (nbuf == 1 && next < num_tiles) {
        __syncthreads();
        stage_tile<kStatic, BLOCK, kRuns>(c, kRuns ? carried.cols : cols, kRuns ? carried.filter : filter, tiles, carried.row0, carried.rows,
                                          nulls, &carried.bases);
      }
      continue;
    }
#endif
    decode_tile_codes<kStatic, V, BLOCK>(c, kRuns ? here.dicts : dicts, tile, trow, rows);
    // plan shapes whose tile holds only the codes of the compressed attributes: their values, into registers
    DecodedRows<V> decoded;
    const DecodedRows<V> *dec = nullptr;
    if constexpr (kStatic) {
      decode_rows<V, BLOCK>(c, kRuns ? here.dicts : dicts, tile, trow, rows, decoded, l_dict, dict_in_lds);
      dec = &decoded;
    }

    // ---- which rows are live -------------------------------------------------
    bool live[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int r = trow + v * BLOCK;
      live[v] = r < rows;
      if (c.filter_lds_off >= 0 && live[v]) {
        const uint64_t word = reinterpret_cast<const uint64_t *>(tile + c.filter_lds_off)[r >> 6];
        live[v] = msb_bit(word, r & 63);
      }
    }
    // NULLs: bit s of nullbits = the row is NULL in null slot s; a NULL group-by key or predicate operand drops the row
    // (PackedPayloadHashTable.hpp:861-867; a comparison with NULL is not true)
    unsigned nullbits[V];
#pragma unroll
    for (int v = 0; v < V; ++v) nullbits[v] = 0;
    if (c.num_null_cols != 0) {
      cfg_for<kStatic, QSX_MAX_COLUMNS>(c.num_null_cols, [&](int s) __attribute__((always_inline)) {
        const uint64_t *words = reinterpret_cast<const uint64_t *>(tile + c.null_lds_off[s]);
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const int r = trow + v * BLOCK;
          if (msb_bit(words[r >> 6], r & 63)) nullbits[v] |= 1u << s;
        }
      });
#pragma unroll
      for (int v = 0; v < V; ++v) live[v] = live[v] && (nullbits[v] & c.row_null_mask) == 0;
    }
    predicate_vec<kStatic, V, BLOCK>(c, tile, trow, live, dec);

    // ---- group of every row ----------------------------------------------------
    unsigned long long code[V];
    key_codes_vec<kStatic, V, BLOCK>(c, tile, trow, code, dec);
    if (ranges > 1 && !by_piece && !(kDense && kDir)) {   // (a dense state in LDS splits by key range, below)
#pragma unroll
      for (int v = 0; v < V; ++v) {
        live[v] = live[v] && static_cast<int>((mix64(code[v]) >> 20) % static_cast<unsigned>(ranges)) == my_range;
      }
    }
    if constexpr (kDirBuild) {
      // distinct codes of the live rows -> LDS set; what does not fit there goes to the directory right away
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if (live[v] && code[v] != kEmptyCode) {
          if (dir->build_step == 1) {
            if (c.wide_words != 0) {
              unsigned long long words[kMaxKeyWords];
#pragma unroll
              for (int w = 0; w < kMaxKeyWords; ++w) words[w] = w < c.wide_words ? key_word_of<V, kStatic>(c, tile, w, trow + v * BLOCK, dec, v) : 0ull;
              bool inserted;
              const int at = lds_find_or_insert_new(l_keys, S, code[v], &inserted);
              if (at < 0) {
                dir_insert_wide(*dir, code[v], words);
              } else if (inserted) {
                // (another key under the same code finds the slot taken and stays out of the directory: global path)
#pragma unroll
                for (int w = 0; w < kMaxKeyWords; ++w) l_set_words[at * kMaxKeyWords + w] = words[w];
              }
              continue;
            }
            if (lds_find_or_insert(l_keys, S, code[v]) < 0) dir_insert(*dir, code[v]);
            continue;
          }
#pragma unroll
          for (int k = 0; k < QSX_MAX_KEYS; ++k) {
            if (k < c.num_keys) {
              const unsigned long long image = static_cast<unsigned long long>(key_field(c, tile, k, trow + v * BLOCK)) ^ kSignBias;
              seen_hi[k] = image > seen_hi[k] ? image : seen_hi[k];
              seen_nlo[k] = ~image > seen_nlo[k] ? ~image : seen_nlo[k];
            }
          }
        }
      }
      continue;
    }
    int slot[V];
    long long global_slot[V];
    bool any_global = false;
    bool any_lds = false;       // REG: a row of this lane goes through the LDS table
    int reg_entry[V];           // REG: the register entry of the row's group, -1: none
    unsigned long long reg_inc[NS > 0 ? NS : 1][V];
    bool run_tail[V];   // kDense: this lane commits the run of equal adjacent keys ending here
    WaveRuns runs[V];   // kDense: the runs of equal adjacent keys of the wave, per row of the thread
    // kDir: the row's group number (or -2: look it up), its wide key, and the aggregates' arguments until the row is classified
    int dir_gid[V];
    unsigned long long dir_words[V][kMaxKeyWords];
    unsigned long long dir_inc[NS > 0 ? NS : 1][V];
    if constexpr (kDense && !kDir) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        global_slot[v] = -1;
        const long long loc = static_cast<long long>(tile_int<V, kStatic>(c, tile, c.key_column[0], trow + v * BLOCK, dec, v));
        if (live[v] && (loc < 0 || loc >= dense.num_entries)) {
          atomicExch(dense.error, 1);  // precondition min >= 0, max < num_entries violated
          live[v] = false;
        }
        // dead rows get a key no neighbour shares so that they never join a run
        const long long run_key = live[v] ? loc : -1 - static_cast<long long>(lane_id());
        runs[v] = wave_runs(run_key);
        run_tail[v] = live[v] && runs[v].tail;
        global_slot[v] = live[v] ? loc : -1;
        slot[v] = runs[v].start;
      }
    } else if constexpr (kDir) {
      // Directory mode reads everything it needs from the tile first (here: the group's position in the key box, or the
      // key words the directory entry has to match; below: the aggregates' arguments) and classifies the row afterwards —
      // with one tile buffer the next tile's DMA then runs under the lookups and the LDS atomics instead of after them.
      // (V rows per thread: the plan shapes take two — the accumulators leave room for one 2048-row tile where the one-row
      // form kept two 1024-row buffers, and with ONE workgroup per CU the bytes in flight per CU are the tile)
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int r = trow + v * BLOCK;
        run_tail[v] = false;
        slot[v] = S + lane_id();   // trash
        global_slot[v] = -1;
        dir_gid[v] = -1;           // -2: to be looked up
#pragma unroll
        for (int w = 0; w < kMaxKeyWords; ++w) dir_words[v][w] = 0ull;
        if constexpr (kDense) {
          // a dense state in LDS: the group number is the key value
          // (ranges > 1: workgroup family r keeps entries [r S, (r + 1) S) and reads every row, like the hash-range families)
          const long long loc = static_cast<long long>(tile_int(c, tile, c.key_column[0], r));
          const long long lo = static_cast<long long>(my_range) * S;
          if (live[v]) {
            if (loc < 0 || loc >= dense.num_entries) {
              atomicExch(dense.error, 1);  // precondition min >= 0, max < num_entries violated
              live[v] = false;
            } else if (loc >= lo && loc < lo + S) {
              dir_gid[v] = (static_cast<int>(loc - lo) << rep_shift) + lane_col;   // (few entries: up to 64 copies, a lane adds into its own)
            } else {
              live[v] = false;             // another family's entry
            }
          }
        } else if (live[v] && code[v] != kEmptyCode) {
          if (box.usable) {
            // group number = position in the key box of the build pass; a key outside the box (the build pass samples) has none
            unsigned int cell = 0;
            bool inside = true;
#pragma unroll
            for (int k = 0; k < QSX_MAX_KEYS; ++k) {
              if (k < c.num_keys) {
                const unsigned long long d = static_cast<unsigned long long>(key_field(c, tile, k, r) - box.lo[k]);
                inside = inside && d < box.range[k];
                cell += static_cast<unsigned int>(d) * box.mult[k];
              }
            }
            dir_gid[v] = inside ? static_cast<int>(cell) : -1;
          } else {
            dir_gid[v] = -2;
            if (c.wide_words != 0) {
#pragma unroll
              for (int w = 0; w < kMaxKeyWords; ++w) dir_words[v][w] = w < c.wide_words ? key_word_of(c, tile, w, r) : 0ull;
            }
          }
        }
      }
    } else {
      if constexpr (REG > 0) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
          reg_entry[v] = -1;
#pragma unroll
          for (int e = 0; e < REG; ++e) {
            if (e < reg_used && live[v] && code[v] == reg_key[e]) reg_entry[v] = e;
          }
        }
      }
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const bool through_lds = live[v] && (REG == 0 || reg_entry[v] < 0);
        classify_row(through_lds, code[v], l_keys, l_acc, S, rep_shift, lane_col, g, slot[v], global_slot[v]);
        any_global = any_global || global_slot[v] >= 0;
        any_lds = any_lds || through_lds;
        run_tail[v] = false;
      }
      if constexpr (REG > 0) {
        // a wave with free entries adopts the keys of rows that found an LDS slot (wave-uniform: ballots and readlanes)
        if (reg_used < REG) {
#pragma unroll
          for (int v = 0; v < V; ++v) {
            unsigned long long candidates = __ballot(live[v] && reg_entry[v] < 0 && slot[v] < (S << rep_shift));
            while (candidates != 0 && reg_used < REG) {
              const int l = __builtin_ctzll(candidates);
              // (readlane returns int: without the casts the low half would sign-extend over the high one)
              const unsigned long long key =
                  static_cast<unsigned long long>(static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(code[v]), l))) |
                  (static_cast<unsigned long long>(static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(code[v] >> 32), l))) << 32);
              candidates &= ~(1ull << l);   // (whatever the comparison below says, this lane is done)
              const int at = __builtin_amdgcn_readlane(slot[v], l) >> rep_shift;
              candidates &= ~__ballot(code[v] == key);
              bool known = false;
#pragma unroll
              for (int e = 0; e < REG; ++e) known = known || (e < reg_used && reg_key[e] == key);
              if (known) continue;   // (adopted from an earlier row of this tile)
#pragma unroll
              for (int e = 0; e < REG; ++e) {
                if (e == reg_used) {
                  reg_key[e] = key;
                  reg_slot[e] = at;
                }
              }
              ++reg_used;
            }
          }
        }
      }
    }
    const bool wave_has_global = __any(any_global);  // rare: groups that did not fit the LDS table
    const bool wave_has_lds = REG == 0 || __any(any_lds);   // (register groups: most waves have no row for the LDS planes)
    if (can_flush && any_global) l_ctl[tile_count % 3] = 1;   // (benign race: everyone stores the same value)
    if constexpr (kDense && !kDir) {
      // existence bit + row count of every run (CollisionFreeVectorTable.hpp:530-645)
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const unsigned long long run_len = static_cast<unsigned long long>(lane_id() - runs[v].start + 1);   // (at the run's tail: its
                                                                                  // length — a dead row is a run of its own)
#ifdef QSX_EXP_NO_DENSE_ATOMICS   // (experiment, tools/agg_dense_exp.sh: the dense per-row path without its global atomics)
        asm volatile("" ::"v"(run_len));
        continue;
#endif
        if (run_tail[v]) {
          const long long loc = global_slot[v];
          const unsigned long long bit = 1ull << (loc & 63);
          unsigned long long *word = &dense.exist[loc >> 6];
          if ((__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) == 0) atomicOr(word, bit);
          if (dense.has_count) atomicAdd(&dense.states[loc], run_len);
        }
      }
    }

    // ---- expression program ------------------------------------------------------
    Temps<V> temps;
    if constexpr (!kStatic) {
      // interpreter: operands and results live in LDS (plan_instrs), nothing of the program in VGPRs
      for (int k = 0; k < c.num_instrs; ++k) {
        const PlanInstr in = c.plan_instrs[k];
        double a[V], b[V], res[V];
        plan_operand_vec<V, BLOCK>(in.a, tile, lds_temps, trow, a);
        plan_operand_vec<V, BLOCK>(in.b, tile, lds_temps, trow, b);
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
        if (in.dst_off >= 0) {
          double *dst = reinterpret_cast<double *>(lds_temps + in.dst_off);
#pragma unroll
          for (int v = 0; v < V; ++v) dst[trow + v * BLOCK] = res[v];
        }
      }
    }
    cfg_for<kStatic, QSX_MAX_INSTRS>(kStatic ? c.num_instrs : 0, [&](int k) __attribute__((always_inline)) {
      const DevInstr in = c.instrs[k];
      double a[V], b[V], res[V];
      operand_vec<V, BLOCK, kStatic>(c, in.a, temps, tile, trow, a, dec);
      operand_vec<V, BLOCK, kStatic>(c, in.b, temps, tile, trow, b, dec);
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
    });

    // ---- accumulate: one LDS atomic per row per accumulator ------------------------
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const DevSum s = c.sums[j];
      unsigned long long inc[V];
      if (s.count_valid) {
        // rows with a non-NULL argument (COUNT(x), AVG's denominator, "saw a value")
#pragma unroll
        for (int v = 0; v < V; ++v) inc[v] = (nullbits[v] & s.null_mask) == 0 ? 1ull : 0ull;
      } else if (s.arg.kind == kOpdKeyWord) {
        // hidden accumulators of a wide key: MIN / MAX of the packed key words (DevConfig::wide_words)
#pragma unroll
        for (int v = 0; v < V; ++v) inc[v] = key_word_of<V, kStatic>(c, tile, s.arg.index, trow + v * BLOCK, dec, v);
      } else if (s.is_int) {
        if constexpr (kStatic) {
#pragma unroll
          for (int v = 0; v < V; ++v) {
            inc[v] = static_cast<unsigned long long>(tile_int<V, kStatic>(c, tile, s.arg.index, trow + v * BLOCK, dec, v));
          }
        } else {
          const PlanSum ps = c.plan_sums[j];
          const char *p = tile + ps.arg.off;
#pragma unroll
          for (int v = 0; v < V; ++v) {
            const int r = trow + v * BLOCK;
            inc[v] = static_cast<unsigned long long>(ps.width == 4 ? static_cast<long long>(reinterpret_cast<const int32_t *>(p)[r])
                                                                    : reinterpret_cast<const long long *>(p)[r]);
          }
        }
      } else {
        double x[V];
        if constexpr (kStatic) {
          operand_vec<V, BLOCK, kStatic>(c, s.arg, temps, tile, trow, x, dec);
        } else {
          plan_operand_vec<V, BLOCK>(c.plan_sums[j].arg, tile, lds_temps, trow, x);
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const long long bits = __double_as_longlong(x[v]);
          inc[v] = static_cast<unsigned long long>(s.kind >= kAccMinI64 ? ordered_from_bits(bits) : bits);
        }
      }
      if (s.count_valid == 0 && s.null_mask != 0) {
        // a NULL argument leaves the accumulator alone (iterateUnaryInl skips it, AggregationHandleSum.hpp:105-120)
#pragma unroll
        for (int v = 0; v < V; ++v) {
          if ((nullbits[v] & s.null_mask) != 0) inc[v] = static_cast<unsigned long long>(acc_identity(s.kind));
        }
      }
      if constexpr (kDense && !kDir) {
        unsigned long long *col = dense.states + static_cast<unsigned long long>((dense.has_count ? 1 : 0) + j) * dense.num_entries;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          unsigned long long run;
          if (s.kind == kAccSumF64) {
            run = static_cast<unsigned long long>(__double_as_longlong(segmented_run_sum_f64(
                runs[v], live[v] ? __longlong_as_double(static_cast<long long>(inc[v])) : 0.0)));
          } else {
            run = segmented_run_combine(runs[v], live[v] ? inc[v] : static_cast<unsigned long long>(acc_identity(s.kind)), s.kind);
          }
#ifdef QSX_EXP_NO_DENSE_ATOMICS
          asm volatile("" ::"v"(run));
          continue;
#endif
          if (run_tail[v]) global_accumulate(&col[global_slot[v]], run, s.kind);
        }
      } else {
        if constexpr (kDir) {
#pragma unroll
          for (int v = 0; v < V; ++v) dir_inc[j][v] = inc[v];
          continue;
        }
        if constexpr (REG > 0) {
#pragma unroll
          for (int v = 0; v < V; ++v) reg_inc[j][v] = inc[v];
        }
#ifndef QSX_EXP_NO_LDS_ADD   // (experiment, tools/agg_coded_exp.sh: what the accumulator atomics cost)
        if (wave_has_lds) {
          unsigned long long *acc_plane = acc_plane_of(j);
#pragma unroll
          for (int v = 0; v < V; ++v) lds_add(&acc_plane[slot[v]], inc[v], s.kind);  // unconditional (trash slot)
        }
#else
#pragma unroll
        for (int v = 0; v < V; ++v) asm volatile("" ::"v"(inc[v]), "v"(slot[v]));
#endif
        if (wave_has_global) {
#pragma unroll
          for (int v = 0; v < V; ++v) {
            if (global_slot[v] >= 0) global_add(g, j + 1, static_cast<unsigned long long>(global_slot[v]), inc[v], s.kind);
          }
        }
      }
    }
    if constexpr (REG > 0) {
      // rows of the wave's register groups: one exec-masked block of adds per (row, entry)
#pragma unroll
      for (int v = 0; v < V; ++v) {
#pragma unroll
        for (int e = 0; e < REG; ++e) {
          if (reg_entry[v] == e) {
            ++reg_cnt[e];
#pragma unroll
            for (int j = 0; j < NS; ++j) reg_acc[e][j] = acc_combine(reg_acc[e][j], reg_inc[j][v], c.sums[j].kind);
          }
        }
      }
    }
    if constexpr (kDir) {
      // every read of this tile is done; the home entry of the row's key is read BEFORE the next tile's DMA is issued —
      // loads return in order, so waiting for a lookup issued behind the DMA would wait for the tile as well
      DirProbe probe[V];
#pragma unroll
      for (int v = 0; v < V; ++v) {
        probe[v] = DirProbe{};
        if (!kDense && dir_gid[v] == -2) probe[v] = dir_first_probe(*dir, code[v]);
      }
      if (nbuf == 1 && next < num_tiles) {
        __syncthreads();
#ifndef QSX_EXP_STAGE_ONCE
        stage_tile<kStatic, BLOCK, kRuns>(c, kRuns ? carried.cols : cols, kRuns ? carried.filter : filter, tiles, carried.row0, carried.rows,
                                          nulls, &carried.bases);
#endif
      }
#pragma unroll
      for (int v = 0; v < V; ++v) {
        int gid = dir_gid[v];
        if (!kDense && gid == -2) {
          gid = c.wide_words != 0 ? dir_lookup_wide_from(*dir, code[v], dir_words[v], probe[v]) : dir_lookup_from(*dir, code[v], probe[v]);
        }
        if (gid >= 0) {
          atomicAdd(&l_cnt[gid], 1u);
#pragma unroll
          for (int j = 0; j < NS; ++j) {
            if (j < ns_lds) lds_add(&acc_plane_of(j)[gid], dir_inc[j][v], c.sums[j].kind);
          }
        } else if (!kDense && live[v]) {
          // no gid with an accumulator here (more groups than the directory was sized for, a group the build pass's sample
          // missed, the sentinel code): the global table, all accumulators — a wide key's hidden ones included
          const unsigned long long gs = global_find_or_insert(g, code[v]);
          if (gs != ~0ull) {
            global_add(g, 0, gs, 1ull, kAccSumI64);
#pragma unroll
            for (int j = 0; j < NS; ++j) global_add(g, j + 1, gs, dir_inc[j][v], c.sums[j].kind);
          }
        }
      }
    }
  }
  if constexpr (kDense && !kDir) return;
  if constexpr (REG > 0) {
    // the lanes' registers -> the planes of the entry's LDS slot (a lane adds into its own copy)
#pragma unroll
    for (int e = 0; e < REG; ++e) {
      if (e < reg_used) {
        const int at = (reg_slot[e] << rep_shift) + lane_col;
        if (reg_cnt[e] != 0) {
          atomicAdd(&l_acc[at], static_cast<unsigned long long>(reg_cnt[e]));
#pragma unroll
          for (int j = 0; j < NS; ++j) lds_add(&acc_plane_of(j)[at], reg_acc[e][j], c.sums[j].kind);
        }
      }
    }
  }
  __syncthreads();
  if constexpr (kDense && kDir) {
    // the workgroup's totals -> the dense arrays: existence bit, row count, one atomic per accumulator and entry it saw
    const int rep = 1 << rep_shift;
    const long long first_entry = static_cast<long long>(my_range) * S;   // (S is a multiple of 64 when there are several ranges)
    for (int base_gid = 0; base_gid < S; base_gid += BLOCK) {
      const int local = base_gid + static_cast<int>(threadIdx.x);
      const long long gid = first_entry + local;
      unsigned long long cnt = 0;
      if (local < S && gid < dense.num_entries) {
        for (int r = 0; r < rep; ++r) cnt += l_cnt[(local << rep_shift) + r];
      }
      // a wave's 64 entries are one existence word: one atomic per word and workgroup (an atomicOr per entry is 64 same-
      // address atomics per word from each of 256 workgroups, all in a handful of memory channels: 1.8 ms for 8000 entries)
      const unsigned long long seen = __ballot(cnt != 0);
      if (lane_id() == 0 && seen != 0) {
        unsigned long long *word = &dense.exist[gid >> 6];
        if ((__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & seen) != seen) atomicOr(word, seen);
      }
      if (cnt == 0) continue;
      if (dense.has_count) atomicAdd(&dense.states[gid], cnt);
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const int kind = c.sums[j].kind;
        const unsigned long long *p = acc_plane_of(j) + (local << rep_shift);
        unsigned long long v = static_cast<unsigned long long>(acc_identity(kind));
        for (int r = 0; r < rep; ++r) v = acc_combine(v, p[r], kind);
        global_accumulate(dense.states + static_cast<unsigned long long>((dense.has_count ? 1 : 0) + j) * dense.num_entries + gid, v, kind);
      }
    }
    return;
  }
  if constexpr (kDirBuild) {
    if (dir->build_step == 1) {
      for (int i = threadIdx.x; i < S; i += BLOCK) {
        const unsigned long long code = l_keys[i];
        if (code == kEmptyCode) continue;
        if (c.wide_words != 0) {
          unsigned long long words[kMaxKeyWords];
#pragma unroll
          for (int w = 0; w < kMaxKeyWords; ++w) words[w] = l_set_words[i * kMaxKeyWords + w];
          dir_insert_wide(*dir, code, words);
        } else {
          dir_insert(*dir, code);
        }
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < QSX_MAX_KEYS; ++k) {
      if (k < c.num_keys) {
        unsigned long long hi = seen_hi[k], nlo = seen_nlo[k];
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
          const unsigned long long h2 = __shfl_xor(hi, off, kWave), n2 = __shfl_xor(nlo, off, kWave);
          hi = h2 > hi ? h2 : hi;
          nlo = n2 > nlo ? n2 : nlo;
        }
        seen_hi[k] = hi;
        seen_nlo[k] = nlo;
      }
    }
    // the workgroup's waves meet in LDS (the tile buffers are free now): one global atomic per word and workgroup — same-
    // address atomics complete one at a time device-wide, 16 waves x 256 workgroups x 4 words of them took 0.2 ms
    unsigned long long *wg_bounds = reinterpret_cast<unsigned long long *>(tiles);
    if (threadIdx.x < 2 * QSX_MAX_KEYS) wg_bounds[threadIdx.x] = 0;
    __syncthreads();
    if (lane_id() == 0) {
#pragma unroll
      for (int k = 0; k < QSX_MAX_KEYS; ++k) {
        if (k < c.num_keys && (seen_hi[k] | seen_nlo[k]) != 0) {
          atomicMax(&wg_bounds[2 * k], seen_hi[k]);
          atomicMax(&wg_bounds[2 * k + 1], seen_nlo[k]);
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < 2 * c.num_keys && wg_bounds[threadIdx.x] != 0) atomicMax(&dir->bounds[threadIdx.x], wg_bounds[threadIdx.x]);
    return;
  }
  if constexpr (kDir) {
    // one global atomic per group this workgroup saw, per accumulator
    const int gids = box.usable ? box.cells : static_cast<int>(dir->lds_gids);
    for (int gid = threadIdx.x; gid < gids; gid += BLOCK) {
      const unsigned long long cnt = l_cnt[gid];
      if (cnt == 0) continue;
      unsigned long long code;
      unsigned long long words[kMaxKeyWords] = {};   // the group's key, word by word (narrow key: word 0 = the code)
      if (box.usable) {
#pragma unroll
        for (int k = 0; k < QSX_MAX_KEYS; ++k) {
          if (k < c.num_keys) {
            const unsigned long long field = static_cast<unsigned long long>(
                box.lo[k] + static_cast<long long>((static_cast<unsigned int>(gid) / box.mult[k]) % box.range[k]));
            const unsigned long long mask = c.key_width[k] >= 8 ? ~0ull : (1ull << (8 * c.key_width[k])) - 1;
#pragma unroll
            for (int w = 0; w < kMaxKeyWords; ++w) {
              if ((c.wide_words != 0 ? c.key_word[k] : 0) == w) words[w] |= (field & mask) << c.key_shift[k];
            }
          }
        }
        code = words[0];
      } else if (c.wide_words != 0) {
#pragma unroll
        for (int w = 0; w < kMaxKeyWords; ++w) {
          words[w] = __hip_atomic_load(&dir->words_by_gid[static_cast<size_t>(gid) * kMaxKeyWords + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        code = 0;
      } else {
        code = __hip_atomic_load(&dir->codes_by_gid[gid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (c.wide_words != 0) code = wide_key_code(words, c.wide_words, c.wide_hash_mask);
      const unsigned long long gs = global_find_or_insert(g, code);
      if (gs == ~0ull) continue;
      global_add(g, 0, gs, cnt, kAccSumI64);
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        if (j < ns_lds) {
          global_add(g, j + 1, gs, acc_plane_of(j)[gid], c.sums[j].kind);
        } else {
          // hidden MIN / MAX of key word (j - ns_lds) / 2: every row of this gid carried exactly these words
          const int w = (j - ns_lds) >> 1;
          global_add(g, j + 1, gs, w == 0 ? words[0] : (w == 1 ? words[1] : words[2]), c.sums[j].kind);
        }
      }
    }
    return;
  }

  flush_table(false);
}

// Interpreter: the configuration arrives as a kernel argument (A/B against a pointer to a device copy: no difference,
// 4.02 vs 4.04 ms per 200 M Q1 rows — the interpreter is bound by the scalar instructions it issues).
static_assert(sizeof(DevConfig) <= 3840, "DevConfig travels by value: the kernarg segment is limited to 4 KiB");
template <int NS, int V>
__global__ __launch_bounds__(kABlock) void agg_hash_update_kernel(DevConfig c, int64_t n,
                                                                 const uint64_t *__restrict__ filter,
                                                                 HashTableView g, int S, int rep_shift, int nbuf,
                                                                 int ranges, const long long *__restrict__ pieces) {
  agg_hash_update_body<false, false, NS, V>(c, c.cols, c.dicts, n, filter, g, DenseView{}, S, rep_shift, nbuf, ranges, pieces, c.nulls);
}

// COLLISION_FREE (K7) through the same staged-tile body with the dense sink.
template <int NS, int V>
__global__ __launch_bounds__(kABlock) void agg_dense_update_kernel(DevConfig c, int64_t n,
                                                                  const uint64_t *__restrict__ filter, DenseView d,
                                                                  int nbuf) {
  agg_hash_update_body<false, true, NS, V>(c, c.cols, c.dicts, n, filter, HashTableView{}, d, 8, 0, nbuf, 1, nullptr, c.nulls);
}

struct ColumnPointers {
  const void *p[QSX_MAX_COLUMNS];
};

// The same kernels over a run of blocks (qsx_agg_update_blocks): stripes and filters come from the table, has_filter only
// says that the plan stages filter words (c.filter_lds_off).
template <int NS, int V>
__global__ __launch_bounds__(kABlock) void agg_hash_update_runs_kernel(DevConfig c, int64_t n, HashTableView g, int S, int rep_shift, int nbuf,
                                                                      int ranges, const long long *__restrict__ block_run) {
  agg_hash_update_body<false, false, NS, V, false, kABlock, false, true>(c, nullptr, nullptr, n, nullptr, g, DenseView{}, S, rep_shift, nbuf,
                                                                         ranges, block_run, nullptr);
}
template <int NS, int V>
__global__ __launch_bounds__(kABlock) void agg_dense_update_runs_kernel(DevConfig c, int64_t n, DenseView d, int nbuf,
                                                                       const long long *__restrict__ block_run) {
  agg_hash_update_body<false, true, NS, V, false, kABlock, false, true>(c, nullptr, nullptr, n, nullptr, HashTableView{}, d, 8, 0, nbuf, 1,
                                                                        block_run, nullptr);
}

// A dense state in LDS (kDense && kDir of the body): `entries` accumulators per aggregate x 2^rep_shift copies per workgroup;
// `ranges` families of workgroups split a state of up to ranges x entries entries between them.
template <int NS, bool kRuns>
__global__ __launch_bounds__(1024) void agg_dense_lds_kernel(DevConfig c, int64_t n, const uint64_t *__restrict__ filter, DenseView d,
                                                            int entries, int rep_shift, int nbuf, int ranges,
                                                            const long long *__restrict__ block_run) {
  agg_hash_update_body<false, true, NS, 1, true, 1024, false, kRuns>(c, kRuns ? nullptr : c.cols, kRuns ? nullptr : c.dicts, n,
                                                                    kRuns ? nullptr : filter, HashTableView{}, d, entries, rep_shift, nbuf, ranges,
                                                                    block_run, kRuns ? nullptr : c.nulls, nullptr);
}

// Group-directory variant (agg_common.hpp DirView): ONE workgroup of 1024 threads per CU owns the CU's LDS — `gids`
// accumulators per aggregate, unreplicated — and 16 waves hide the directory's L2 round trip.  One row per thread and tile.
constexpr int kDirBlock = 1024;
// kRuns: the rows are a run of blocks (block_run = the table of agg_common.hpp BlockRunView; stripes, filters and dictionaries
// come from it).
template <int NS, bool kRuns>
__global__ __launch_bounds__(kDirBlock) void agg_dir_update_kernel(DevConfig c, int64_t n, const uint64_t *__restrict__ filter,
                                                                  HashTableView g, DirView d, int gids, int nbuf,
                                                                  const long long *__restrict__ block_run) {
  agg_hash_update_body<false, false, NS, 1, true, kDirBlock, false, kRuns>(c, kRuns ? nullptr : c.cols, kRuns ? nullptr : c.dicts, n,
                                                                           kRuns ? nullptr : filter, g, DenseView{}, gids, 0, nbuf, 1, block_run,
                                                                           kRuns ? nullptr : c.nulls, &d);
}
// Build pass: key (and predicate) columns only, `set_slots` LDS slots for the workgroup's distinct codes.
template <bool kRuns>
__global__ __launch_bounds__(kDirBlock) void agg_dir_build_kernel(DevConfig c, int64_t n, const uint64_t *__restrict__ filter, DirView d,
                                                                 int set_slots, int nbuf, const long long *__restrict__ block_run) {
  agg_hash_update_body<false, false, 0, 1, false, kDirBlock, true, kRuns>(c, kRuns ? nullptr : c.cols, kRuns ? nullptr : c.dicts, n,
                                                                          kRuns ? nullptr : filter, HashTableView{}, DenseView{}, set_slots, 0,
                                                                          nbuf, 1, block_run, kRuns ? nullptr : c.nulls, &d);
}
template <typename Shape>
__global__ __launch_bounds__(kDirBlock) void agg_dir_shape_kernel(ColumnPointers cols, int64_t n, HashTableView g, DirView d, int gids,
                                                                 int nbuf) {
  static constexpr Translated T = Shape::translated(kDirBlock);
  agg_hash_update_body<true, false, T.num_sums, 1, true, kDirBlock>(T.dev, cols.p, nullptr, n, nullptr, g, DenseView{}, gids, 0, nbuf, 1,
                                                                    nullptr, nullptr, &d);
}
// The same with one 2048-row tile, two rows per thread, in the LDS the two 1024-row buffers take (nbuf = 1 in the body).
template <typename Shape>
__global__ __launch_bounds__(kDirBlock) void agg_dir_shape_wide_kernel(ColumnPointers cols, int64_t n, HashTableView g, DirView d, int gids) {
  static constexpr Translated T = Shape::translated(2 * kDirBlock);
  agg_hash_update_body<true, false, T.num_sums, 2, true, kDirBlock>(T.dev, cols.p, nullptr, n, nullptr, g, DenseView{}, gids, 0, 1, 1,
                                                                    nullptr, nullptr, &d);
}
template <typename Shape>
__global__ __launch_bounds__(kDirBlock) void agg_dir_shape_runs_kernel(int64_t n, HashTableView g, DirView d, int gids, int nbuf,
                                                                      const long long *__restrict__ block_run) {
  static constexpr Translated T = Shape::translated(kDirBlock);
  agg_hash_update_body<true, false, T.num_sums, 1, true, kDirBlock, false, true>(T.dev, nullptr, nullptr, n, nullptr, g, DenseView{}, gids, 0,
                                                                                 nbuf, 1, block_run, nullptr, &d);
}


// AOT plan shape: the translated configuration is a function-local static constexpr
// object, i.e. a true constant of the code object, so after the configuration loops are
// unrolled every load from it constant-folds and the interpreter disappears: what is left
// is the straight-line arithmetic of that plan.
template <typename Shape, int V>
__global__ __launch_bounds__(kABlock) void agg_hash_shape_kernel(ColumnPointers cols, int64_t n, HashTableView g, int S,
                                                                int rep_shift, int nbuf, int ranges,
                                                                const long long *__restrict__ pieces) {
  static constexpr Translated T = Shape::translated(kABlock * V);
  agg_hash_update_body<true, false, T.num_sums, V>(T.dev, cols.p, nullptr, n, nullptr, g, DenseView{}, S, rep_shift, nbuf, ranges, pieces);
}

template <typename Shape, int V>
__global__ __launch_bounds__(kABlock) void agg_hash_shape_runs_kernel(int64_t n, HashTableView g, int S, int rep_shift, int nbuf, int ranges,
                                                                     const long long *__restrict__ block_run) {
  static constexpr Translated T = Shape::translated(kABlock * V);
  agg_hash_update_body<true, false, T.num_sums, V, false, kABlock, false, true>(T.dev, nullptr, nullptr, n, nullptr, g, DenseView{}, S, rep_shift,
                                                                                nbuf, ranges, block_run);
}

// Register groups of a fixed-geometry shape: tables sized for a handful of groups (the launchers' smallest, 16 slots), and
// accumulators that fit the register file next to the body's own (REG x (2 NS + 1) VGPRs).
#ifndef QSX_AGG_REG_GROUPS
#define QSX_AGG_REG_GROUPS 4
#endif
constexpr int reg_groups_for(int slots, int num_sums) {
  return slots <= 16 && num_sums >= 1 && num_sums <= 6 ? QSX_AGG_REG_GROUPS : 0;
}

template <typename Shape, int V, int S_, int REP_SHIFT_, int RANGES_, bool kRegGroups = false>
__global__ __launch_bounds__(kABlock) void agg_hash_shape_fixed_runs_kernel(int64_t n, HashTableView g, const long long *__restrict__ block_run) {
  static constexpr Translated T = Shape::translated(kABlock * V);
  agg_hash_update_body<true, false, T.num_sums, V, false, kABlock, false, true, kRegGroups ? reg_groups_for(S_, T.num_sums) : 0>(
      T.dev, nullptr, nullptr, n, nullptr, g, DenseView{}, S_, REP_SHIFT_, 1, RANGES_, block_run);
}

// The same with the launch geometry of the common small-group case (one tile buffer, one workgroup family) as constants:
// the plane / slot arithmetic of the tile loop folds instead of occupying scalar registers (what made the run-time shapes
// 10 % faster once they got their geometry as constants).
template <typename Shape, int V, int S_, int REP_SHIFT_, int RANGES_, bool kRegGroups = false>
__global__ __launch_bounds__(kABlock) void agg_hash_shape_fixed_kernel(ColumnPointers cols, int64_t n, HashTableView g,
                                                                      const long long *__restrict__ pieces) {
  static constexpr Translated T = Shape::translated(kABlock * V);
  agg_hash_update_body<true, false, T.num_sums, V, false, kABlock, false, false, kRegGroups ? reg_groups_for(S_, T.num_sums) : 0>(
      T.dev, cols.p, nullptr, n, nullptr, g, DenseView{}, S_, REP_SHIFT_, 1, RANGES_, pieces);
}

}  // namespace qsx

#endif  // QSX_CSRC_AGG_HASH_UPDATE_HPP_
