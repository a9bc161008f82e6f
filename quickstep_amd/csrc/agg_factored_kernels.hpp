// agg_factored_kernels.hpp — device code of the factored aggregation (agg_factored.hpp has the story); included by
// agg_factored.hip only.
#ifndef QSX_CSRC_AGG_FACTORED_KERNELS_HPP_
#define QSX_CSRC_AGG_FACTORED_KERNELS_HPP_

#include "agg_factored.hpp"
#include "agg_hash_update.hpp"

namespace qsx {

__device__ inline double factored_dict_value(const DevConfig &c, int col, int code) {
  const void *d = as_global(c.dicts[col]);
  switch (c.column_type[col]) {
    case QSX_INT: return static_cast<double>(static_cast<const int32_t *>(d)[code]);
    case QSX_LONG: return static_cast<double>(static_cast<const long long *>(d)[code]);
    case QSX_FLOAT: return static_cast<double>(static_cast<const float *>(d)[code]);
    default: return static_cast<const double *>(d)[code];
  }
}
__device__ inline long long factored_dict_int(const DevConfig &c, int col, int code) {
  const void *d = as_global(c.dicts[col]);
  return c.column_type[col] == QSX_INT ? static_cast<long long>(static_cast<const int32_t *>(d)[code]) : static_cast<const long long *>(d)[code];
}
// Element i of a small array that lives in registers: a chain of selects (indexing it with i would put the array into scratch
// memory — the first form of the coefficient kernel moved ~2 KB of scratch per thread and block: 0.2 ms for 1860 blocks).
template <int N>
__device__ __forceinline__ double reg_pick(const double (&a)[N], int i) {
  double r = a[0];
#pragma unroll
  for (int k = 1; k < N; ++k) r = i == k ? a[k] : r;
  return r;
}
template <int N>
__device__ __forceinline__ void reg_put(double (&a)[N], int i, double v) {
#pragma unroll
  for (int k = 0; k < N; ++k) a[k] = i == k ? v : a[k];
}
// The expression program over one assignment of column values (every node rounded on its own, like the row-wise
// evaluation: -ffp-contract=off): all temporaries at once — every aggregate argument is then one operand read.
// c: the configuration in LDS.
__device__ __forceinline__ double factored_operand(const DevConfig &c, const double (&vals)[QSX_MAX_COLUMNS], const double (&temps)[QSX_MAX_TEMPS],
                                                   const DevOperand &o) {
  if (o.kind == QSX_OPD_COLUMN) return reg_pick(vals, o.index);
  if (o.kind == QSX_OPD_CONST) return c.consts[o.index];
  return reg_pick(temps, o.index);
}
__device__ inline void factored_eval_all(const DevConfig &c, const double (&vals)[QSX_MAX_COLUMNS], double (&temps)[QSX_MAX_TEMPS]) {
#pragma unroll
  for (int t = 0; t < QSX_MAX_TEMPS; ++t) temps[t] = 0.0;
  for (int k = 0; k < c.num_instrs; ++k) {
    const double a = factored_operand(c, vals, temps, c.instrs[k].a), b = factored_operand(c, vals, temps, c.instrs[k].b);
    double r;
    switch (c.instrs[k].op) {
      case QSX_EX_ADD: r = a + b; break;
      case QSX_EX_SUB: r = a - b; break;
      case QSX_EX_MUL: r = a * b; break;
      default: r = a / b; break;
    }
    reg_put(temps, c.instrs[k].dst, r);
  }
}
__global__ __launch_bounds__(kABlock) void factored_coef_kernel(DevConfig c_arg, FactoredCoefArgs a_all) {
  // (the configuration is indexed by thread-dependent values: from the kernel argument that is a private copy of all of it per
  // lane — 2.7 KB of scratch, 32 us for 600 threads; from LDS it is a handful of ds_reads)
  __shared__ DevConfig c;
  __shared__ FactoredCoefArgs a_lds;           // (indexed by thread-dependent values like the configuration)
  __shared__ int s_entries[QSX_MAX_COLUMNS];   // entries of this block's dictionaries (a run of blocks); kFacMaxDict otherwise
  {
    const unsigned int *src = reinterpret_cast<const unsigned int *>(&c_arg);
    unsigned int *dst = reinterpret_cast<unsigned int *>(&c);
    for (int i = threadIdx.x; i < static_cast<int>(sizeof(DevConfig) / 4); i += kABlock) dst[i] = src[i];
    static_assert(sizeof(FactoredCoefArgs) % 4 == 0, "copied word by word");
    const unsigned int *asrc = reinterpret_cast<const unsigned int *>(&a_all);
    unsigned int *adst = reinterpret_cast<unsigned int *>(&a_lds);
    for (int i = threadIdx.x; i < static_cast<int>(sizeof(FactoredCoefArgs) / 4); i += kABlock) adst[i] = asrc[i];
  }
  __syncthreads();
  // (a code beyond the block's own dictionary does not occur in the block; its coefficient is that of the last entry)
  auto clamped = [&](int col, int code) { return code < s_entries[col] ? code : s_entries[col] - 1; };
  // One block's tables.  A run of blocks: workgroup row y takes blocks y, y + gridDim.y, ... — the configuration is copied to LDS
  // once per workgroup, not once per block (a workgroup per block: 0.19 ms for the 1860 blocks of 600 M lineitems).
  // A thread per cell and per dictionary code: the program runs once per assignment (plus once per carrier) and serves every
  // sum of the state (a thread per (sum, cell) ran it eight times over for Q1).
  auto one_block = [&](int blk) {
    const FactoredCoefArgs &a = a_lds;
    unsigned long long *const coef_b = a_all.coef + static_cast<size_t>(blk) * a_all.coef_words;
    unsigned long long *const hcoef_b = a_all.hcoef + static_cast<size_t>(blk) * a_all.hcoef_words;
    const int i = blockIdx.x * kABlock + threadIdx.x;
    if (i >= a.cells + kFacMaxDict) return;
    double vals[QSX_MAX_COLUMNS], temps[QSX_MAX_TEMPS];
#pragma unroll
    for (int col = 0; col < QSX_MAX_COLUMNS; ++col) vals[col] = 0.0;
    if (i >= a.cells) {     // the histogram coefficients of every sum: H[j][code]
      const int code = i - a.cells;
      for (int j = 0; j < a.nsums; ++j) if (a.sum_hist[j] < 0) hcoef_b[j * kFacMaxDict + code] = 0;
      for (int h = 0; h < a.nhist; ++h) {
        const int col = a.hist_col[h];
        const bool in_range = code < a.hist_size[h];
        const double v = factored_dict_value(c, col, clamped(col, code));
        reg_put(vals, col, v);
        factored_eval_all(c, vals, temps);
        for (int j = 0; j < a.nsums; ++j) {
          if (a.sum_hist[j] != h) continue;
          unsigned long long word;
          if (c.sums[j].kind == kAccSumI64) {
            word = static_cast<unsigned long long>(factored_dict_int(c, col, clamped(col, code)));
          } else {
            word = static_cast<unsigned long long>(__double_as_longlong(factored_operand(c, vals, temps, c.sums[j].arg)));
          }
          hcoef_b[j * kFacMaxDict + code] = in_range ? word : 0ull;
        }
        reg_put(vals, col, 0.0);
      }
      return;
    }
    const int cell = i;
    int cell_code[kFacMaxCell] = {};
#pragma unroll
    for (int q = 0; q < kFacMaxCell; ++q) {
      if (q >= a.ncell) break;
      cell_code[q] = clamped(a.cell_col[q], (cell / a.cell_stride[q]) % a.cell_radix[q]);
      reg_put(vals, a.cell_col[q], factored_dict_value(c, a.cell_col[q], cell_code[q]));
    }
    // every carrier 0: the part that multiplies the count; then one carrier at 1 and the others at 0
    double temps_car[kFacMaxCarriers][QSX_MAX_TEMPS];
    factored_eval_all(c, vals, temps);
#pragma unroll
    for (int k = 0; k < kFacMaxCarriers; ++k) {
      if (k >= a.ncar) break;
      reg_put(vals, a.car_col[k], 1.0);
      factored_eval_all(c, vals, temps_car[k]);
      reg_put(vals, a.car_col[k], 0.0);
    }
    for (int j = 0; j < a.nsums; ++j) {
      unsigned long long *out = coef_b + static_cast<size_t>(j) * (1 + a.ncar) * a.cells;
      if (a.sum_hist[j] >= 0) {   // depends on a histogram column only: nothing comes from the cells
        for (int k = 0; k <= a.ncar; ++k) out[static_cast<size_t>(k) * a.cells + cell] = 0;
        continue;
      }
      const DevOperand arg = c.sums[j].arg;
      if (c.sums[j].kind == kAccSumI64) {
        // SUM over an INT / LONG column: the cell's integer dictionary value times its count (a plain integer column is a
        // carrier with an i64 plane and needs no coefficient, FactoredArgs::sum_car_int); COUNT-like sums of the constant: 1
        long long v = 1;
        if (arg.kind == QSX_OPD_COLUMN) {
          v = 0;
#pragma unroll
          for (int q = 0; q < kFacMaxCell; ++q) {
            if (q < a.ncell && a.cell_col[q] == arg.index) v = factored_dict_int(c, a.cell_col[q], cell_code[q]);
          }
        }
        out[cell] = static_cast<unsigned long long>(v);
        for (int k = 1; k <= a.ncar; ++k) out[static_cast<size_t>(k) * a.cells + cell] = 0;
        continue;
      }
      const double a0 = factored_operand(c, vals, temps, arg);
      out[cell] = static_cast<unsigned long long>(__double_as_longlong(a0));
#pragma unroll
      for (int k = 0; k < kFacMaxCarriers; ++k) {
        if (k >= a.ncar) break;
        // (the argument itself may be the carrier column: its value in this evaluation is 1)
        const double with_k = arg.kind == QSX_OPD_COLUMN ? (arg.index == a.car_col[k] ? 1.0 : reg_pick(vals, arg.index))
                                                         : factored_operand(c, vals, temps_car[k], arg);
        out[static_cast<size_t>(k + 1) * a.cells + cell] = static_cast<unsigned long long>(__double_as_longlong(with_k - a0));
      }
    }
  };
  for (int blk = blockIdx.y; blk < a_all.num_blocks; blk += gridDim.y) {
    __syncthreads();   // the previous block's readers are done with the dictionary pointers
    if (threadIdx.x < QSX_MAX_COLUMNS) {
      s_entries[threadIdx.x] = kFacMaxDict;
      if (a_all.run_dicts != nullptr) {
        const size_t at = static_cast<size_t>(blk) * QSX_MAX_COLUMNS + threadIdx.x;
        c.dicts[threadIdx.x] = reinterpret_cast<const void *>(static_cast<uintptr_t>(a_all.run_dicts[at]));
        s_entries[threadIdx.x] = a_all.run_entries[at] > 0 ? a_all.run_entries[at] : 1;
      }
    }
    __syncthreads();
    one_block(blk);
  }
}

// ---- the state's predicate as a filter bitmap (FactoredPredArgs) ---------------------------------------------------------
// (branch-free in the operator: the six comparisons from "less" and "equal" — a switch per row would put every load of the
// pass into a basic block of its own)
template <typename T>
__device__ __forceinline__ bool factored_compare(T a, int op, T b) {
  const bool lt = a < b, eq = a == b;
  return op == QSX_EQ ? eq : (op == QSX_NE ? !eq : (op == QSX_LT ? lt : (op == QSX_LE ? (lt || eq) : (op == QSX_GT ? !(lt || eq) : !lt))));
}
__device__ __forceinline__ bool factored_pred_holds(int type, unsigned long long raw, int op, unsigned long long literal) {
  switch (type) {
    case QSX_INT: return factored_compare<int32_t>(static_cast<int32_t>(raw), op, static_cast<int32_t>(literal));
    case QSX_LONG: return factored_compare<int64_t>(static_cast<int64_t>(raw), op, static_cast<int64_t>(literal));
    case QSX_FLOAT: return factored_compare<float>(__uint_as_float(static_cast<uint32_t>(raw)), op, __uint_as_float(static_cast<uint32_t>(literal)));
    case QSX_DATE: return factored_compare<long long>(date_ordered(raw), op, date_ordered(literal));
    default: return factored_compare<double>(__longlong_as_double(static_cast<long long>(raw)), op, __longlong_as_double(static_cast<long long>(literal)));
  }
}
// Sixteen elements of a stripe of `width`-byte elements, element at[w] each: the width is decided once, the sixteen loads are
// issued back to back.
__device__ __forceinline__ void factored_load16(const void *base, int width, const uint32_t (&at)[16], unsigned long long (&raw)[16]) {
  switch (width) {
    case 1:
#pragma unroll
      for (int w = 0; w < 16; ++w) raw[w] = load_global(static_cast<const uint8_t *>(base) + at[w]);
      break;
    case 2:
#pragma unroll
      for (int w = 0; w < 16; ++w) raw[w] = load_global(static_cast<const uint16_t *>(base) + at[w]);
      break;
    case 4:
#pragma unroll
      for (int w = 0; w < 16; ++w) raw[w] = load_global(static_cast<const uint32_t *>(base) + at[w]);
      break;
    default:
#pragma unroll
      for (int w = 0; w < 16; ++w) raw[w] = load_global(static_cast<const unsigned long long *>(base) + at[w]);
      break;
  }
}
// A WAVE per 1024-row tile: the 16 values a lane needs of a term's column (rows lane, 64 + lane, ...) are requested together
// — 4 KB in flight per wave and term — then compared; lane w keeps bitmap word w and the tile's 16 words leave with one store.
// (A first form read one 256-byte row of a wave at a time: 1.5 ms per 600 M rows of a 4-byte column, a quarter of what the
// column's bytes take.)
__global__ __launch_bounds__(kABlock) void factored_predicate_kernel(FactoredPredArgs a, long long tiles) {
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  constexpr int kWaves = kABlock / kWave;
  for (long long tile = static_cast<long long>(blockIdx.x) * kWaves + wave; tile < tiles; tile += static_cast<long long>(gridDim.x) * kWaves) {
    const unsigned long long *filter_in = a.filter_in;
    long long row0 = tile * 1024, rows = a.n, b = 0;
    BlockRunView run{};
    if (a.run != nullptr) {
      run = block_run_view(a.run, 1024);
      b = block_of_tile(run, tile);
      row0 = (tile - run.first_tile[b]) * 1024;
      rows = run.rows[b];
      filter_in = a.filters_in != nullptr ? reinterpret_cast<const unsigned long long *>(static_cast<uintptr_t>(a.filters_in[b])) : nullptr;
    }
    unsigned int ok = 0xFFFFu;   // bit w: row row0 + 64 w + lane passes every term so far
    uint32_t at[16];   // (a stripe of the boundary holds fewer than 2^31 rows: qsx_agg_update's tuple ids are 32-bit)
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const long long row = row0 + w * 64 + lane;
      at[w] = static_cast<uint32_t>(row < rows ? row : rows - 1);   // clamped, not guarded
    }
    for (int p = 0; p < a.num_pred; ++p) {
      // (the term's stripe and dictionary: wave-uniform reads of the kernel argument / the run table)
      const void *col_p = a.col[p], *dict_p = a.dict[p];
      if (a.run != nullptr) {
        col_p = run.cols[b * QSX_MAX_COLUMNS + a.pred[p].column];
        dict_p = run.dicts != nullptr ? run.dicts[b * QSX_MAX_COLUMNS + a.pred[p].column] : nullptr;
      }
      unsigned long long raw[16];
      factored_load16(col_p, a.width[p], at, raw);
      if (a.coded[p] != 0 && dict_p != nullptr) {   // (a truncation-compressed attribute: value = code)
        uint32_t code[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) code[w] = static_cast<uint32_t>(raw[w]);
        factored_load16(dict_p, a.type[p] == QSX_INT || a.type[p] == QSX_FLOAT ? 4 : 8, code, raw);
      }
      const int type = a.type[p], op = a.pred[p].op;
      const unsigned long long literal = a.pred[p].literal;
      switch (type) {   // (decided once per term, not per row)
        case QSX_INT:
#pragma unroll
          for (int w = 0; w < 16; ++w) ok &= factored_compare<int32_t>(static_cast<int32_t>(raw[w]), op, static_cast<int32_t>(literal)) ? ~0u : ~(1u << w);
          break;
        case QSX_DOUBLE:
#pragma unroll
          for (int w = 0; w < 16; ++w) {
            ok &= factored_compare<double>(__longlong_as_double(static_cast<long long>(raw[w])), op, __longlong_as_double(static_cast<long long>(literal))) ? ~0u : ~(1u << w);
          }
          break;
        default:
#pragma unroll
          for (int w = 0; w < 16; ++w) ok &= factored_pred_holds(type, raw[w], op, literal) ? ~0u : ~(1u << w);
          break;
      }
    }
    unsigned long long mine = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const long long row = row0 + w * 64 + lane;
      const unsigned long long bits = msb_first(__ballot(row < rows && ((ok >> w) & 1u) != 0u));
      if (lane == w) mine = bits;
    }
    const long long word = (row0 >> 6) + lane;
    if (lane < 16 && word * 64 < rows) {
      if (filter_in != nullptr) mine &= load_global(&filter_in[word]);
      (a.out + tile * 16)[lane] = mine;
    }
  }
}

// ---- accumulate ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long factored_read(const char *tile, int off, int width, int row) {
  switch (width) {
    case 1: return reinterpret_cast<const uint8_t *>(tile + off)[row];
    case 2: return reinterpret_cast<const uint16_t *>(tile + off)[row];
    case 4: return reinterpret_cast<const uint32_t *>(tile + off)[row];
    default: return reinterpret_cast<const unsigned long long *>(tile + off)[row];
  }
}
__device__ __forceinline__ unsigned long long factored_carrier_word(unsigned long long raw, int type, bool as_int) {
  if (as_int) return type == QSX_INT ? static_cast<unsigned long long>(static_cast<long long>(static_cast<int32_t>(raw))) : raw;
  double v;
  switch (type) {
    case QSX_INT: v = static_cast<double>(static_cast<int32_t>(raw)); break;
    case QSX_LONG: v = static_cast<double>(static_cast<long long>(raw)); break;
    case QSX_FLOAT: v = static_cast<double>(__uint_as_float(static_cast<uint32_t>(raw))); break;
    default: v = __longlong_as_double(static_cast<long long>(raw)); break;
  }
  return static_cast<unsigned long long>(__double_as_longlong(v));
}

// The workgroup's cells -> the state: a wave per group slot; every accumulator of the state is a dot product over the slot's
// cells (coefficients from this call's dictionaries, factored_coef_kernel).
__device__ __forceinline__ void factored_flush(const FactoredArgs &a, const unsigned long long *l_keys, const unsigned long long *l_plane,
                                               const unsigned int *l_cnt, const unsigned int *l_hist, const HashTableView &g,
                                               const unsigned long long *coef, const unsigned long long *hcoef) {
  const int S = a.S, cells = a.cells;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  for (int s = wave; s < S; s += kABlock / kWave) {
    const unsigned long long code = l_keys[s];
    if (code == kEmptyCode) continue;
    unsigned long long gs = 0;
    if (lane == 0) gs = global_find_or_insert(g, code);
    gs = __shfl(gs, 0, kWave);
    if (gs == ~0ull) continue;
    const unsigned int *cnt = l_cnt + static_cast<size_t>(s) * cells;
    unsigned long long rows_in_group = 0;
    for (int c = lane; c < cells; c += kWave) rows_in_group += cnt[c];
    rows_in_group = wave_reduce_add(rows_in_group);
    if (lane == 0) global_add(g, 0, gs, rows_in_group, kAccSumI64);
    for (int j = 0; j < a.nsums; ++j) {
      const unsigned long long *cj = coef + static_cast<size_t>(j) * (1 + a.ncar) * cells;
      if (a.sum_kind[j] == kAccSumI64) {
        long long acc = 0;
        if (a.sum_car_int[j] >= 0) {
          const unsigned long long *plane = l_plane + static_cast<size_t>(a.sum_car_int[j]) * S * cells + static_cast<size_t>(s) * cells;
          for (int c = lane; c < cells; c += kWave) acc += static_cast<long long>(plane[c]);
        } else if (a.sum_hist[j] >= 0) {
          const int h = a.sum_hist[j];
          const unsigned int *hist = l_hist + s * a.hist_words + a.hist_off[h];
          for (int c = lane; c < a.hist_size[h]; c += kWave) acc += static_cast<long long>(hcoef[j * kFacMaxDict + c]) * static_cast<long long>(hist[c]);
        } else {
          for (int c = lane; c < cells; c += kWave) acc += static_cast<long long>(cj[c]) * static_cast<long long>(cnt[c]);
        }
        const unsigned long long total = wave_reduce_add(static_cast<unsigned long long>(acc));
        if (lane == 0) global_add(g, j + 1, gs, total, kAccSumI64);
      } else {
        double acc = 0.0;
        if (a.sum_hist[j] >= 0) {
          const int h = a.sum_hist[j];
          const unsigned int *hist = l_hist + s * a.hist_words + a.hist_off[h];
          for (int c = lane; c < a.hist_size[h]; c += kWave) {
            acc += __longlong_as_double(static_cast<long long>(hcoef[j * kFacMaxDict + c])) * static_cast<double>(hist[c]);
          }
        } else {
          for (int c = lane; c < cells; c += kWave) {
            if (cnt[c] == 0u) continue;
            double term = __longlong_as_double(static_cast<long long>(cj[c])) * static_cast<double>(cnt[c]);
            for (int k = 0; k < a.ncar; ++k) {
              if (a.car_int[k] != 0) continue;
              const unsigned long long *plane = l_plane + static_cast<size_t>(k) * S * cells + static_cast<size_t>(s) * cells;
              term += __longlong_as_double(static_cast<long long>(cj[static_cast<size_t>(k + 1) * cells + c])) * __longlong_as_double(static_cast<long long>(plane[c]));
            }
            acc += term;
          }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
        if (lane == 0) global_add(g, j + 1, gs, static_cast<unsigned long long>(__double_as_longlong(acc)), kAccSumF64);
      }
    }
  }
}

// One row that found no slot in the workgroup's table: its terms go straight to the state's table (a group the optimizer's
// estimate did not foresee; slow and exact).  hist_code[h]: the row's code of histogram column h.
__device__ __forceinline__ void factored_spill_row(const FactoredArgs &a, const HashTableView &g, unsigned long long code, int cell,
                                                   const int (&hist_code)[kFacMaxHist], unsigned long long car0, unsigned long long car1,
                                                   const unsigned long long *coef, const unsigned long long *hcoef) {
  const int cells = a.cells;
  const unsigned long long gs = global_find_or_insert(g, code);
  if (gs == ~0ull) return;
  global_add(g, 0, gs, 1ull, kAccSumI64);
  for (int j = 0; j < a.nsums; ++j) {
    const unsigned long long *cj = coef + static_cast<size_t>(j) * (1 + a.ncar) * cells;
    const int h = a.sum_hist[j];
    int hc = 0;
    if (h >= 0) {
      hc = h == 0 ? hist_code[0] : (h == 1 ? hist_code[1] : (h == 2 ? hist_code[2] : hist_code[3]));
      hc = hc < a.hist_size[h] ? hc : a.hist_size[h] - 1;
    }
    if (a.sum_kind[j] == kAccSumI64) {
      long long inc = 0;
      if (a.sum_car_int[j] >= 0) inc = static_cast<long long>(a.sum_car_int[j] == 0 ? car0 : car1);
      else if (h >= 0) inc = static_cast<long long>(hcoef[j * kFacMaxDict + hc]);
      else inc = static_cast<long long>(cj[cell]);
      global_add(g, j + 1, gs, static_cast<unsigned long long>(inc), kAccSumI64);
    } else {
      double inc = 0.0;
      if (h >= 0) {
        inc = __longlong_as_double(static_cast<long long>(hcoef[j * kFacMaxDict + hc]));
      } else {
        inc = __longlong_as_double(static_cast<long long>(cj[cell]));
        if (a.ncar > 0 && a.car_int[0] == 0) {
          inc += __longlong_as_double(static_cast<long long>(cj[static_cast<size_t>(1) * cells + cell])) * __longlong_as_double(static_cast<long long>(car0));
        }
        if (a.ncar > 1 && a.car_int[1] == 0) {
          inc += __longlong_as_double(static_cast<long long>(cj[static_cast<size_t>(2) * cells + cell])) * __longlong_as_double(static_cast<long long>(car1));
        }
      }
      global_add(g, j + 1, gs, static_cast<unsigned long long>(__double_as_longlong(inc)), kAccSumF64);
    }
  }
}

// (the table is private to the workgroup: any hash will do, so a cheap one — one 32-bit multiply)
__device__ __forceinline__ int factored_home_slot(unsigned long long code, int S) {
  return static_cast<int>(((static_cast<uint32_t>(code) ^ static_cast<uint32_t>(code >> 32)) * 0x9E3779B9u) >> 16) & (S - 1);
}
// The group's slot in the workgroup's table (bounded linear probing); -1: the sentinel code, or the table is full.
__device__ __forceinline__ int factored_group_slot(unsigned long long *l_keys, int S, unsigned long long code) {
  if (code == kEmptyCode) return -1;
  int s = factored_home_slot(code, S);
  for (int probes = 0; probes < S; ++probes) {
    unsigned long long k = l_keys[s];
    if (k == kEmptyCode) k = atomicCAS(&l_keys[s], kEmptyCode, code);
    if (k == kEmptyCode || k == code) return s;
    s = (s + 1) & (S - 1);
  }
  return -1;
}

template <bool kFilter>
__global__ __launch_bounds__(kABlock) void agg_factored_kernel(FactoredArgs a, int64_t n, const uint64_t *__restrict__ filter, HashTableView g) {
  extern __shared__ __align__(16) char lds[];
  const int S = a.S, cells = a.cells;
  unsigned long long *l_keys = reinterpret_cast<unsigned long long *>(lds);
  unsigned long long *l_plane = l_keys + S;                                          // [ncar][S * cells]
  unsigned int *l_cnt = reinterpret_cast<unsigned int *>(l_plane + static_cast<size_t>(a.ncar) * S * cells);   // [S * cells]
  unsigned int *l_hist = l_cnt + static_cast<size_t>(S) * cells;                     // [S * hist_words]
  const size_t table_bytes = (static_cast<size_t>(S) * 8 + static_cast<size_t>(a.ncar) * S * cells * 8 + static_cast<size_t>(S) * cells * 4 +
                              static_cast<size_t>(S) * a.hist_words * 4 + 15) & ~static_cast<size_t>(15);
  char *tile = lds + table_bytes;
  for (int i = threadIdx.x; i < S; i += kABlock) l_keys[i] = kEmptyCode;
  for (int i = threadIdx.x; i < a.ncar * S * cells; i += kABlock) l_plane[i] = 0;
  for (int i = threadIdx.x; i < S * cells + S * a.hist_words; i += kABlock) l_cnt[i] = 0;   // (counts and histograms are contiguous)
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int64_t num_tiles = (n + kFacTileRows - 1) / kFacTileRows;
  // The plan's descriptors, read ONCE: indexed from the kernel argument inside the row loop they are chains of dependent
  // scalar loads per row (the first version of this kernel: 6.6 ms per 600 M rows of Q1 against 1.5 for the hand-written
  // prototype).  Fixed-size locals under unrolled loops stay in scalar registers (or their spill lanes).
  int k_off[QSX_MAX_KEYS], k_w[QSX_MAX_KEYS], k_sh[QSX_MAX_KEYS];
#pragma unroll
  for (int k = 0; k < QSX_MAX_KEYS; ++k) {
    const int q = k < a.nkeys ? a.key_slot[k] : 0;
    k_off[k] = a.off[q];
    k_w[k] = a.width[q];
    k_sh[k] = a.key_shift[k];
  }
  int c_off[kFacMaxCell], c_w[kFacMaxCell], c_stride[kFacMaxCell], c_radix[kFacMaxCell];
#pragma unroll
  for (int q = 0; q < kFacMaxCell; ++q) {
    const int sl = q < a.ncell ? a.cell_slot[q] : 0;
    c_off[q] = a.off[sl];
    c_w[q] = a.width[sl];
    c_stride[q] = a.cell_stride[q];
    c_radix[q] = a.cell_radix[q];
  }
  int h_off[kFacMaxHist], h_w[kFacMaxHist], h_at[kFacMaxHist], h_size[kFacMaxHist];
#pragma unroll
  for (int h = 0; h < kFacMaxHist; ++h) {
    const int sl = h < a.nhist ? a.hist_slot[h] : 0;
    h_off[h] = a.off[sl];
    h_w[h] = a.width[sl];
    h_at[h] = a.hist_off[h];
    h_size[h] = a.hist_size[h];
  }
  int r_off[kFacMaxCarriers], r_w[kFacMaxCarriers], r_type[kFacMaxCarriers], r_int[kFacMaxCarriers];
#pragma unroll
  for (int k = 0; k < kFacMaxCarriers; ++k) {
    const int sl = k < a.ncar ? a.car_slot[k] : 0;
    r_off[k] = a.off[sl];
    r_w[k] = a.width[sl];
    r_type[k] = a.car_type[k];
    r_int[k] = a.car_int[k];
  }
  const int nkeys = a.nkeys, ncell = a.ncell, nhist = a.nhist, ncar = a.ncar, hist_words = a.hist_words, filter_off = a.filter_off;
  for (int64_t t = blockIdx.x; t < num_tiles; t += gridDim.x) {
    __syncthreads();   // the previous tile has been read (first pass: the tables are initialised)
    const int64_t row0 = t * kFacTileRows;
    const int rows = static_cast<int>(n - row0 < kFacTileRows ? n - row0 : kFacTileRows);
    for (int q = 0; q < a.nstaged; ++q) {
      const int w = a.width[q];
      const char *src = static_cast<const char *>(a.col[q]) + row0 * w;
      char *dst = tile + a.off[q];
      const int bytes = rows * w;
      if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int full = bytes & ~15;
        const int chunks = (full + 1023) >> 10;
        for (int k = wave; k < chunks; k += kABlock / kWave) {
          const int o = (k << 10) + (lane << 4);
          if (o < full) dma16(src + o, dst + (k << 10));
        }
        if (full != bytes) copy_elements_to_lds<kABlock>(src + full, dst + full, (bytes - full) / w, w);
      } else {
        copy_elements_to_lds<kABlock>(src, dst, rows, w);
      }
    }
    if (kFilter) copy_elements_to_lds<kABlock>(reinterpret_cast<const char *>(filter + (row0 >> 6)), tile + filter_off, (rows + 63) >> 6, 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // column by column over the thread's V rows: the reads of a column are independent and in flight together
    bool live[kFacV];
    unsigned long long code[kFacV];
    int cell[kFacV], slot[kFacV];
#pragma unroll
    for (int v = 0; v < kFacV; ++v) {
      const int row = v * kABlock + threadIdx.x;
      live[v] = row < rows;
      if (kFilter && live[v]) live[v] = msb_bit(reinterpret_cast<const uint64_t *>(tile + filter_off)[row >> 6], row & 63);
      code[v] = 0;
      cell[v] = 0;
    }
#pragma unroll
    for (int k = 0; k < QSX_MAX_KEYS; ++k) {
      if (k >= nkeys) break;
#pragma unroll
      for (int v = 0; v < kFacV; ++v) code[v] |= factored_read(tile, k_off[k], k_w[k], live[v] ? v * kABlock + threadIdx.x : 0) << k_sh[k];
    }
#pragma unroll
    for (int q = 0; q < kFacMaxCell; ++q) {
      if (q >= ncell) break;
#pragma unroll
      for (int v = 0; v < kFacV; ++v) {
        int cc = static_cast<int>(factored_read(tile, c_off[q], c_w[q], live[v] ? v * kABlock + threadIdx.x : 0));
        cc = cc < c_radix[q] ? cc : c_radix[q] - 1;   // (a code beyond the dictionary is outside the contract: stay inside the table)
        cell[v] += cc * c_stride[q];
      }
    }
#pragma unroll
    for (int v = 0; v < kFacV; ++v) slot[v] = live[v] ? factored_group_slot(l_keys, S, code[v]) : -1;
    unsigned long long car[kFacMaxCarriers][kFacV];
#pragma unroll
    for (int k = 0; k < kFacMaxCarriers; ++k) {
#pragma unroll
      for (int v = 0; v < kFacV; ++v) car[k][v] = 0;
      if (k >= ncar) continue;
#pragma unroll
      for (int v = 0; v < kFacV; ++v) {
        car[k][v] = factored_carrier_word(factored_read(tile, r_off[k], r_w[k], live[v] ? v * kABlock + threadIdx.x : 0), r_type[k], r_int[k] != 0);
      }
    }
    bool any_overflow = false;
#pragma unroll
    for (int v = 0; v < kFacV; ++v) {
      if (!live[v]) continue;
      if (slot[v] < 0) {
        any_overflow = true;
        continue;
      }
      const int at = slot[v] * cells + cell[v];
      atomicAdd(&l_cnt[at], 1u);
#pragma unroll
      for (int k = 0; k < kFacMaxCarriers; ++k) {
        if (k >= ncar) break;
        unsigned long long *p = l_plane + static_cast<size_t>(k) * S * cells + at;
        if (r_int[k] != 0) atomicAdd(p, car[k][v]);
        else unsafeAtomicAdd(reinterpret_cast<double *>(p), __longlong_as_double(static_cast<long long>(car[k][v])));
      }
    }
#pragma unroll
    for (int h = 0; h < kFacMaxHist; ++h) {
      if (h >= nhist) break;
#pragma unroll
      for (int v = 0; v < kFacV; ++v) {
        int hc = static_cast<int>(factored_read(tile, h_off[h], h_w[h], live[v] ? v * kABlock + threadIdx.x : 0));
        hc = hc < h_size[h] ? hc : h_size[h] - 1;
        if (live[v] && slot[v] >= 0) atomicAdd(&l_hist[slot[v] * hist_words + h_at[h] + hc], 1u);
      }
    }
    if (__any(any_overflow)) {
      // more groups than the table holds (the optimizer's estimate was far off): those rows' terms go straight to the state
      // (unrolled: live / slot / code / cell / car indexed by a loop variable would live in scratch)
#pragma unroll
      for (int v = 0; v < kFacV; ++v) {
        if (!live[v] || slot[v] >= 0) continue;
        const int row = v * kABlock + threadIdx.x;
        int hist_code[kFacMaxHist] = {};
#pragma unroll
        for (int h = 0; h < kFacMaxHist; ++h) {
          if (h < nhist) hist_code[h] = static_cast<int>(factored_read(tile, h_off[h], h_w[h], row));
        }
        factored_spill_row(a, g, code[v], cell[v], hist_code, car[0][v], car[1][v], a.coef, a.hcoef);
      }
    }
  }
  __syncthreads();
  factored_flush(a, l_keys, l_plane, l_cnt, l_hist, g, a.coef, a.hcoef);
}

// ---- the common signatures without LDS staging ------------------------------------------------------------------------
// Rows reach the lanes by direct loads — a thread owns 8 consecutive rows: 8 bytes of every 1-byte column (the codes of a
// dictionary of <= 64 entries ARE one byte), 32 bytes of an INT key, 64 bytes of a DOUBLE carrier — with the next tile
// requested before the current one is consumed, no barrier and no wait for a DMA in the tile loop: what the prototype
// (tools/ubench/q1_factored.hip) measured at 1.48 ms per 600 M rows of Q1 against 4.5 ms for the staged kernel above, whose
// waves sit out every tile's copy.  Instantiated for the signatures that cover star-schema aggregations over dictionary
// columns: one or two keys of one width (CHAR(1) or INT), one or two cell columns, at most one histogram column, at most
// one DOUBLE carrier; everything else keeps the other kernels.
template <int KEYW, int NK, int NC, int NH, bool kCar>
struct FactoredDirectTile {
  unsigned long long key8[KEYW == 1 ? NK : 1];   // 1-byte keys: 8 rows in 8 bytes
  uint4 key32[KEYW == 4 ? NK : 1][2];            // 4-byte keys: 8 rows in 32 bytes
  unsigned long long cell[NC];
  unsigned long long hist[NH > 0 ? NH : 1];
  uint4 car[kCar ? 4 : 1];
  unsigned int live;                         // bit r: row r of the thread is inside the stripe and selected by the filter
};
// kRuns: the rows are a run of blocks (FactoredRunArgs); d's stripe pointers, `n` and `filter` then change from block to block
// (kFilter says whether ANY block has a filter; a block's own pointer may be null).
// (four waves per SIMD = four workgroups per CU: 128 registers; the single-stripe forms fit by themselves, the run forms are held to it)
template <bool kFilter, int KEYW, int NK, int NC, int NH, bool kCar, bool kRuns = false>
__global__ __launch_bounds__(kABlock, 4) void agg_factored_direct_kernel(const FactoredArgs *__restrict__ a_dev, FactoredDirectArgs d, int64_t n,
                                                                     const uint64_t *__restrict__ filter, HashTableView g, FactoredRunArgs runs) {
  static_assert((KEYW == 1 || KEYW == 4) && NK >= 1 && NK <= 2 && NC >= 1 && NC <= 2 && NH >= 0 && NH <= 1, "instantiated signatures");
  extern __shared__ __align__(16) char lds[];
  const int S = d.S, cells = d.cells;
  unsigned long long *l_keys = reinterpret_cast<unsigned long long *>(lds);
  unsigned long long *l_plane = l_keys + S;                                        // [kCar][S * cells]
  unsigned int *l_cnt = reinterpret_cast<unsigned int *>(l_plane + (kCar ? static_cast<size_t>(S) * cells : 0));
  unsigned int *l_hist = l_cnt + static_cast<size_t>(S) * cells;
  for (int i = threadIdx.x; i < S; i += kABlock) l_keys[i] = kEmptyCode;
  auto clear_cells = [&]() {
    if (kCar) for (int i = threadIdx.x; i < S * cells; i += kABlock) l_plane[i] = 0;
    for (int i = threadIdx.x; i < S * cells + S * d.hist_words; i += kABlock) l_cnt[i] = 0;
  };
  clear_cells();
  __syncthreads();
  const unsigned long long *coef = kRuns ? runs.coef : a_dev->coef, *hcoef = kRuns ? runs.hcoef : a_dev->hcoef;
  using Tile = FactoredDirectTile<KEYW, NK, NC, NH, kCar>;
  // (plain loads, not the non-temporal ones of the scans: a thread owns 8 CONSECUTIVE rows, so a 128-byte line of the DOUBLE
  // carrier is read by four instructions of the wave (two lanes each) — under the non-temporal hint the line did not outlive
  // the first of them and came back from L2 / HBM for the others: tools/ubench/q1_factored.hip 1.63 ms with the hint, 1.33 ms
  // without it, and 1.56 ms with or without any LDS atomic at all: the load path was the whole bound)
  auto load16 = [](const void *p) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = load_global(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  };
  auto request = [&](int64_t tile, Tile &x) {
    const int64_t row = tile * kFacDirectTile + static_cast<int64_t>(threadIdx.x) * kFacDirectRows;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      if constexpr (KEYW == 1) {
        x.key8[k] = load_global(reinterpret_cast<const unsigned long long *>(static_cast<const unsigned char *>(d.key[k]) + row));
      } else {
        x.key32[k][0] = load16(static_cast<const uint32_t *>(d.key[k]) + row);
        x.key32[k][1] = load16(static_cast<const uint32_t *>(d.key[k]) + row + 4);
      }
    }
#pragma unroll
    for (int q = 0; q < NC; ++q) x.cell[q] = load_global(reinterpret_cast<const unsigned long long *>(d.cellc[q] + row));
    if constexpr (NH > 0) x.hist[0] = load_global(reinterpret_cast<const unsigned long long *>(d.histc + row));
    if constexpr (kCar) {
#pragma unroll
      for (int j = 0; j < 4; ++j) x.car[j] = load16(d.carrier + row + 2 * j);
    }
    x.live = 0xFFu;
    if (kFilter && (!kRuns || filter != nullptr)) {   // rows row .. row + 7 sit in one word (row is a multiple of 8): bit 63 - (row & 63) is the first of them
      const uint64_t w = load_global(&filter[row >> 6]);
      x.live = __brev(static_cast<unsigned int>((w >> (56 - (row & 63))) & 0xFFu)) >> 24;   // MSB-first -> bit r = row + r
    }
  };
  auto byte_of = [](unsigned long long v, int r) -> unsigned int { return static_cast<unsigned int>(v >> (r * 8)) & 0xFFu; };
  // A tile's rows in three sweeps, so that the eight group lookups of a thread are in flight together (row by row — hash,
  // ds_read, compare, atomics, next row — every row waited out its own LDS round trip: 1.61 ms per 600 M rows of Q1):
  // (1) key code, cell, histogram code, carrier of every row from the registers; (2) the home slot of every code read from
  // the table, compared; a code that is not at home yet (first sight, or a collision) goes through the probing insert;
  // (3) the atomics.
  auto consume = [&](const Tile &x) {
    unsigned long long code[kFacDirectRows];
    int cell[kFacDirectRows], hcode[kFacDirectRows], home[kFacDirectRows], slot[kFacDirectRows];
#pragma unroll
    for (int r = 0; r < kFacDirectRows; ++r) {
      code[r] = 0;
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        unsigned long long kv;
        if constexpr (KEYW == 1) {
          kv = byte_of(x.key8[k], r);
        } else {
          const uint4 &w = x.key32[k][r >> 2];
          kv = (r & 3) == 0 ? w.x : ((r & 3) == 1 ? w.y : ((r & 3) == 2 ? w.z : w.w));
        }
        code[r] |= kv << d.key_shift[k];
      }
      cell[r] = 0;
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        int cc = static_cast<int>(byte_of(x.cell[q], r));
        cc = cc < d.cell_radix[q] ? cc : d.cell_radix[q] - 1;
        cell[r] += cc * d.cell_stride[q];
      }
      hcode[r] = 0;
      if constexpr (NH > 0) {
        hcode[r] = static_cast<int>(byte_of(x.hist[0], r));
        hcode[r] = hcode[r] < d.hist_size ? hcode[r] : d.hist_size - 1;
      }
      home[r] = factored_home_slot(code[r], S);
    }
    unsigned long long at_home[kFacDirectRows];
#pragma unroll
    for (int r = 0; r < kFacDirectRows; ++r) at_home[r] = l_keys[home[r]];
#pragma unroll
    for (int r = 0; r < kFacDirectRows; ++r) {
      const bool live = ((x.live >> r) & 1u) != 0u;
      slot[r] = !live ? -2 : ((at_home[r] == code[r] && code[r] != kEmptyCode) ? home[r] : factored_group_slot(l_keys, S, code[r]));
    }
#pragma unroll
    for (int r = 0; r < kFacDirectRows; ++r) {
      if (slot[r] == -2) continue;
      unsigned long long carw = 0;
      if constexpr (kCar) {
        const uint4 &pw = x.car[r >> 1];
        carw = (static_cast<unsigned long long>((r & 1) ? pw.w : pw.y) << 32) | ((r & 1) ? pw.z : pw.x);
      }
      if (slot[r] >= 0) {
        const int at = slot[r] * cells + cell[r];
        atomicAdd(&l_cnt[at], 1u);
        if constexpr (kCar) unsafeAtomicAdd(reinterpret_cast<double *>(l_plane) + at, __longlong_as_double(static_cast<long long>(carw)));
        if constexpr (NH > 0) atomicAdd(&l_hist[slot[r] * d.hist_words + hcode[r]], 1u);
      } else {
        const int hist_code[kFacMaxHist] = {hcode[r], 0, 0, 0};
        factored_spill_row(*a_dev, g, code[r], cell[r], hist_code, carw, 0ull, coef, hcoef);
      }
    }
  };
  // rows [first, n) of the stripe, one per thread and step (the tail behind the stripe's last full tile)
  auto row_by_row = [&](int64_t first) {
    for (int64_t row = first + threadIdx.x; row < n; row += kABlock) {
      if (kFilter && (!kRuns || filter != nullptr) && !msb_bit(load_global(&filter[row >> 6]), static_cast<int>(row & 63))) continue;
      unsigned long long code = 0;
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const unsigned long long kv = KEYW == 1 ? static_cast<unsigned long long>(load_global(static_cast<const unsigned char *>(d.key[k]) + row))
                                                : static_cast<unsigned long long>(load_global(static_cast<const uint32_t *>(d.key[k]) + row));
        code |= kv << d.key_shift[k];
      }
      int cell = 0;
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        int cc = load_global(d.cellc[q] + row);
        cc = cc < d.cell_radix[q] ? cc : d.cell_radix[q] - 1;
        cell += cc * d.cell_stride[q];
      }
      int hc = 0;
      if constexpr (NH > 0) {
        hc = load_global(d.histc + row);
        hc = hc < d.hist_size ? hc : d.hist_size - 1;
      }
      unsigned long long carw = 0;
      if constexpr (kCar) carw = static_cast<unsigned long long>(__double_as_longlong(load_global(d.carrier + row)));
      const int slot = factored_group_slot(l_keys, S, code);
      if (slot >= 0) {
        const int at = slot * cells + cell;
        atomicAdd(&l_cnt[at], 1u);
        if constexpr (kCar) unsafeAtomicAdd(reinterpret_cast<double *>(l_plane) + at, __longlong_as_double(static_cast<long long>(carw)));
        if constexpr (NH > 0) atomicAdd(&l_hist[slot * d.hist_words + hc], 1u);
      } else {
        const int hist_code[kFacMaxHist] = {hc, 0, 0, 0};
        factored_spill_row(*a_dev, g, code, cell, hist_code, carw, 0ull, coef, hcoef);
      }
    }
  };
  // tiles lo, lo + step, ... below hi of the current stripe, the next one requested before the current one is consumed
  auto full_tiles_of = [&](int64_t lo, int64_t hi, int64_t step) {
    Tile cur, nxt;
    int64_t tile = lo;
    if (tile < hi) request(tile, cur);
    for (; tile < hi; tile += step) {
      if (tile + step < hi) request(tile + step, nxt);
      consume(cur);
      cur = nxt;
    }
  };
  if constexpr (!kRuns) {
    const int64_t full_tiles = n / kFacDirectTile;          // the tail (< one tile) goes row by row
    full_tiles_of(blockIdx.x, full_tiles, gridDim.x);
    if (static_cast<int64_t>(blockIdx.x) == full_tiles % gridDim.x) row_by_row(full_tiles * kFacDirectTile);   // by the workgroup that would own that tile
    __syncthreads();
    factored_flush(*a_dev, l_keys, l_plane, l_cnt, l_hist, g, coef, hcoef);
  } else {
    // this workgroup's share of the run: tiles [t0, t1) of the run's tile sequence, block by block
    const BlockRunView run = block_run_view(runs.run, 1024);
    int64_t t0 = runs.total_tiles * blockIdx.x / gridDim.x;
    const int64_t t1 = runs.total_tiles * (blockIdx.x + 1) / gridDim.x;
    long long b = 0;
    {
      long long lo = 0, hi = runs.num_blocks;   // first_tile[lo] <= t0 < first_tile[hi]
      while (hi - lo > 1) {
        const long long mid = (lo + hi) >> 1;
        if (runs.first_tile[mid] <= t0) lo = mid; else hi = mid;
      }
      b = static_cast<long long>(wave_broadcast_first(static_cast<uint64_t>(lo)));
    }
    for (; t0 < t1; ++b) {
      const int64_t block_first = static_cast<int64_t>(wave_broadcast_first(static_cast<uint64_t>(runs.first_tile[b]))),
                    block_end = static_cast<int64_t>(wave_broadcast_first(static_cast<uint64_t>(runs.first_tile[b + 1])));
      if (block_end <= t0) continue;   // (a block without tiles)
      // (the block's stripes, filter, row count and coefficient tables are the same for every lane: in scalar registers — as
      // loaded they sit in 2 vector registers each, 162 instead of 125 registers and a wave less per SIMD)
      auto uniform = [](const void *p) { return reinterpret_cast<const void *>(static_cast<uintptr_t>(wave_broadcast_first(reinterpret_cast<uintptr_t>(p)))); };
      const void *const *cols = run.cols + b * QSX_MAX_COLUMNS;
#pragma unroll
      for (int k = 0; k < NK; ++k) d.key[k] = uniform(cols[runs.key_col[k]]);
#pragma unroll
      for (int q = 0; q < NC; ++q) d.cellc[q] = static_cast<const unsigned char *>(uniform(cols[runs.cell_col[q]]));
      if constexpr (NH > 0) d.histc = static_cast<const unsigned char *>(uniform(cols[runs.hist_col]));
      if constexpr (kCar) d.carrier = static_cast<const double *>(uniform(cols[runs.car_col]));
      filter = kFilter && run.filters != nullptr ? static_cast<const uint64_t *>(uniform(run.filters[b])) : nullptr;
      n = static_cast<int64_t>(wave_broadcast_first(static_cast<uint64_t>(run.rows[b])));
      coef = runs.coef + b * runs.coef_words;
      hcoef = runs.hcoef + b * runs.hcoef_words;
      const int64_t lo = t0 - block_first, hi = (t1 < block_end ? t1 : block_end) - block_first;
      const int64_t full_tiles = n / kFacDirectTile;
      full_tiles_of(lo, hi < full_tiles ? hi : full_tiles, 1);
      if (hi > full_tiles) row_by_row(full_tiles * kFacDirectTile);   // the block's tail tile is one of this workgroup's
      __syncthreads();
      factored_flush(*a_dev, l_keys, l_plane, l_cnt, l_hist, g, coef, hcoef);
      __syncthreads();
      clear_cells();       // the group codes stay: the next block's rows find their groups where they are
      __syncthreads();
      t0 = block_first + hi;
    }
  }
}

}  // namespace qsx

#endif  // QSX_CSRC_AGG_FACTORED_KERNELS_HPP_
