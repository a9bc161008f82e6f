// agg_factored.hpp — hash aggregation over code stripes with the aggregates FACTORED through the dictionary codes.
//
// Reference: AggregationOperationState::aggregateBlock over a CompressedColumnStore block
// (storage/AggregationOperationState.cpp:428-474; the accessor decodes every value, storage/
// CompressedColumnStoreValueAccessor.hpp:90-150; upsert loop storage/ThreadPrivateCompactKeyHashTable.cpp:216-304;
// layout storage/CompressedColumnStoreTupleStorageSubBlock.cpp:71-160).  The reference's TPC-H DDL stores lineitem that
// way (benchmarks/tpch/create.sql:69-121): l_quantity, l_discount, l_tax are 1-byte dictionary codes of 50 / 11 / 9 values.
//
// What bounds the decoding kernel (DESIGN.md "Aggregation on code stripes"): one LDS atomic per (row, aggregate) — six for
// Q1 — at ~6 ns per wave instruction, 1.0 of its 1.64 ms of compute per 600 M rows; u32 atomics cost what f64 atomics cost
// (tools/ubench/lds_atomic.hip), so "histograms instead of sums" alone buys nothing.  What does: FEWER atomics per row.
// A dictionary column with a handful of values is an extra group-by key in disguise:
//     SUM(price * (1 - disc) * (1 + tax))  =  sum over (d, t) of (1 - d) * (1 + t) * SUM(price | disc = d, tax = t)
// so a row adds its plain ("carrier") values and a 1 to the CELL (group, disc code, tax code) and every aggregate of the
// state is a dot product of the cells with coefficients that depend on the dictionaries only:
//     sum_j(group) = sum over cells c of  A0[j][c] * count(c) + sum over carriers k of Ak[j][c] * carrier_sum_k(c)
//                    (+ sum over codes of H[j][code] * histogram(code) for a dictionary column that only occurs alone)
// Q1 per row: SUM(price) and COUNT into the cell (g, d, t), quantity's code into the group's histogram — three atomics
// instead of six, no dictionary read, no expression.  tools/ubench/q1_factored.hip priced it: 1.48 ms per 600 M rows of
// 13 B (0.66 of 8 TB/s) against 2.1-2.2 ms.
// Exactness: COUNT and integer sums are integer arithmetic on counts; SUM over a dictionary column of integer-valued
// doubles (l_quantity) is a sum of count * value products, exact below 2^53; the other sums regroup the same products
// (relative error ~1e-15: the contract is 1e-6).
//
// Valid for any aggregate argument that is AFFINE in the plain columns once the dictionary columns are fixed (products
// with dictionary-only factors, sums / differences, division by dictionary-only terms) — decided by factored_analyse() on
// the state's expression program; everything else keeps the decoding kernels.
#ifndef QSX_CSRC_AGG_FACTORED_HPP_
#define QSX_CSRC_AGG_FACTORED_HPP_

#include "agg_common.hpp"

namespace qsx {

constexpr int kFacMaxStaged = 12;    // keys + cell columns + histogram columns + carriers
constexpr int kFacMaxCell = 4;
constexpr int kFacMaxHist = 4;
constexpr int kFacMaxCarriers = 2;
constexpr int kFacMaxDict = 64;      // entries of a dictionary column that may become part of a cell / a histogram
constexpr int kFacMaxCells = 4096;   // cells per group
constexpr int kFacTileRows = 1024;
constexpr int kFacV = kFacTileRows / kABlock;

// What the state's plan looks like through the dictionaries (host, once per state).
struct FactoredStatic {
  bool ok = false;
  int ncell = 0, cell_col[kFacMaxCell] = {};
  int nhist = 0, hist_col[kFacMaxHist] = {};
  int ncar = 0, car_col[kFacMaxCarriers] = {}, car_int[kFacMaxCarriers] = {};   // car_int: an i64 plane (SUM(int column))
  int sum_hist[kMaxSums] = {};      // >= 0: the sum's argument depends on that histogram column only
  int sum_car_int[kMaxSums] = {};   // >= 0: SUM of a plain INT / LONG column = that carrier's i64 plane, coefficient 1
  bool prepass = false;             // the state has predicate terms: a pass of its own turns them into the call's filter bitmap
};

// Kernel argument of the accumulate kernel (< 512 bytes).
struct FactoredArgs {
  int nstaged;
  const void *col[kFacMaxStaged];
  int width[kFacMaxStaged];     // bytes per row as staged (code width, or the column's width)
  int off[kFacMaxStaged];       // byte offset inside a tile
  int filter_off, tile_bytes;
  int nkeys, key_slot[QSX_MAX_KEYS], key_shift[QSX_MAX_KEYS];
  int ncell, cell_slot[kFacMaxCell], cell_stride[kFacMaxCell], cell_radix[kFacMaxCell];
  int cells;
  int nhist, hist_slot[kFacMaxHist], hist_off[kFacMaxHist], hist_size[kFacMaxHist];
  int hist_words;               // histogram words per group
  int ncar, car_slot[kFacMaxCarriers], car_type[kFacMaxCarriers], car_int[kFacMaxCarriers];
  int S;                        // group slots of the workgroup's table (power of two)
  int nsums, sum_kind[kMaxSums], sum_hist[kMaxSums], sum_car_int[kMaxSums];
  const unsigned long long *coef;    // [nsums][1 + ncar][cells]: A0, A1, A2 (raw words: double, or int64 for integer sums)
  const unsigned long long *hcoef;   // [nsums][kFacMaxDict]
};

// ---- coefficients (a tiny kernel per call: the dictionaries live in device memory) -----------------------------------
struct FactoredCoefArgs {
  int ncell, cell_col[kFacMaxCell], cell_stride[kFacMaxCell], cell_radix[kFacMaxCell];
  int cells;
  int nhist, hist_col[kFacMaxHist], hist_size[kFacMaxHist];
  int ncar, car_col[kFacMaxCarriers];
  int nsums, sum_hist[kMaxSums];
  unsigned long long *coef, *hcoef;
  // a run of blocks (grid.y = block): the dictionaries of block b at run_dicts[b * QSX_MAX_COLUMNS + column], their entry counts at
  // run_entries[...] (a block's dictionary may be shorter than the radix the cells are laid out for); block b's coefficients at
  // coef + b * coef_words, hcoef + b * hcoef_words.  nullptr: one stripe, the dictionaries of the DevConfig.
  const long long *run_dicts;
  const int *run_entries;
  long long coef_words, hcoef_words;
  int num_blocks;   // (1 for one stripe)
};
constexpr int kFacDirectRows = 8;
constexpr int kFacDirectTile = kABlock * kFacDirectRows;
struct FactoredDirectArgs {
  const void *key[2];
  int key_shift[2];
  const unsigned char *cellc[2];
  int cell_stride[2], cell_radix[2];
  const unsigned char *histc;
  int hist_size;
  const double *carrier;
  int S, cells, hist_words;
};
// A run of blocks through the direct kernel (qsx_agg_update_coded_blocks_sized): every block has its own stripes, filter and
// DICTIONARIES (the reference builds one per block, storage/CompressedBlockBuilder.cpp:300-368), so the cells of a workgroup
// mean something else in every block: a workgroup takes a CONTIGUOUS range of the run's tiles and flushes its cells with the
// block's coefficients whenever it moves on to another block (1-3 flushes per workgroup at 4 workgroups per CU over 2 K blocks).
struct FactoredRunArgs {
  const long long *run;          // the call's block run table (agg_common.hpp BlockRunView: rows, stripes, filters)
  const long long *first_tile;   // [num_blocks + 1]: direct tiles before block b — rows / kFacDirectTile full ones and one for a tail
  long long total_tiles;
  int num_blocks;
  int key_col[2], cell_col[2], hist_col, car_col;   // the state's column behind every stripe the kernel reads
  const unsigned long long *coef, *hcoef;           // block b: coef + b * coef_words, hcoef + b * hcoef_words
  long long coef_words, hcoef_words;
};
// The state's predicate (AggregationOperationState's predicate_: TPC-H Q1's l_shipdate <= DATE, storage/
// AggregationOperationState.cpp:428-440) in front of the factored kernels: one pass over the predicate's columns — values, or
// codes through the block's dictionary — writes the TupleIdSequence the accumulate kernel then takes as its filter (AND the
// caller's own filter).  One workgroup per 1024 rows; a run of blocks: the tiles of the call's run table, bitmap of block b at
// out + 16 * first_tile[b] words.
struct FactoredPredArgs {
  int num_pred;
  DevPred pred[QSX_MAX_PRED_TERMS];
  int type[QSX_MAX_PRED_TERMS];        // the column's type
  int width[QSX_MAX_PRED_TERMS];       // bytes per row of the stripe (the code width of a compressed attribute)
  int coded[QSX_MAX_PRED_TERMS];       // != 0: a compressed attribute (through its dictionary when it has one, else value = code)
  // one stripe
  const void *col[QSX_MAX_PRED_TERMS];
  const void *dict[QSX_MAX_PRED_TERMS];
  const unsigned long long *filter_in;
  long long n;
  // a run of blocks (run != nullptr)
  const long long *run;
  const long long *filters_in;          // [num_blocks] addresses of the caller's filters (0: none), or nullptr
  unsigned long long *out;
};
int launch_factored_predicate(const FactoredPredArgs &pa, long long tiles, hipStream_t s);
// Launchers (agg_factored.hip: the kernels live in a translation unit of their own).
int launch_factored_coef(const DevConfig &dc, const FactoredCoefArgs &ca, hipStream_t s, int num_blocks = 1);
int launch_factored_staged(const FactoredArgs &a, size_t lds_bytes, int per_cu, int64_t n, const uint64_t *filter_dev, const HashTableView &g, hipStream_t s);
// false: the signature is not one of the instantiated ones (nothing was launched)
// Whether launch_factored_direct has a kernel for this plan (asked BEFORE a call pays for its predicate pass and coefficients).
bool factored_direct_signature(const FactoredArgs &a, int key_width);
bool launch_factored_direct(const FactoredArgs &a, const FactoredArgs *a_dev, const FactoredDirectArgs &da, int key_width, size_t lds_bytes, int grid, int64_t n,
                            const uint64_t *filter_dev, const HashTableView &g, hipStream_t s, const FactoredRunArgs *runs = nullptr);

#ifndef __HIPCC_RTC__
// ---- host: is the plan affine in its plain columns once the dictionary columns are fixed? ------------------------------
inline FactoredStatic factored_analyse(const DevConfig &d, bool dense) {
  FactoredStatic f;
  for (int j = 0; j < kMaxSums; ++j) f.sum_hist[j] = f.sum_car_int[j] = -1;
  if (dense || d.wide_words != 0 || d.num_null_cols != 0 || d.num_keys < 1 || d.num_sums < 1) return f;
  unsigned key_mask = 0;
  for (int k = 0; k < d.num_keys; ++k) {
    const int c = d.key_column[k];
    if (d.code_width[c] != 0 || d.column_type[c] == QSX_DATE) return f;   // (keys: plain stripes; a DATE key has padding bytes)
    key_mask |= 1u << c;
  }
  // degree in the plain columns (0 / 1; 2 = not affine), the dictionary columns and the plain columns a node depends on
  struct Node { int deg; unsigned dmask, cmask; };
  auto leaf = [&](const DevOperand &o, const Node (&temps)[QSX_MAX_TEMPS], Node *out) -> bool {
    if (o.kind == QSX_OPD_CONST) { *out = Node{0, 0u, 0u}; return true; }
    if (o.kind == QSX_OPD_TEMP) { *out = temps[o.index]; return true; }
    if (o.kind != QSX_OPD_COLUMN) return false;
    const int c = o.index;
    const int t = d.column_type[c];
    if (t != QSX_INT && t != QSX_LONG && t != QSX_FLOAT && t != QSX_DOUBLE) return false;
    if (d.code_width[c] != 0) *out = Node{0, 1u << c, 0u}; else *out = Node{1, 0u, 1u << c};
    return true;
  };
  Node temps[QSX_MAX_TEMPS] = {};
  unsigned expr_carriers = 0;   // plain columns that occur inside expressions (their planes are f64)
  for (int k = 0; k < d.num_instrs; ++k) {
    Node a{}, b{};
    if (!leaf(d.instrs[k].a, temps, &a) || !leaf(d.instrs[k].b, temps, &b)) return f;
    Node r{0, a.dmask | b.dmask, a.cmask | b.cmask};
    switch (d.instrs[k].op) {
      case QSX_EX_ADD: case QSX_EX_SUB: r.deg = a.deg > b.deg ? a.deg : b.deg; break;
      case QSX_EX_MUL: r.deg = a.deg + b.deg; break;
      default: r.deg = b.deg == 0 ? a.deg : 2; break;
    }
    if (r.deg > 1) r.deg = 2;
    temps[d.instrs[k].dst] = r;
    expr_carriers |= r.cmask;
  }
  unsigned cell_mask = 0, hist_mask = 0, carrier_mask = 0, int_carriers = 0;
  Node of_sum[kMaxSums] = {};
  for (int j = 0; j < d.num_sums; ++j) {
    const DevSum &s = d.sums[j];
    if (s.count_valid != 0 || s.null_mask != 0 || (s.kind != kAccSumF64 && s.kind != kAccSumI64)) return f;
    Node nd{};
    if (!leaf(s.arg, temps, &nd) || nd.deg > 1) return f;
    if (s.kind == kAccSumI64 && s.arg.kind != QSX_OPD_COLUMN && s.arg.kind != QSX_OPD_CONST) return f;
    of_sum[j] = nd;
    carrier_mask |= nd.cmask;
    if (s.kind == kAccSumI64 && nd.cmask != 0) int_carriers |= nd.cmask;
    if (nd.deg == 1 || __builtin_popcount(nd.dmask) >= 2) cell_mask |= nd.dmask;
  }
  if ((int_carriers & expr_carriers) != 0) return f;   // a column summed as an integer AND used in double arithmetic: two planes — not built
  for (int j = 0; j < d.num_sums; ++j) {
    // an integer SUM over a carrier next to a double use of the same column cannot happen (SUM / AVG share an accumulator)
    if (d.sums[j].kind == kAccSumF64 && (of_sum[j].cmask & int_carriers) != 0) return f;
    if (of_sum[j].deg == 0 && __builtin_popcount(of_sum[j].dmask) == 1 && (of_sum[j].dmask & cell_mask) == 0) hist_mask |= of_sum[j].dmask;
  }
  if ((cell_mask | hist_mask) == 0) return f;          // nothing to factor: the decoding kernels are as good
  if (((cell_mask | hist_mask | carrier_mask) & key_mask) != 0) return f;
  for (int c = 0; c < d.num_columns; ++c) {
    if ((cell_mask >> c) & 1u) { if (f.ncell == kFacMaxCell) return f; f.cell_col[f.ncell++] = c; }
    if ((hist_mask >> c) & 1u) { if (f.nhist == kFacMaxHist) return f; f.hist_col[f.nhist++] = c; }
    if ((carrier_mask >> c) & 1u) {
      if (f.ncar == kFacMaxCarriers) return f;
      f.car_int[f.ncar] = (int_carriers >> c) & 1u;
      f.car_col[f.ncar++] = c;
    }
  }
  if (d.num_keys + f.ncell + f.nhist + f.ncar > kFacMaxStaged) return f;
  for (int j = 0; j < d.num_sums; ++j) {
    if (of_sum[j].deg == 0 && (of_sum[j].dmask & hist_mask) != 0) {
      for (int h = 0; h < f.nhist; ++h) if (of_sum[j].dmask == (1u << f.hist_col[h])) f.sum_hist[j] = h;
    }
    if (d.sums[j].kind == kAccSumI64 && of_sum[j].cmask != 0) {
      for (int k = 0; k < f.ncar; ++k) if (of_sum[j].cmask == (1u << f.car_col[k])) f.sum_car_int[j] = k;
    }
  }
  f.prepass = d.num_pred != 0;
  f.ok = true;
  return f;
}
#endif  // __HIPCC_RTC__

}  // namespace qsx

#endif  // QSX_CSRC_AGG_FACTORED_HPP_
