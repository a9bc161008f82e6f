// agg_translate.hpp — qsx_agg_config_t (public descriptor) -> device-side plan.
// constexpr so that the same code runs at state creation (host, run time) and
// inside the ahead-of-time plan-shape specialisations (agg_shapes.hpp, compile time).
#ifndef QSX_CSRC_AGG_TRANSLATE_HPP_
#define QSX_CSRC_AGG_TRANSLATE_HPP_

#include "agg_common.hpp"

namespace qsx {

struct FinalizeDesc {
  int num_aggs;
  int fn[QSX_MAX_AGGS];
  int sum_col[QSX_MAX_AGGS];  // state column of the aggregate's accumulator (>= 1), 0 for COUNT(*)
  int nn_col[QSX_MAX_AGGS];   // state column counting the rows with a non-NULL argument (nullable arguments only), else -1:
                              // the row count stands for it
  int is_int[QSX_MAX_AGGS];   // the accumulator word is a plain int64 (INT/LONG argument)
  int val_type[QSX_MAX_AGGS]; // MIN/MAX: type of the output column (= the argument's; DOUBLE for expressions)
  int num_keys;
  int key_width[QSX_MAX_KEYS];
  int key_shift[QSX_MAX_KEYS];
  int key_type[QSX_MAX_KEYS];
  // wide keys (DevConfig::wide_words): key k sits in word key_word[k]; word w is state column wide_min_col[w], and must
  // equal column wide_max_col[w]
  int wide_words;
  int key_word[QSX_MAX_KEYS];
  int wide_min_col[kMaxKeyWords];
  int wide_max_col[kMaxKeyWords];
  int *collision;   // set by finalize when MIN != MAX somewhere
  void *out_keys[QSX_MAX_KEYS];
  void *out_vals[QSX_MAX_AGGS];
  uint8_t *out_nulls[QSX_MAX_AGGS];
};

struct Translated {
  int status;
  DevConfig dev;     // everything but cols[] and the LDS plan
  FinalizeDesc fin;  // everything but the output pointers
  int num_sums;
  int num_cols;      // state columns in the image
  int col_kind[QSX_MAX_AGGS + 1];  // AccKind of every state column (merge / all-reduce operator)
  unsigned int_col_mask;           // columns that combine as integers (everything but kAccSumF64)
  unsigned used_columns;
  bool dense;
  bool dense_has_count;
};

constexpr int type_width_ce(int type) {
  return (type == QSX_INT || type == QSX_FLOAT) ? 4 : ((type == QSX_LONG || type == QSX_DOUBLE || type == QSX_DATE) ? 8 : 0);
}

constexpr bool valid_operand(const qsx_agg_config_t &c, const qsx_operand_t &o, int defined_temps_mask) {
  switch (o.kind) {
    case QSX_OPD_COLUMN: {
      if (o.index < 0 || o.index >= c.num_columns) return false;
      const int t = c.column_type[o.index];
      return t == QSX_INT || t == QSX_LONG || t == QSX_FLOAT || t == QSX_DOUBLE;
    }
    case QSX_OPD_CONST: return o.index >= 0 && o.index < QSX_MAX_CONSTS;
    case QSX_OPD_TEMP: return o.index >= 0 && o.index < QSX_MAX_TEMPS && ((defined_temps_mask >> o.index) & 1);
    default: return false;
  }
}

constexpr Translated fail(Translated t, int status) {
  t.status = status;
  return t;
}

constexpr Translated translate(const qsx_agg_config_t &c) {
  Translated t{};
  t.status = QSX_OK;
  if (c.num_columns < 0 || c.num_columns > QSX_MAX_COLUMNS || c.num_keys < 0 || c.num_keys > QSX_MAX_KEYS ||
      c.num_aggs < 0 || c.num_aggs > QSX_MAX_AGGS || c.num_instrs < 0 || c.num_instrs > QSX_MAX_INSTRS ||
      c.num_pred_terms < 0 || c.num_pred_terms > QSX_MAX_PRED_TERMS) {
    return fail(t, QSX_ERR_INVALID_ARGUMENT);
  }
  DevConfig &d = t.dev;
  d.num_columns = c.num_columns;
  for (int i = 0; i < c.num_columns; ++i) {
    const int ty = c.column_type[i], w = c.column_width[i];
    if (ty == QSX_CHAR) {
      if (w != 1 && w != 2 && w != 4 && w != 8) return fail(t, QSX_ERR_UNSUPPORTED);
    } else if (type_width_ce(ty) == 0 || w != type_width_ce(ty)) {
      return fail(t, QSX_ERR_INVALID_ARGUMENT);
    }
    d.column_type[i] = ty;
    d.column_width[i] = w;
    const int cw = c.column_code_width[i];
    if (cw != 0) {   // compressed attribute: 1 / 2 / 4-byte codes of a numeric column
      if ((cw != 1 && cw != 2 && cw != 4) || ty == QSX_CHAR) return fail(t, QSX_ERR_UNSUPPORTED);   // (a DATE always has a dictionary)
      d.code_width[i] = cw;
    }
  }
  switch (c.strategy) {
    case QSX_AGG_SINGLE_STATE:
      if (c.num_keys != 0) return fail(t, QSX_ERR_INVALID_ARGUMENT);
      break;
    case QSX_AGG_COLLISION_FREE:
      if (c.num_keys != 1 || c.num_entries <= 0) return fail(t, QSX_ERR_INVALID_ARGUMENT);
      break;
    case QSX_AGG_COMPACT_KEY:
    case QSX_AGG_GENERIC:
      if (c.num_keys < 1) return fail(t, QSX_ERR_INVALID_ARGUMENT);
      break;
    default: return fail(t, QSX_ERR_INVALID_ARGUMENT);
  }
  d.num_keys = c.num_keys;
  int offset_bytes = 0, total_key_bytes = 0;
  for (int k = 0; k < c.num_keys; ++k) {
    const int col = c.key_column[k];
    if (col < 0 || col >= c.num_columns) return fail(t, QSX_ERR_INVALID_ARGUMENT);
    total_key_bytes += c.column_width[col];
  }
  // Both hash strategies pack a key of up to 8 bytes into one 64-bit code (ThreadPrivateCompactKeyHashTable's KeyCode);
  // a wider key is packed into several words and hashed (DevConfig::wide_words).
  const bool wide = total_key_bytes > 8;
  int word_bytes[QSX_MAX_KEYS] = {};   // wide: first fit, a component never straddles two words
  int words = 0;
  for (int k = 0; k < c.num_keys; ++k) {
    const int col = c.key_column[k];
    const int ty = c.column_type[col];
    if (c.strategy == QSX_AGG_COLLISION_FREE && ty != QSX_INT && ty != QSX_LONG) return fail(t, QSX_ERR_UNSUPPORTED);
    if (c.strategy == QSX_AGG_GENERIC && ty == QSX_CHAR) return fail(t, QSX_ERR_UNSUPPORTED);  // FarmHash keys: out of scope
    int word = 0;
    if (wide) {
      while (word < words && word_bytes[word] + c.column_width[col] > 8) ++word;
      if (word == words) ++words;
      offset_bytes = word_bytes[word];
      word_bytes[word] += c.column_width[col];
    }
    d.key_column[k] = col;
    d.key_width[k] = c.column_width[col];
    d.key_shift[k] = offset_bytes * 8;
    d.key_word[k] = word;
    if (!wide) offset_bytes += c.column_width[col];
  }
  d.wide_words = wide ? words : 0;
  d.wide_hash_mask = ~0ull;
  if (wide && (c.strategy == QSX_AGG_COLLISION_FREE || d.wide_words > kMaxKeyWords)) return fail(t, QSX_ERR_UNSUPPORTED);
  // expression program
  int defined = 0;
  d.num_instrs = c.num_instrs;
  for (int k = 0; k < c.num_instrs; ++k) {
    const qsx_expr_instr_t &in = c.instrs[k];
    if (in.op < QSX_EX_ADD || in.op > QSX_EX_DIV || in.dst < 0 || in.dst >= QSX_MAX_TEMPS) return fail(t, QSX_ERR_INVALID_ARGUMENT);
    if (!valid_operand(c, in.a, defined) || !valid_operand(c, in.b, defined)) return fail(t, QSX_ERR_INVALID_ARGUMENT);
    d.instrs[k].op = in.op;
    d.instrs[k].dst = in.dst;
    d.instrs[k].a = DevOperand{in.a.kind, in.a.index};
    d.instrs[k].b = DevOperand{in.b.kind, in.b.index};
    defined |= 1 << in.dst;
  }
  for (int k = 0; k < QSX_MAX_CONSTS; ++k) d.consts[k] = c.consts[k];
  // nullable columns -> null slots; which temps are NULL when which slot is
  unsigned col_null_mask[QSX_MAX_COLUMNS] = {};
  d.num_null_cols = 0;
  d.row_null_mask = 0;
  auto null_slot_mask = [&](int col) constexpr -> unsigned {
    if (c.column_nullable[col] == 0) return 0u;
    if (col_null_mask[col] == 0) {
      d.null_column[d.num_null_cols] = col;
      col_null_mask[col] = 1u << d.num_null_cols++;
    }
    return col_null_mask[col];
  };
  for (int k = 0; k < c.num_keys; ++k) d.row_null_mask |= null_slot_mask(c.key_column[k]);
  for (int p = 0; p < c.num_pred_terms; ++p) {
    if (c.pred[p].column >= 0 && c.pred[p].column < c.num_columns) d.row_null_mask |= null_slot_mask(c.pred[p].column);
  }
  unsigned temp_null_mask[QSX_MAX_TEMPS] = {};
  auto operand_null_mask = [&](const qsx_operand_t &o) constexpr -> unsigned {
    return o.kind == QSX_OPD_COLUMN ? null_slot_mask(o.index) : (o.kind == QSX_OPD_TEMP ? temp_null_mask[o.index] : 0u);
  };
  for (int k = 0; k < c.num_instrs; ++k) {
    temp_null_mask[c.instrs[k].dst] = operand_null_mask(c.instrs[k].a) | operand_null_mask(c.instrs[k].b);
  }
  // aggregates -> state columns
  FinalizeDesc &f = t.fin;
  f.num_aggs = c.num_aggs;
  int ns = 0;
  bool needs_count = false;
  for (int a = 0; a < c.num_aggs; ++a) {
    const qsx_agg_desc_t &ag = c.aggs[a];
    f.fn[a] = ag.fn;
    f.nn_col[a] = -1;
    if (ag.fn == QSX_AGG_COUNT_STAR) {
      needs_count = true;
      f.sum_col[a] = 0;
      continue;
    }
    if (ag.fn != QSX_AGG_SUM && ag.fn != QSX_AGG_AVG && ag.fn != QSX_AGG_MIN && ag.fn != QSX_AGG_MAX && ag.fn != QSX_AGG_COUNT) return fail(t, QSX_ERR_UNSUPPORTED);
    if (ag.arg.kind == QSX_OPD_CONST || !valid_operand(c, ag.arg, defined)) return fail(t, QSX_ERR_INVALID_ARGUMENT);
    const unsigned arg_nulls = operand_null_mask(ag.arg);
    if (arg_nulls != 0) {
      // the rows this aggregate sees = the rows with a non-NULL argument: one counting accumulator per distinct mask
      int j = 0;
      while (j < ns && !(d.sums[j].count_valid != 0 && d.sums[j].null_mask == arg_nulls)) ++j;
      if (j == ns) {
        if (ns == kMaxSums) return fail(t, QSX_ERR_UNSUPPORTED);
        d.sums[ns].arg = DevOperand{QSX_OPD_CONST, 0};
        d.sums[ns].is_int = 1;
        d.sums[ns].kind = kAccSumI64;
        d.sums[ns].null_mask = arg_nulls;
        d.sums[ns].count_valid = 1;
        ++ns;
      }
      f.nn_col[a] = j + 1;
    }
    if (ag.fn == QSX_AGG_COUNT) {
      needs_count = true;
      f.sum_col[a] = 0;   // COUNT(x) over a non-nullable x is the row count; else nn_col
      continue;
    }
    if (ag.fn == QSX_AGG_AVG || ag.fn == QSX_AGG_MIN || ag.fn == QSX_AGG_MAX) needs_count = true;   // NULL over zero rows
    const bool is_int = ag.arg.kind == QSX_OPD_COLUMN &&
                        (c.column_type[ag.arg.index] == QSX_INT || c.column_type[ag.arg.index] == QSX_LONG);
    f.is_int[a] = is_int ? 1 : 0;
    f.val_type[a] = ag.arg.kind == QSX_OPD_COLUMN ? c.column_type[ag.arg.index] : QSX_DOUBLE;
    const int kind = ag.fn == QSX_AGG_MIN ? kAccMinI64 : (ag.fn == QSX_AGG_MAX ? kAccMaxI64 : (is_int ? kAccSumI64 : kAccSumF64));
    // SUM(x) and AVG(x) over the same argument share one accumulator (what
    // ReuseAggregateExpressions does on the optimizer side,
    // query_optimizer/rules/ReuseAggregateExpressions.hpp:43-80); so do repeated MIN(x) / MAX(x).
    int j = 0;
    while (j < ns && !(d.sums[j].count_valid == 0 && d.sums[j].arg.kind == ag.arg.kind && d.sums[j].arg.index == ag.arg.index && d.sums[j].kind == kind)) ++j;
    if (j == ns) {
      if (ns == kMaxSums) return fail(t, QSX_ERR_UNSUPPORTED);
      d.sums[ns].arg = DevOperand{ag.arg.kind, ag.arg.index};
      d.sums[ns].is_int = is_int ? 1 : 0;
      d.sums[ns].kind = kind;
      d.sums[ns].null_mask = arg_nulls;
      d.sums[ns].count_valid = 0;
      ++ns;
    }
    f.sum_col[a] = j + 1;  // fixed up below for dense states without a count column
  }
  // wide keys: MIN and MAX of every key word, behind the aggregates' accumulators
  f.wide_words = d.wide_words;
  for (int w = 0; w < d.wide_words; ++w) {
    if (ns + 2 > kMaxSums) return fail(t, QSX_ERR_UNSUPPORTED);   // (accumulators per state are limited: fewer aggregates or a narrower key)
    for (int which = 0; which < 2; ++which) {
      d.sums[ns].arg = DevOperand{kOpdKeyWord, w};
      d.sums[ns].is_int = 1;
      d.sums[ns].kind = which == 0 ? kAccMinI64 : kAccMaxI64;
      d.sums[ns].null_mask = 0;
      d.sums[ns].count_valid = 0;
      ++ns;
    }
    f.wide_min_col[w] = ns - 1;   // state column = accumulator index + 1
    f.wide_max_col[w] = ns;
  }
  d.num_sums = ns;
  t.num_sums = ns;
  // predicate
  d.num_pred = c.num_pred_terms;
  for (int p = 0; p < c.num_pred_terms; ++p) {
    const qsx_pred_term_t &term = c.pred[p];
    if (term.column < 0 || term.column >= c.num_columns || term.op < QSX_EQ || term.op > QSX_GE) return fail(t, QSX_ERR_INVALID_ARGUMENT);
    d.pred[p].column = term.column;
    d.pred[p].op = term.op;
    unsigned long long bits = 0;
    switch (c.column_type[term.column]) {
      case QSX_INT: bits = static_cast<uint32_t>(term.literal.i32); break;
      case QSX_LONG: bits = static_cast<unsigned long long>(term.literal.i64); break;
      case QSX_FLOAT: bits = __builtin_bit_cast(uint32_t, term.literal.f32); break;
      case QSX_DOUBLE: bits = __builtin_bit_cast(unsigned long long, term.literal.f64); break;
      case QSX_DATE: bits = static_cast<unsigned long long>(term.literal.i64); break;   // the DateLit bytes
      default: return fail(t, QSX_ERR_UNSUPPORTED);
    }
    d.pred[p].literal = bits;
  }
  // finalize key description
  f.num_keys = c.num_keys;
  for (int k = 0; k < c.num_keys; ++k) {
    f.key_width[k] = d.key_width[k];
    f.key_shift[k] = d.key_shift[k];
    f.key_type[k] = c.column_type[c.key_column[k]];
    f.key_word[k] = d.key_word[k];
  }
  // columns the update kernel has to stage: keys, predicate, expression and aggregate operands
  unsigned used = 0;
  for (int k = 0; k < c.num_keys; ++k) used |= 1u << c.key_column[k];
  for (int p = 0; p < c.num_pred_terms; ++p) used |= 1u << c.pred[p].column;
  for (int k = 0; k < c.num_instrs; ++k) {
    if (c.instrs[k].a.kind == QSX_OPD_COLUMN) used |= 1u << c.instrs[k].a.index;
    if (c.instrs[k].b.kind == QSX_OPD_COLUMN) used |= 1u << c.instrs[k].b.index;
  }
  for (int j = 0; j < ns; ++j) {
    if (d.sums[j].count_valid == 0 && d.sums[j].arg.kind == QSX_OPD_COLUMN) used |= 1u << d.sums[j].arg.index;
  }
  t.used_columns = used;
  t.dense = c.strategy == QSX_AGG_COLLISION_FREE;
  t.dense_has_count = needs_count;
  if (t.dense && !needs_count) {
    for (int a = 0; a < c.num_aggs; ++a) {   // no count column in front
      f.sum_col[a] -= 1;
      if (f.nn_col[a] >= 0) f.nn_col[a] -= 1;
    }
  }
  t.num_cols = t.dense ? ns + (needs_count ? 1 : 0) : ns + 1;
  t.int_col_mask = 0;
  {
    int col = 0;
    if (!t.dense || needs_count) {
      t.col_kind[col] = kAccSumI64;
      t.int_col_mask |= 1u << col++;
    }
    for (int j = 0; j < ns; ++j, ++col) {
      t.col_kind[col] = d.sums[j].kind;
      if (d.sums[j].kind != kAccSumF64) t.int_col_mask |= 1u << col;
    }
  }
  return t;
}

constexpr size_t align16_ce(size_t v) { return (v + 15) & ~static_cast<size_t>(15); }

// LDS layout of one tile of `tile_rows` rows: referenced columns back to back (16-byte
// aligned), then the filter words.  Returns the tile size in bytes.
// reg_decode (plan shapes of the hash path): a compressed attribute gets NO value slots in the tile — lds_off = kRegDecoded,
// the thread decodes the codes of its rows into registers (agg_hash_update.hpp DecodedRows) — so the tile is as small as
// the bytes that are read (Q1 over lineitem's codes: 13 instead of 37 KiB per 1024 rows).
constexpr int plan_tile(DevConfig &d, unsigned used_columns, int tile_rows, bool has_filter, bool reg_decode = false) {
  size_t off = 0;
  for (int col = 0; col < QSX_MAX_COLUMNS; ++col) {
    if (col < d.num_columns && ((used_columns >> col) & 1u)) {
      d.code_off[col] = -1;
      if (reg_decode && d.code_width[col] != 0) {
        d.lds_off[col] = kRegDecoded;
        d.code_off[col] = static_cast<int>(off);
        off += align16_ce(static_cast<size_t>(tile_rows) * d.code_width[col]);
        continue;
      }
      d.lds_off[col] = static_cast<int>(off);
      off += align16_ce(static_cast<size_t>(tile_rows) * d.column_width[col]);
      if (d.code_width[col] != 0) {   // compressed attribute: the codes are staged, the slots above receive the values
        d.code_off[col] = static_cast<int>(off);
        off += align16_ce(static_cast<size_t>(tile_rows) * d.code_width[col]);
      }
    } else {
      d.lds_off[col] = -1;
      d.code_off[col] = -1;
    }
  }
  d.filter_lds_off = -1;
  if (has_filter) {
    d.filter_lds_off = static_cast<int>(off);
    off += align16_ce(tile_rows / 64 * 8);
  }
  for (int s = 0; s < QSX_MAX_COLUMNS; ++s) {
    d.null_lds_off[s] = -1;
    if (s < d.num_null_cols) {
      d.null_lds_off[s] = static_cast<int>(off);
      off += align16_ce(tile_rows / 64 * 8);
    }
  }
  if (off == 0) off = 16;
  d.tile_bytes = static_cast<int>(off);
  return d.tile_bytes;
}

}  // namespace qsx

#endif  // QSX_CSRC_AGG_TRANSLATE_HPP_
