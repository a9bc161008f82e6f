// agg_shapes.hpp — ahead-of-time plan shapes for the hash-strategy aggregation kernel.
//
// The update kernel (agg_hash_update.hpp) is an interpreter over a qsx_agg_config_t.
// For plan shapes that are known when the library is built, the same kernel body is
// instantiated with the configuration as a compile-time constant: every switch on a
// type / operand kind / opcode folds away and what remains is the straight-line
// arithmetic of that plan (the "prepared statement" of this path).  A state whose
// configuration matches a registered shape uses its specialisation; anything else runs
// the interpreter.  QSX_AGG_NO_SPECIALIZE=1 forces the interpreter (parity tests run both).
//
// Registered shapes = the aggregation plans of the reference's own TPC-H workload
// (benchmarks/tpch/queries/01.sql after ReuseAggregateExpressions) plus
// the "minimal" group-by of BASELINE config 3.
#ifndef QSX_CSRC_AGG_SHAPES_HPP_
#define QSX_CSRC_AGG_SHAPES_HPP_

#include "agg_translate.hpp"

namespace qsx {

// constexpr builder of the public descriptor (mirrors quickstep_amd/types.py:make_agg_config)
struct ConfigBuilder {
  qsx_agg_config_t c{};
  constexpr explicit ConfigBuilder(int strategy) { c.strategy = strategy; }
  constexpr ConfigBuilder &column(int type, int width) {
    c.column_type[c.num_columns] = type;
    c.column_width[c.num_columns] = width;
    ++c.num_columns;
    return *this;
  }
  constexpr ConfigBuilder &key(int column) { c.key_column[c.num_keys++] = column; return *this; }
  constexpr ConfigBuilder &constant(double v) {
    int i = 0;
    while (i < QSX_MAX_CONSTS - 1 && c.consts[i] != 0.0) ++i;
    c.consts[i] = v;
    return *this;
  }
  constexpr ConfigBuilder &instr(int op, int dst, qsx_operand_t a, qsx_operand_t b) {
    c.instrs[c.num_instrs].op = op;
    c.instrs[c.num_instrs].dst = dst;
    c.instrs[c.num_instrs].a = a;
    c.instrs[c.num_instrs].b = b;
    ++c.num_instrs;
    return *this;
  }
  constexpr ConfigBuilder &agg(int fn, qsx_operand_t arg) {
    c.aggs[c.num_aggs].fn = fn;
    c.aggs[c.num_aggs].arg = arg;
    ++c.num_aggs;
    return *this;
  }
  constexpr ConfigBuilder &pred_i64(int column, int op, int64_t literal) {
    c.pred[c.num_pred_terms].column = column;
    c.pred[c.num_pred_terms].op = op;
    c.pred[c.num_pred_terms].literal.i64 = literal;
    ++c.num_pred_terms;
    return *this;
  }
};
constexpr qsx_operand_t Col(int i) { return qsx_operand_t{QSX_OPD_COLUMN, i}; }
constexpr qsx_operand_t Const(int i) { return qsx_operand_t{QSX_OPD_CONST, i}; }
constexpr qsx_operand_t Temp(int i) { return qsx_operand_t{QSX_OPD_TEMP, i}; }

template <typename Shape>
struct ShapeBase {
  static constexpr Translated translated(int tile_rows) {
    Translated t = translate(Shape::config());
    plan_tile(t.dev, t.used_columns, tile_rows, /*has_filter=*/false);
    return t;
  }
};

// TPC-H Q1: GROUP BY l_returnflag, l_linestatus (CHAR(1) x 2); SUM(qty), SUM(price),
// SUM(price*(1-disc)), SUM(price*(1-disc)*(1+tax)), AVG(qty), AVG(price), AVG(disc), COUNT(*).
struct ShapeTpchQ1 : ShapeBase<ShapeTpchQ1> {
  static constexpr qsx_agg_config_t config() {
    ConfigBuilder b(QSX_AGG_COMPACT_KEY);
    b.column(QSX_CHAR, 1).column(QSX_CHAR, 1).column(QSX_DOUBLE, 8).column(QSX_DOUBLE, 8).column(QSX_DOUBLE, 8).column(QSX_DOUBLE, 8);
    b.key(0).key(1).constant(1.0);
    b.instr(QSX_EX_SUB, 0, Const(0), Col(4)).instr(QSX_EX_MUL, 1, Col(3), Temp(0));
    b.instr(QSX_EX_ADD, 2, Const(0), Col(5)).instr(QSX_EX_MUL, 3, Temp(1), Temp(2));
    b.agg(QSX_AGG_SUM, Col(2)).agg(QSX_AGG_SUM, Col(3)).agg(QSX_AGG_SUM, Temp(1)).agg(QSX_AGG_SUM, Temp(3));
    b.agg(QSX_AGG_AVG, Col(2)).agg(QSX_AGG_AVG, Col(3)).agg(QSX_AGG_AVG, Col(4)).agg(QSX_AGG_COUNT_STAR, Col(0));
    return b.c;
  }
};

// BASELINE config 3 "minimal variant": two INT32 keys + one DOUBLE value, SUM / COUNT / AVG.
struct ShapeTwoIntKeysSumCountAvg : ShapeBase<ShapeTwoIntKeysSumCountAvg> {
  static constexpr qsx_agg_config_t config() {
    ConfigBuilder b(QSX_AGG_COMPACT_KEY);
    b.column(QSX_INT, 4).column(QSX_INT, 4).column(QSX_DOUBLE, 8);
    b.key(0).key(1);
    b.agg(QSX_AGG_SUM, Col(2)).agg(QSX_AGG_COUNT_STAR, Col(0)).agg(QSX_AGG_AVG, Col(2));
    return b.c;
  }
};

// Two configurations describe the same plan when everything but the size hints agrees
// (literals of predicate terms included: a shape with a predicate is specific to its literal).
inline bool same_plan(const qsx_agg_config_t &a, const qsx_agg_config_t &b) {
  if (a.strategy != b.strategy || a.num_columns != b.num_columns || a.num_keys != b.num_keys ||
      a.num_instrs != b.num_instrs || a.num_aggs != b.num_aggs || a.num_pred_terms != b.num_pred_terms) {
    return false;
  }
  for (int i = 0; i < a.num_columns; ++i) {
    if (a.column_type[i] != b.column_type[i] || a.column_width[i] != b.column_width[i] ||
        a.column_code_width[i] != b.column_code_width[i] || (a.column_nullable[i] != 0) != (b.column_nullable[i] != 0)) {
      return false;
    }
  }
  for (int i = 0; i < a.num_keys; ++i) {
    if (a.key_column[i] != b.key_column[i]) return false;
  }
  bool const_used[QSX_MAX_CONSTS] = {};
  auto same_operand = [&](const qsx_operand_t &x, const qsx_operand_t &y) {
    if (x.kind == QSX_OPD_CONST && y.kind == QSX_OPD_CONST && x.index == y.index && x.index >= 0 && x.index < QSX_MAX_CONSTS) {
      const_used[x.index] = true;
    }
    return x.kind == y.kind && x.index == y.index;
  };
  for (int i = 0; i < a.num_instrs; ++i) {
    if (a.instrs[i].op != b.instrs[i].op || a.instrs[i].dst != b.instrs[i].dst ||
        !same_operand(a.instrs[i].a, b.instrs[i].a) || !same_operand(a.instrs[i].b, b.instrs[i].b)) {
      return false;
    }
  }
  for (int i = 0; i < QSX_MAX_CONSTS; ++i) {
    if (const_used[i] && std::memcmp(&a.consts[i], &b.consts[i], sizeof(double)) != 0) return false;
  }
  for (int i = 0; i < a.num_aggs; ++i) {
    if (a.aggs[i].fn != b.aggs[i].fn) return false;
    if (a.aggs[i].fn != QSX_AGG_COUNT_STAR && !same_operand(a.aggs[i].arg, b.aggs[i].arg)) return false;
  }
  for (int i = 0; i < a.num_pred_terms; ++i) {
    if (a.pred[i].column != b.pred[i].column || a.pred[i].op != b.pred[i].op ||
        std::memcmp(&a.pred[i].literal, &b.pred[i].literal, sizeof(a.pred[i].literal)) != 0) {
      return false;
    }
  }
  return true;
}

}  // namespace qsx

#endif  // QSX_CSRC_AGG_SHAPES_HPP_
