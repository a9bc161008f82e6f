#!/usr/bin/env python3
"""Writes the golden fixtures tests/golden/*.json.

The reference (Quickstep, C++) cannot be run in this image, so
these vectors are the reference's OWN known answers, transcribed as data from
its unit tests and SQL golden files: the inputs are regenerated from the data
generators those tests define (closed forms) and the expected outputs are the
values the tests assert / the result tables the .test files print.  Each entry
cites its source (path:line in the Quickstep tree).  Only data is stored here —
no reference source text.

Run:  python tests/golden/make_golden.py     (idempotent; output is committed)
"""
import json
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def dump(name, obj):
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
        f.write("\n")


# --------------------------------------------------------------------------
# 1. hashing / partitioning
# --------------------------------------------------------------------------
def test_table_rows():
    """query_optimizer/tests/TestDatabaseLoader.cpp:118-170 — 25 rows, x = 0..24:
    int_col = (-1)^x * x (NULL when x % 10 == 0), long_col = x^2,
    float_col = sqrt(x), double_col = (-1)^x * x * sqrt(x) (NULL when x % 10 == 0)."""
    rows = []
    sign = 1
    for x in range(25):
        null = x % 10 == 0
        rows.append({
            "int_col": None if null else sign * x,
            "long_col": x * x,
            "float_col": math.sqrt(x),
            "double_col": None if null else sign * math.sqrt(x) * x,
        })
        sign = -sign
    return rows


dump("hash_partition", {
    "source": [
        "utility/HashPair.hpp:47-58 (CombineHashes; value verified by compiling the header, SURVEY.md §8c)",
        "query_optimizer/tests/execution_generator/Partition.test:18-73 (PARTITION BY HASH(id) PARTITIONS 4 listing)",
        "types/TypedValue.hpp:575-592 (identity hash of the zero-extended bit pattern)",
    ],
    "combine_hashes": [{"a": 1, "b": 2, "expected_hex": "86668a560ec835a1"}],
    # printed partition by partition, in partition order 0..3
    "partition_by_hash_4": {
        "ids_inserted": [r["int_col"] for r in test_table_rows() if r["int_col"] is not None],
        "expected_partitions": [
            [4, 8, 12, 16, 24],
            [-3, -7, -11, -15, -19, -23],
            [2, 6, 14, 18, 22],
            [-1, -5, -9, -13, -17, -21],
        ],
    },
    "scalar_hash": [
        {"type": "int", "value": -3, "expected_hex": "00000000fffffffd"},
        {"type": "int", "value": 7, "expected_hex": "0000000000000007"},
        {"type": "long", "value": -3, "expected_hex": "fffffffffffffffd"},
    ],
})

# --------------------------------------------------------------------------
# 2. HashJoinOperator unit test
# --------------------------------------------------------------------------
dump("join_unittest", {
    "source": [
        "relational_operators/tests/HashJoinOperator_unittest.cpp:97-99 (sizes), :196-270 (data), "
        ":379-514 (LongKeyHashJoinTest), :516-690 (IntDuplicateKeyHashJoinTest), "
        ":999-1177 (CompositeKeyHashJoinTest), :1187-1375 (CompositeKeyHashJoinWithResidualPredicateTest)",
    ],
    "num_dim_tuples": 200,
    "num_fact_tuples": 300,
    "block_size": 10,
    # dim (build side) columns as functions of tid: long = tid, int = tid % block_size
    # fact (probe side): long = tid, int = tid
    "long_key": {
        "expected_num_results": 200,
        "expected_count_per_dim_long": 1,           # every dim.long value appears exactly once
    },
    "int_duplicate_key": {
        "expected_num_results": 200,
        "expected_count_per_dim_row": 1,            # every dim row appears exactly once
        "expected_fact_count_first_rows": 20,       # fact rows 0..9 match 200/10 dim rows each
        "expected_fact_count_other_rows": 0,        # fact rows >= 10 never match
    },
    # keys (long, varchar): dim.varchar = tid / 2 * 2, fact.varchar = tid, so only even tids join (:1159-1177)
    "composite_key": {
        "expected_num_results": 100,
        "second_component_dim": "tid // 2 * 2",
        "second_component_fact": "tid",
        "matching_tids": "even tids below num_dim_tuples, each once on both sides",
    },
    # + residual predicate dim.long < 15 (:1268-1272): tids 0, 2, ..., 14 (:1350-1368)
    "composite_key_residual": {
        "residual_dim_long_less_than": 15,
        "expected_num_results": 8,
    },
})

# --------------------------------------------------------------------------
# 3. AggregationOperator unit test
# --------------------------------------------------------------------------
def summation(n):
    return n * (n + 1) // 2


def arithmetic_sum(a, d, n):
    return n * (2 * a + (n - 1) * d) // 2


groupby = {}
for with_pred, repeats in (("without_predicate", 15), ("with_predicate", 15 >> 1)):
    groupby[with_pred] = {
        "num_repeats": repeats,
        "sum_int_per_group": [arithmetic_sum(g, 20, repeats) for g in range(20)],
        "sum_float_per_group": [0.1 * arithmetic_sum(g, 20, repeats) for g in range(20)],
        "avg_int_per_group": [arithmetic_sum(g, 20, repeats) / float(repeats) for g in range(20)],
        "count_per_group": [repeats] * 20,
    }

dump("agg_unittest", {
    "source": [
        "relational_operators/tests/AggregationOperator_unittest.cpp:93-97 (sizes), :192-207 (row generator), "
        ":215-219,:492,:539 (predicates), :571-602,:877-886 (scalar sums), :675-712,:807-864,:959-996 (scalar MIN/MAX), "
        ":1349-1470 (group-by checks), :1550-1680 (group-by MIN/MAX checks), "
        ":585-587 (float tolerance 1e-5 relative)",
    ],
    "num_tuples": 300,
    "group_by_width": 20,
    "group_by_1_size": 4,
    # columns of row val (0..299): GroupBy-0 = (val % 20) % 4, GroupBy-1 = (val % 20) / 4,
    # IntType = LongType = val, FloatType = DoubleType = 0.1 * val
    "scalar": {
        "sum_int_no_predicate": summation(299),                 # 44850, result type LONG
        "sum_double_no_predicate": 0.1 * summation(299),
        "avg_int_no_predicate": summation(299) / 300.0,
        "count_no_predicate": 300,
        "predicate_less_than": 30,
        "sum_int_with_predicate": summation(29),                # 435
        "count_with_predicate": 30,
        "zero_rows_predicate_less_than": -1,                    # SUM/AVG -> NULL, COUNT -> 0 (:1160-1345)
        # MIN / MAX of the attribute (:675-712, :959-996); FloatType/DoubleType hold 0.1 * val
        "max_no_predicate": 299, "max_with_predicate": 29, "min": 0,
        # ... and of the expressions the reference tests build (:807-864): attr + attr, attr * attr
        "max_expr_add_no_predicate": 2 * 299, "max_expr_mul_no_predicate": 299 * 299,
    },
    # GROUP BY MAX / MIN (:1550-1582, :1647-1680): max = 20 * (repeats - 1) + gid, min = gid, with
    # repeats = 15 without and 15 >> 1 with the predicate; float/double columns 0.1 * that
    "group_by_min_max": {
        "without_predicate": {"max_int_per_group": [20 * (15 - 1) + g for g in range(20)],
                              "min_int_per_group": list(range(20))},
        "with_predicate": {"max_int_per_group": [20 * ((15 >> 1) - 1) + g for g in range(20)],
                           "min_int_per_group": list(range(20))},
    },
    "group_by_predicate_less_than": 20 * (15 >> 1),             # IntType < 140
    "group_by": groupby,
    "float_rel_tol": 1e-5,
})

# --------------------------------------------------------------------------
# 4. SQL golden results (execution_generator/*.test)
# --------------------------------------------------------------------------
a_rows = [{"w": i, "x": 10 * i, "y": 100.0 * i} for i in range(20)]
dump("sql_golden", {
    "source": [
        "query_optimizer/tests/execution_generator/LIP.test:20-29,39-75,77-146",
        "query_optimizer/tests/execution_generator/Join.test:17-74, 136-196 (LEFT JOIN chains: NULL join keys)",
        "query_optimizer/tests/execution_generator/Select.test:582-679 (aggregates over the 25-row test table)",
        "query_optimizer/tests/TestDatabaseLoader.cpp:118-170 (test table)",
    ],
    "lip": {
        # R(x, y) = (i, i) for i in 0..100000 step 2 ; S(z) = i for i in 0..100000 step 3
        "r_step": 2, "s_step": 3, "limit": 100000,
        "semi_join_mod_10000": [0, 30000, 60000, 90000],
        "sum_x_union_mod5_mod7": 285685710,
    },
    "join": {
        # a(w, x, y): (i, 10 i, 100 i), i = 0..19
        # b = SELECT w, x + (w/2)%2 FROM a WHERE w % 2 = 0
        # c = SELECT x, y + (x/3)%3 - 1 FROM a WHERE x % 3 = 0
        # d = SELECT y FROM a
        "a": a_rows,
        "three_way_join_expected": [
            {"w": 0, "b_x": 0, "c_y": -1.0},
            {"w": 6, "b_x": 61, "c_y": 601.0},
            {"w": 12, "b_x": 120, "c_y": 1200.0},
            {"w": 18, "b_x": 181, "c_y": 1799.0},
        ],
        # Join.test:136-165: a LEFT JOIN b ON a.w = b.w LEFT JOIN c ON a.x = c.x LEFT JOIN d ON a.y = d.y
        # (d.z = 'C<w>' for every row: d holds every a.y); None = NULL.  Columns a.w, b.x, c.y, row w = index.
        "left_join_on_a": {
            "b_x": [0, None, 21, None, 40, None, 61, None, 80, None, 101, None, 120, None, 141, None, 160, None, 181, None],
            "c_y": [-1.0, None, None, 300.0, None, None, 601.0, None, None, 899.0, None, None, 1200.0, None, None, 1501.0,
                    None, None, 1799.0, None],
        },
        # Join.test:167-196: a LEFT JOIN b ON a.w = b.w LEFT JOIN c ON b.x = c.x LEFT JOIN d ON c.y = d.y — the second
        # and third joins probe with NULL keys (b.x / c.y of the padded rows): a NULL key matches nothing.
        # d_z_w: the w of the matching d row ('C<w>'), None = NULL.
        "left_join_chained": {
            "b_x": [0, None, 21, None, 40, None, 61, None, 80, None, 101, None, 120, None, 141, None, 160, None, 181, None],
            "c_y": [-1.0, None, None, None, None, None, None, None, None, None, None, None, 1200.0, None, None, None,
                    None, None, None, None],
            "d_z_w": [None, None, None, None, None, None, None, None, None, None, None, None, 12, None, None, None,
                      None, None, None, None],
        },
    },
    "test_table": test_table_rows(),
    "select": {
        "count_star": 25,
        # SELECT long_col/100 AS g, COUNT(*), SUM(int_col) ... GROUP BY g HAVING MIN(float_col) > 0
        "group_by_long_div_100": [
            {"g": 1, "count": 5, "sum_int": 2},
            {"g": 2, "count": 3, "sum_int": -16},
            {"g": 3, "count": 2, "sum_int": -1},
            {"g": 4, "count": 3, "sum_int": 1},
            {"g": 5, "count": 2, "sum_int": 1},
        ],
        # SELECT COUNT(*), long_col/100, long_col/50 GROUP BY both HAVING group_col2 > 5
        "group_by_two_keys_gt5": [
            {"count": 1, "g1": 3, "g2": 6}, {"count": 1, "g1": 3, "g2": 7},
            {"count": 2, "g1": 4, "g2": 8}, {"count": 1, "g1": 4, "g2": 9},
            {"count": 1, "g1": 5, "g2": 10}, {"count": 1, "g1": 5, "g2": 11},
        ],
        # Select.test:609-623: SELECT COUNT(*), COUNT(1), COUNT(0), SUM(int_col) / COUNT(*), AVG(int_col+0) * COUNT(1),
        # MAX(double_col+100), MIN(float_col+1) FROM test — int_col and double_col are NULL in rows 0, 10, 20
        "scalar_with_nulls": {
            "count_star": 25, "sum_int_div_count": 0, "avg_int_plus_0_times_count": -20.454545454545457,
            "max_double_plus_100": 217.57550765359252, "min_float_plus_1": 1.0,
        },
        # SELECT int_col FROM test GROUP BY int_col ORDER BY int_col  (NULL group not printed)
        "distinct_int_col": [-23, -21, -19, -17, -15, -13, -11, -9, -7, -5, -3, -1,
                             2, 4, 6, 8, 12, 14, 16, 18, 22, 24],
    },
})

# --------------------------------------------------------------------------
# 5. join table sizing (storage/SimpleScalarSeparateChainingHashTable.hpp:283-397)
# --------------------------------------------------------------------------
dump("bitvector", {
    "source": ["utility/BitVector.hpp:893-935 (bit i = word[i>>6] & (1<<63 >> (i&63)); probe-verified, SURVEY.md §8a a6)"],
    "cases": [
        {"n": 70, "set_bits": [0, 1, 63, 64, 69], "expected_words_hex": ["c000000000000001", "8400000000000000"]},
        {"n": 5, "set_bits": [4], "expected_words_hex": ["0800000000000000"]},
    ],
})
# --------------------------------------------------------------------------
# 6. DATE and CHAR comparisons (types/operations/comparisons/tests/Comparison_unittest.cpp)
# --------------------------------------------------------------------------
# The test compares every pair of its sample values under all six comparisons and expects what the literals' own
# operators / strncmp-then-length give (:547-558, :604-641 dates; :665-742 strings).  Stored: the sample values and, per
# comparison, the expected truth table over all ordered pairs.
_DATES = [[2016, 7, 15], [-18017, 4, 13], [99999, 12, 31], [-99999, 1, 1]]           # :153-156
_SHORT = "foo"                                                                       # :58
_LONG = ("Space is big. You just won't believe how vastly, hugely, mind-bogglingly "
         "big it is. I mean, you may think it's a long way down the road to the "
         "chemist's, but that's just peanuts to space.")                            # :60-63 (the test's sample text, data)
# CHAR values as (text, field width): exact width (no NUL), width + 1, width + 5 (NUL padded)   :240-254, :303-317
_STRINGS = [[_SHORT, len(_SHORT)], [_SHORT, len(_SHORT) + 1], [_SHORT, len(_SHORT) + 5],
            [_LONG, len(_LONG)], [_LONG, len(_LONG) + 1], [_LONG, len(_LONG) + 5]]
_OPS = ["eq", "ne", "lt", "le", "gt", "ge"]


def _cmp_table(values, key):
    def holds(op, a, b):
        return {"eq": a == b, "ne": a != b, "lt": a < b, "le": a <= b, "gt": a > b, "ge": a >= b}[op]
    return {op: [[holds(op, key(a), key(b)) for b in values] for a in values] for op in _OPS}


dump("comparison_unittest", {
    "source": ["types/operations/comparisons/tests/Comparison_unittest.cpp:58-63, 153-156, 240-254, 303-317 (samples), "
               ":547-558, 604-641 (dates: the DateLit operators, types/DatetimeLit.hpp:65-90), "
               ":665-742 (strings: strncmp over the shorter ASCII length, then the lengths)"],
    "dates": _DATES,
    "date_tables": _cmp_table(_DATES, key=lambda d: tuple(d)),
    "strings": _STRINGS,
    # strncmp(min length) then length == ordering of the NUL-trimmed byte strings
    "string_tables": _cmp_table(_STRINGS, key=lambda s: s[0].encode()),
})
print("golden fixtures written to", HERE)
