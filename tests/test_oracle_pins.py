"""CPU: pin the oracle (oracle/qsx_oracle.cpp) against the reference's own
golden vectors (tests/golden/*.json, see make_golden.py for sources) and
against oracle/_ref/ref_pin, the one piece of the reference that compiles here."""
import os

import numpy as np
import pytest

from quickstep_amd import types as T


def test_combine_hashes_matches_reference_header(oracle, golden):
    for case in golden["hash_partition"]["combine_hashes"]:
        assert oracle.combine_hashes(case["a"], case["b"]) == int(case["expected_hex"], 16)
    if not os.path.exists(oracle.REF_PIN):
        pytest.skip("oracle/_ref/ref_pin not built (reference tree absent)")
    rng = np.random.default_rng(1)
    pairs = [(int(a), int(b)) for a, b in rng.integers(0, 2**63, size=(64, 2), dtype=np.int64)]
    pairs += [(0, 0), (2**64 - 1, 2**64 - 1), (1, 2)]
    ref = oracle.ref_combine_hashes(pairs)
    assert ref == [oracle.combine_hashes(a, b) for a, b in pairs]


def test_scalar_hash_is_zero_extended_bit_pattern(oracle, golden):
    qt = {"int": T.INT, "long": T.LONG}
    for case in golden["hash_partition"]["scalar_hash"]:
        assert oracle.hash_scalar(qt[case["type"]], case["value"]) == int(case["expected_hex"], 16)
    assert oracle.hash_scalar(T.DOUBLE, -0.0) == oracle.hash_scalar(T.DOUBLE, 0.0) == 0
    assert oracle.hash_scalar(T.FLOAT, -0.0) == 0


def test_partition_listing_of_partition_test(oracle, golden):
    g = golden["hash_partition"]["partition_by_hash_4"]
    parts = [[] for _ in range(4)]
    for v in g["ids_inserted"]:
        parts[oracle.partition_id(oracle.hash_scalar(T.INT, v), 4)].append(v)
    assert parts == g["expected_partitions"]
    # the scatter restatement gives the same membership, in insertion order
    keys = np.array(g["ids_inserted"], dtype=np.int32)
    offs = oracle.partition_offsets(keys, 4)
    out = oracle.partition_scatter(keys, 4, keys)
    for p in range(4):
        assert out[offs[p]:offs[p + 1]].tolist() == g["expected_partitions"][p]
    # non power of two: h >= P ? h % P : h
    assert oracle.partition_id(7, 3) == 1 and oracle.partition_id(2, 3) == 2


def test_primes(oracle):
    assert [oracle.next_prime(n) for n in (0, 1, 2, 3, 4, 90, 2000000)] == [2, 2, 2, 3, 5, 97, 2000003]
    assert [oracle.prev_prime(n) for n in (0, 1, 2, 3, 4, 100)] == [0, 0, 2, 3, 3, 97]


def test_bitvector_is_msb_first(oracle, golden):
    for case in golden["bitvector"]["cases"]:
        bools = np.zeros(case["n"], dtype=bool)
        bools[case["set_bits"]] = True
        col = bools.astype(np.int32)
        bm = oracle.select_cmp(col, T.EQ, 1)
        assert [f"{w:016x}" for w in bm] == case["expected_words_hex"]
        assert oracle.bitmap_count(bm, case["n"]) == len(case["set_bits"])
        assert oracle.bitmap_to_tids(bm, case["n"]).tolist() == case["set_bits"]
        assert np.array_equal(oracle.bitmap_from_bools(bools), bm)
        assert np.array_equal(oracle.bools_from_bitmap(bm, case["n"]), bools)


def test_select_filter_short_circuit(oracle):
    rng = np.random.default_rng(3)
    col = rng.integers(-100, 100, size=1000).astype(np.int64)
    first = oracle.select_cmp(col, T.GT, -20)
    both = oracle.select_cmp(col, T.LT, 30, filter_bitmap=first)
    want = (col > -20) & (col < 30)
    assert np.array_equal(oracle.bools_from_bitmap(both, col.size), want)
    assert np.array_equal(oracle.compact_gather(col, both), col[want])


def test_join_table_sizing_follows_reference_formula(oracle):
    # storage/SimpleScalarSeparateChainingHashTable.hpp:283-397
    info = oracle.JoinTable(T.INT, 1_000_000).info()
    assert info["blob_bytes"] == 23 * 2 * 1024 * 1024          # ceil((128 + 24*2000003) / 2 MiB) slots
    avail = info["blob_bytes"] - 128
    buckets = avail // 48
    assert info["num_slots"] == oracle.prev_prime(buckets * 2)
    assert info["num_buckets"] == info["num_slots"] // 2
    small = oracle.JoinTable(T.LONG, 10).info()
    assert small["blob_bytes"] == 2 * 1024 * 1024 and small["num_slots"] == oracle.prev_prime(2 * ((2 * 1024 * 1024 - 128) // 48))


def _join_unittest_tables(g):
    dim_tid = np.arange(g["num_dim_tuples"])
    fact_tid = np.arange(g["num_fact_tuples"])
    return dim_tid, fact_tid


def test_join_unittest_long_key(oracle, golden):
    g = golden["join_unittest"]
    dim_tid, fact_tid = _join_unittest_tables(g)
    table = oracle.JoinTable(T.LONG, g["num_dim_tuples"])
    bs = g["block_size"]
    for b in range(0, g["num_dim_tuples"], bs):       # one BuildHashWorkOrder per 10-row block
        table.build(dim_tid[b:b + bs].astype(np.int64), block_id=b // bs, base_tid=b)
    counts = np.zeros(g["num_dim_tuples"], dtype=np.int64)
    total = 0
    for b in range(0, g["num_fact_tuples"], bs):      # one HashInnerJoinWorkOrder per probe block
        p, d = table.probe(fact_tid[b:b + bs].astype(np.int64), probe_base_tid=b)
        total += p.size
        np.add.at(counts, dim_tid[d], 1)              # projected dim.long == dim tid
        assert np.array_equal(dim_tid[d], fact_tid[p])
    assert total == g["long_key"]["expected_num_results"]
    assert (counts == g["long_key"]["expected_count_per_dim_long"]).all()


def test_join_unittest_int_duplicate_key(oracle, golden):
    g = golden["join_unittest"]
    dim_tid, fact_tid = _join_unittest_tables(g)
    bs = g["block_size"]
    dim_int = (dim_tid % bs).astype(np.int32)
    fact_int = fact_tid.astype(np.int32)
    table = oracle.JoinTable(T.INT, g["num_dim_tuples"])
    for b in range(0, g["num_dim_tuples"], bs):
        table.build(dim_int[b:b + bs], block_id=b // bs, base_tid=b)
    p, d, blocks = table.probe(fact_int, with_blocks=True)
    e = g["int_duplicate_key"]
    assert p.size == e["expected_num_results"]
    assert (np.bincount(d, minlength=g["num_dim_tuples"]) == e["expected_count_per_dim_row"]).all()
    fact_counts = np.bincount(p, minlength=g["num_fact_tuples"])
    assert (fact_counts[:bs] == e["expected_fact_count_first_rows"]).all()
    assert (fact_counts[bs:] == e["expected_fact_count_other_rows"]).all()
    assert np.array_equal(blocks, (d // bs).astype(np.uint64))   # TupleReference.block of each match
    # chain order = insertion order (tail append, SimpleScalarSeparateChainingHashTable.hpp:1062-1113)
    assert np.array_equal(d[p == 3], np.arange(3, 200, 10))


def test_join_table_resize_keeps_all_entries(oracle):
    table = oracle.JoinTable(T.INT, 4)                     # tiny estimate: forces resize() under load
    rng = np.random.default_rng(11)
    keys = rng.integers(0, 50_000, size=200_000).astype(np.int32)
    for b in range(0, keys.size, 50_000):
        table.build(keys[b:b + 50_000], block_id=b // 50_000, base_tid=b)
    assert table.info()["buckets_allocated"] == keys.size
    probe = np.arange(0, 50_000, 7, dtype=np.int32)
    p, d = table.probe(probe)
    assert p.size == sum(np.count_nonzero(keys == k) for k in probe[:50]) + \
        np.isin(keys, probe[50:]).sum()
    assert np.array_equal(keys[d], probe[p])


def _agg_rows(g):
    val = np.arange(g["num_tuples"])
    gid = val % g["group_by_width"]
    return dict(gb0=(gid % g["group_by_1_size"]).astype(np.int32), gb1=(gid // g["group_by_1_size"]).astype(np.int32),
                i=val.astype(np.int32), l=val.astype(np.int64), f=(0.1 * val).astype(np.float32), d=0.1 * val)


def test_agg_unittest_scalar(oracle, golden):
    g = golden["agg_unittest"]
    rows = _agg_rows(g)
    s = g["scalar"]
    cols = [rows["i"], rows["l"], rows["f"], rows["d"]]
    layout = [(T.INT, None), (T.LONG, None), (T.FLOAT, None), (T.DOUBLE, None)]
    aggs = [(T.AGG_SUM, T.col(0)), (T.AGG_SUM, T.col(1)), (T.AGG_SUM, T.col(3)), (T.AGG_AVG, T.col(0)),
            (T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(2))]
    for pred, exp_sum, exp_cnt in ((None, s["sum_int_no_predicate"], s["count_no_predicate"]),
                                   (s["predicate_less_than"], s["sum_int_with_predicate"], s["count_with_predicate"])):
        cfg = T.make_agg_config(T.AGG_SINGLE_STATE, layout, aggs=aggs,
                                pred=[] if pred is None else [(0, T.LT, pred)])
        st = oracle.AggState(cfg)
        for b in range(0, g["num_tuples"], 10):          # 10 tuples per block (:107)
            st.update([c[b:b + 10] for c in cols], 10)
        _, vals, nulls = st.finalize()
        assert vals[0][0] == exp_sum and vals[1][0] == exp_sum and vals[4][0] == exp_cnt
        assert vals[0].dtype == np.int64                  # SUM(INT) is LONG (:593-602)
        assert vals[2][0] == pytest.approx(0.1 * exp_sum, rel=g["float_rel_tol"])
        assert vals[5][0] == pytest.approx(0.1 * exp_sum, rel=g["float_rel_tol"])
        assert vals[3][0] == pytest.approx(exp_sum / exp_cnt, rel=1e-12)
        assert not any(z[0] for z in nulls)
    # zero rows: SUM / AVG NULL, COUNT 0, still exactly one row (:1160-1345)
    cfg = T.make_agg_config(T.AGG_SINGLE_STATE, layout, aggs=aggs, pred=[(0, T.LT, s["zero_rows_predicate_less_than"])])
    st = oracle.AggState(cfg)
    st.update(cols)
    keys, vals, nulls = st.finalize()
    assert vals[4].tolist() == [0] and [int(z[0]) for z in nulls] == [1, 1, 1, 1, 0, 1]


@pytest.mark.parametrize("strategy", [T.AGG_COMPACT_KEY, T.AGG_GENERIC])
@pytest.mark.parametrize("with_pred", [False, True])
def test_agg_unittest_group_by(oracle, golden, strategy, with_pred):
    g = golden["agg_unittest"]
    rows = _agg_rows(g)
    e = g["group_by"]["with_predicate" if with_pred else "without_predicate"]
    cols = [rows["gb0"], rows["gb1"], rows["i"], rows["d"], rows["f"]]
    cfg = T.make_agg_config(
        strategy, [(T.INT, None), (T.INT, None), (T.INT, None), (T.DOUBLE, None), (T.FLOAT, None)], keys=[0, 1],
        aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.col(3)), (T.AGG_AVG, T.col(2)), (T.AGG_COUNT_STAR, None),
              (T.AGG_SUM, T.col(4))],
        pred=[(2, T.LT, g["group_by_predicate_less_than"])] if with_pred else [], est_groups=20)
    st = oracle.AggState(cfg)
    for b in range(0, g["num_tuples"], 10):
        st.update([c[b:b + 10] for c in cols], 10)
    keys, vals, _ = st.finalize()
    assert keys[0].size == g["group_by_width"]
    gid = keys[0] + keys[1] * g["group_by_1_size"]       # :517
    order = np.argsort(gid)
    assert gid[order].tolist() == list(range(20))
    assert vals[0][order].tolist() == e["sum_int_per_group"]
    assert vals[3][order].tolist() == e["count_per_group"]
    assert np.allclose(vals[1][order], e["sum_float_per_group"], rtol=g["float_rel_tol"])
    assert np.allclose(vals[4][order], e["sum_float_per_group"], rtol=g["float_rel_tol"])
    assert np.allclose(vals[2][order], e["avg_int_per_group"], rtol=g["float_rel_tol"])


def _test_table(golden):
    rows = golden["sql_golden"]["test_table"]
    keep = [r for r in rows if r["int_col"] is not None]
    return rows, keep


def test_sql_golden_select_group_by(oracle, golden):
    rows, keep = _test_table(golden)
    sel = golden["sql_golden"]["select"]
    assert len(rows) == sel["count_star"]
    # GROUP BY long_col/100 (integer division) over rows with non-NULL int_col contributions:
    # SUM(int_col) skips NULLs, COUNT(*) counts all rows; HAVING MIN(float_col) > 0 drops the group holding x = 0.
    g1 = np.array([r["long_col"] // 100 for r in rows], dtype=np.int64)
    # the stripe holds an arbitrary value under a NULL (here: a large one that would show in every sum)
    ints = np.array([1 << 30 if r["int_col"] is None else r["int_col"] for r in rows], dtype=np.int32)
    int_nulls = oracle.bitmap_from_bools(np.array([r["int_col"] is None for r in rows]))
    cfg = T.make_agg_config(T.AGG_GENERIC, [(T.LONG, None), (T.INT, None)], keys=[0],
                            aggs=[(T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(1))], est_groups=8, nullable=[1])
    st = oracle.AggState(cfg)
    st.update_nullable([g1, ints], [None, int_nulls])
    keys, vals, _ = st.finalize()
    got = {int(k): (int(c), int(s)) for k, c, s in zip(keys[0], vals[0], vals[1])}
    for e in sel["group_by_long_div_100"]:
        assert got[e["g"]] == (e["count"], e["sum_int"])
    # two keys, HAVING group_col2 > 5
    g2 = np.array([r["long_col"] // 50 for r in rows], dtype=np.int64)
    cfg2 = T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.INT, None)], keys=[0, 1],
                             aggs=[(T.AGG_COUNT_STAR, None)], est_groups=16)
    st2 = oracle.AggState(cfg2)
    st2.update([g1.astype(np.int32), g2.astype(np.int32)])
    keys2, vals2, _ = st2.finalize()
    got2 = sorted((int(c), int(a), int(b)) for a, b, c in zip(keys2[0], keys2[1], vals2[0]) if b > 5)
    assert got2 == sorted((e["count"], e["g1"], e["g2"]) for e in sel["group_by_two_keys_gt5"])
    # GROUP BY int_col: tuples with a NULL key form no group (the listing has 22 rows, PackedPayloadHashTable.hpp:861-867)
    cfg3 = T.make_agg_config(T.AGG_GENERIC, [(T.INT, None)], keys=[0], aggs=[(T.AGG_COUNT_STAR, None)], est_groups=32, nullable=[0])
    st3 = oracle.AggState(cfg3)
    st3.update_nullable([ints], [int_nulls])
    keys3, vals3, _ = st3.finalize()
    assert sorted(keys3[0].tolist()) == sel["distinct_int_col"] and vals3[0].tolist() == [1] * len(sel["distinct_int_col"])


def nullable_test_table(oracle, golden):
    """(columns, null bitmaps, config columns) of the 25-row test table with its NULLs as NULLs."""
    rows = golden["sql_golden"]["test_table"]
    ints = np.array([-77 if r["int_col"] is None else r["int_col"] for r in rows], dtype=np.int32)
    longs = np.array([r["long_col"] for r in rows], dtype=np.int64)
    floats = np.array([r["float_col"] for r in rows], dtype=np.float32)
    doubles = np.array([1e300 if r["double_col"] is None else r["double_col"] for r in rows], dtype=np.float64)
    int_nulls = oracle.bitmap_from_bools(np.array([r["int_col"] is None for r in rows]))
    double_nulls = oracle.bitmap_from_bools(np.array([r["double_col"] is None for r in rows]))
    return [ints, longs, floats, doubles], [int_nulls, None, None, double_nulls]


def select_scalar_with_nulls_config():
    # SELECT COUNT(*), SUM(int_col), AVG(int_col+0), MAX(double_col+100), MIN(float_col+1), COUNT(int_col) FROM test
    return T.make_agg_config(
        T.AGG_SINGLE_STATE, [(T.INT, None), (T.LONG, None), (T.FLOAT, None), (T.DOUBLE, None)],
        instrs=[(T.EX_ADD, 0, T.col(0), T.const(0)), (T.EX_ADD, 1, T.col(3), T.const(1)), (T.EX_ADD, 2, T.col(2), T.const(2))],
        consts=[0.0, 100.0, 1.0],
        aggs=[(T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(0)), (T.AGG_AVG, T.temp(0)), (T.AGG_MAX, T.temp(1)),
              (T.AGG_MIN, T.temp(2)), (T.AGG_COUNT, T.col(0))],
        nullable=[0, 3])


def check_select_scalar_with_nulls(vals, nulls, e):
    count_star = int(vals[0][0])
    assert count_star == e["count_star"] and int(vals[5][0]) == 22
    assert int(vals[1][0] / count_star) == e["sum_int_div_count"]          # integer division of -18 by 25
    assert vals[2][0] * count_star == pytest.approx(e["avg_int_plus_0_times_count"], rel=1e-12)
    assert vals[3][0] == pytest.approx(e["max_double_plus_100"], rel=1e-12)
    assert vals[4][0] == pytest.approx(e["min_float_plus_1"], rel=1e-6)
    assert not any(int(z[0]) for z in nulls)


def test_sql_golden_select_scalar_aggregates_skip_nulls(oracle, golden):
    """Select.test:609-623: aggregates over a table whose int_col / double_col are NULL in rows 0, 10, 20."""
    cols, nulls = nullable_test_table(oracle, golden)
    st = oracle.AggState(select_scalar_with_nulls_config())
    st.update_nullable(cols, nulls)
    _, vals, flags = st.finalize()
    check_select_scalar_with_nulls(vals, flags, golden["sql_golden"]["select"]["scalar_with_nulls"])
    # every argument NULL: SUM / AVG / MIN / MAX are NULL, COUNT(x) is 0, COUNT(*) counts the rows
    all_null = oracle.bitmap_from_bools(np.ones(25, dtype=bool))
    st = oracle.AggState(select_scalar_with_nulls_config())
    st.update_nullable(cols, [all_null, None, None, all_null])
    _, vals, flags = st.finalize()
    assert int(vals[0][0]) == 25 and int(vals[5][0]) == 0
    assert [int(z[0]) for z in flags] == [0, 1, 1, 1, 0, 0]


def left_outer_join_unique(oracle, probe_keys, probe_is_null, build_keys):
    """Build tid (or -1) per probe row for a build side with unique keys; NULL probe keys are not looked up
    (HashTable.hpp:2158-2160)."""
    t = oracle.JoinTable(T.LONG, build_keys.size)
    t.build(build_keys)
    lookup = oracle.bitmap_from_bools(~probe_is_null)
    p, b = t.probe(probe_keys, filter_bitmap=lookup)
    out = np.full(probe_keys.size, -1, dtype=np.int64)
    out[p] = b
    return out


def test_sql_golden_left_join_chains_with_null_keys(oracle, golden):
    """Join.test:136-196."""
    j = golden["sql_golden"]["join"]
    a = j["a"]
    aw = np.array([r["w"] for r in a], dtype=np.int64)
    ax = np.array([r["x"] for r in a], dtype=np.int64)
    ay = np.array([r["y"] for r in a], dtype=np.int64)          # whole numbers: joined as LONG images
    bw = aw[aw % 2 == 0]
    bx = (ax + (aw // 2) % 2)[aw % 2 == 0]
    cx = ax[ax % 3 == 0]
    cy = (ay + (ax // 3) % 3 - 1)[ax % 3 == 0]
    never = np.zeros(aw.size, dtype=bool)

    def column(values, tids):
        return [None if t < 0 else int(values[t]) for t in tids]

    # a LEFT JOIN b ON a.w = b.w LEFT JOIN c ON a.x = c.x
    e = j["left_join_on_a"]
    tb = left_outer_join_unique(oracle, aw, never, bw)
    tc = left_outer_join_unique(oracle, ax, never, cx)
    assert column(bx, tb) == e["b_x"] and column(cy, tc) == [None if v is None else int(v) for v in e["c_y"]]
    # ... LEFT JOIN c ON b.x = c.x LEFT JOIN d ON c.y = d.y: the padded rows probe with NULL keys
    e = j["left_join_chained"]
    b_x = np.where(tb < 0, 123456, bx[np.maximum(tb, 0)])      # arbitrary value under the NULLs
    tc2 = left_outer_join_unique(oracle, b_x, tb < 0, cx)
    c_y = np.where(tc2 < 0, 1200, cy[np.maximum(tc2, 0)])      # 1200 would match d if it were looked up
    td = left_outer_join_unique(oracle, c_y, tc2 < 0, ay)
    assert column(bx, tb) == e["b_x"]
    assert column(cy, tc2) == [None if v is None else int(v) for v in e["c_y"]]
    assert column(aw, td) == e["d_z_w"]


def test_sql_golden_lip(oracle, golden):
    lip = golden["sql_golden"]["lip"]
    x = np.arange(0, lip["limit"] + 1, lip["r_step"], dtype=np.int32)     # R.x = R.y
    z = np.arange(0, lip["limit"] + 1, lip["s_step"], dtype=np.int32)     # S.z
    # exact filter on [min, max] of S.z (AttachLIPFilters picks BitVectorExactFilter)
    f = oracle.LipFilter(T.LIP_BITVECTOR_EXACT, int(z.max() - z.min() + 1), int(z.min()))
    f.build(z)
    hit = oracle.bools_from_bitmap(f.probe(x), x.size)
    semi = x[hit]
    assert semi[semi % 10000 == 0].tolist() == lip["semi_join_mod_10000"]
    assert int(semi[semi % 5 == 0].sum()) + int(semi[semi % 7 == 0].sum()) == lip["sum_x_union_mod5_mod7"]
    # same answer through the hash table's existence probe (FilterJoin == semi join)
    t = oracle.JoinTable(T.INT, z.size)
    t.build(z)
    assert np.array_equal(oracle.bools_from_bitmap(t.probe_exists(x), x.size), hit)
    # the approximate filter may only add false positives
    g = oracle.LipFilter(T.LIP_SINGLE_IDENTITY_HASH, 1024)
    g.build(z)
    approx = oracle.bools_from_bitmap(g.probe(x), x.size)
    assert (approx | ~hit).all()
    # the SUM through the aggregation restatement (Long result)
    both = np.concatenate([semi[semi % 5 == 0], semi[semi % 7 == 0]])
    cfg = T.make_agg_config(T.AGG_SINGLE_STATE, [(T.INT, None)], aggs=[(T.AGG_SUM, T.col(0))])
    st = oracle.AggState(cfg)
    st.update([both])
    assert st.finalize()[1][0].tolist() == [lip["sum_x_union_mod5_mod7"]]


def test_sql_golden_three_way_join(oracle, golden):
    j = golden["sql_golden"]["join"]
    a = j["a"]
    aw = np.array([r["w"] for r in a], dtype=np.int32)
    ax = np.array([r["x"] for r in a], dtype=np.int64)
    ay = np.array([r["y"] for r in a])
    bsel = aw % 2 == 0
    bw, bx = aw[bsel], ax[bsel] + (aw[bsel] // 2) % 2
    csel = ax % 3 == 0
    cx, cy = ax[csel], ay[csel] + (ax[csel] // 3) % 3 - 1
    # a JOIN b ON a.w = b.w
    tb = oracle.JoinTable(T.INT, bw.size)
    tb.build(bw)
    pa, pb = tb.probe(aw)
    # JOIN c ON a.x = c.x
    tc = oracle.JoinTable(T.LONG, cx.size)
    tc.build(cx)
    p2, pc = tc.probe(ax[pa])
    rows = sorted((int(aw[pa][i]), int(bx[pb][i]), float(cy[k])) for i, k in zip(p2, pc))
    # JOIN d ON a.y = d.y keeps every row (d = all y of a)
    assert rows == [(e["w"], e["b_x"], e["c_y"]) for e in j["three_way_join_expected"]]


def test_collision_free_matches_generic(oracle):
    rng = np.random.default_rng(5)
    n = 20000
    key = rng.integers(0, 700, size=n).astype(np.int32)
    val = rng.integers(-1000, 1000, size=n).astype(np.int64)
    dbl = rng.normal(size=n)
    layout = [(T.INT, None), (T.LONG, None), (T.DOUBLE, None)]
    aggs = [(T.AGG_SUM, T.col(1)), (T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(2))]
    cf = oracle.AggState(T.make_agg_config(T.AGG_COLLISION_FREE, layout, keys=[0], aggs=aggs, num_entries=701))
    ge = oracle.AggState(T.make_agg_config(T.AGG_GENERIC, layout, keys=[0], aggs=aggs, est_groups=700))
    for st in (cf, ge):
        st.update([key, val, dbl])
    parts = [cf.finalize(p, 3) for p in range(3)]
    k = np.concatenate([p[0][0] for p in parts])
    assert np.array_equal(k, np.unique(key))              # ascending key order across range partitions
    gk, gv, _ = ge.finalize()
    order = np.argsort(gk[0])
    for a in range(3):
        v = np.concatenate([p[1][a] for p in parts])
        assert np.allclose(v, gv[a][order], rtol=1e-12)
    # generic partitioned finalize covers every group exactly once
    pk = np.concatenate([ge.finalize(p, 41)[0][0] for p in range(41)])
    assert np.array_equal(np.sort(pk), np.unique(key))


def test_existence_map_equals_left_outer_join_then_group_by(oracle):
    """CrossReferenceCoalesceAggregate is the fusion of `left LEFT OUTER JOIN right GROUP BY left.key` with aggregates
    over the right side (rules/FuseAggregateJoin.cpp:60-160): existence bits from the left keys, states from the right
    rows.  Checked against the unfused plan written with numpy: every left key appears, COUNT = number of matching
    right rows passing the fused filter, SUM over no rows = 0 (CollisionFreeVectorTable.hpp:700-727)."""
    rng = np.random.default_rng(8)
    left = rng.permutation(500).astype(np.int32)[:400]                  # unique left keys
    right_key = rng.choice(left[:300], size=5000).astype(np.int32)      # FK into the left relation; 100 keys unmatched
    right_val = rng.integers(-50, 50, size=5000).astype(np.int64)
    cfg = T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.LONG, None)], keys=[0],
                            aggs=[(T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(1))], pred=[(1, T.GT, -10)], num_entries=500)
    st = oracle.AggState(cfg)
    st.mark_existence(left)
    st.update([right_key, right_val])
    keys, vals, _ = st.finalize()
    assert np.array_equal(keys[0], np.sort(left))
    sel = right_val > -10
    want_count = np.bincount(right_key[sel], minlength=500)[keys[0]]
    want_sum = np.bincount(right_key[sel], weights=right_val[sel], minlength=500)[keys[0]].astype(np.int64)
    assert np.array_equal(vals[0], want_count) and np.array_equal(vals[1], want_sum)
    assert (vals[0] == 0).sum() >= 100


def test_sql_golden_distinct(oracle):
    """query_optimizer/tests/execution_generator/Distinct.test:18-72: foo(x, y, z) = (i, (i + 0.5) % 100, i % 3), i < 30000.
    The distinctify table (oracle.distinct_rows) feeding the aggregate over the distinct tuples reproduces the three
    result tables of the reference."""
    i = np.arange(30000)
    x, y, z = i.astype(np.int32), np.fmod(i + 0.5, 100.0), (i % 3).astype(np.int32)
    # COUNT(*), COUNT(DISTINCT x), COUNT(DISTINCT y), COUNT(DISTINCT z)  ->  30000, 30000, 100, 3
    assert [oracle.distinct_rows([c]).size for c in (x, y, z)] == [30000, 100, 3]
    # SUM(y), SUM(DISTINCT y), COUNT(DISTINCT y), AVG(DISTINCT y), z GROUP BY z  ->  500000, 5000, 100, 50 per z
    rows = oracle.distinct_rows([z, y])
    cfg = T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.DOUBLE, None)], keys=[0],
                            aggs=[(T.AGG_SUM, T.col(1)), (T.AGG_COUNT_STAR, None), (T.AGG_AVG, T.col(1))], est_groups=3)
    st = oracle.AggState(cfg)
    st.update([z[rows], y[rows]])
    keys, vals, _ = st.finalize()
    order = np.argsort(keys[0])
    assert keys[0][order].tolist() == [0, 1, 2]
    assert vals[0][order].tolist() == [5000.0] * 3 and vals[1][order].tolist() == [100] * 3 and vals[2][order].tolist() == [50.0] * 3
    plain = oracle.AggState(T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.DOUBLE, None)], keys=[0],
                                              aggs=[(T.AGG_SUM, T.col(1))], est_groups=3))
    plain.update([z, y])
    assert np.allclose(plain.finalize()[1][0], 500000.0, rtol=1e-12)
    # MAX(x) * SUM(DISTINCT y), COUNT(DISTINCT x % y) + z  ->  149985000 | 196, 149990000 | 197, 149995000 | 195
    w = np.fmod(x.astype(np.float64), y)
    rows = oracle.distinct_rows([z, w])
    counts = np.bincount(z[rows], minlength=3)
    assert (counts + np.arange(3)).tolist() == [196, 197, 195]
    assert [int(x[z == g].max() * 5000.0) for g in range(3)] == [149985000, 149990000, 149995000]
    # the representative of a tuple is its first occurrence, listed in tuple order
    rows = oracle.distinct_rows([z])
    assert rows.tolist() == [0, 1, 2]


def test_join_unittest_composite_key_and_residual(oracle, golden):
    """CompositeKeyHashJoinTest / ...WithResidualPredicateTest (HashJoinOperator_unittest.cpp:999-1375)
    on the composite-key restatement: 100 results (even tids), 8 with the residual dim.long < 15."""
    g = golden["join_unittest"]
    dim = np.arange(g["num_dim_tuples"], dtype=np.int64)
    fact = np.arange(g["num_fact_tuples"], dtype=np.int64)
    t = oracle.CompositeJoinTable([T.LONG, T.LONG], dim.size)
    for b in range(0, dim.size, g["block_size"]):                 # one build work order per block
        blk = dim[b:b + g["block_size"]]
        t.build([blk, blk // 2 * 2], block_id=b // g["block_size"], base_tid=b)
    p, d = t.probe([fact, fact])
    assert p.size == g["composite_key"]["expected_num_results"]
    assert np.array_equal(np.sort(d), np.arange(0, dim.size, 2)) and np.array_equal(np.sort(p), np.sort(d))
    keep = dim[d] < g["composite_key_residual"]["residual_dim_long_less_than"]
    assert int(keep.sum()) == g["composite_key_residual"]["expected_num_results"]
    assert sorted(d[keep].tolist()) == list(range(0, 15, 2))


def test_composite_hash_is_the_combine_hashes_fold(oracle):
    """hashCompositeKey (storage/HashTable.hpp:2109-2119): fold of CombineHashes over the components'
    identity hashes; CombineHashes itself is pinned against the reference header above."""
    t = oracle.CompositeJoinTable([T.LONG, T.INT, T.LONG], 4)
    a = np.array([7, -1], dtype=np.int64)
    b = np.array([3, -2], dtype=np.int32)
    c = np.array([1 << 40, 0], dtype=np.int64)
    got = t.hash_rows([a, b, c])
    for i in range(2):
        h = oracle.combine_hashes(oracle.hash_scalar(T.LONG, a[i]), oracle.hash_scalar(T.INT, b[i]))
        h = oracle.combine_hashes(h, oracle.hash_scalar(T.LONG, c[i]))
        assert int(got[i]) == h


def test_agg_unittest_min_max(oracle, golden):
    """ScalarAttribute_*_{Max,Min}_* (:675-712, :959-996) and GroupBy_{Max,Min}_* (:1550-1680)."""
    g = golden["agg_unittest"]
    val = np.arange(g["num_tuples"])
    gid = val % g["group_by_width"]
    cols = [(gid % g["group_by_1_size"]).astype(np.int32), (gid // g["group_by_1_size"]).astype(np.int32),
            val.astype(np.int32), 0.1 * val, (0.1 * val).astype(np.float32)]
    layout = [(T.INT, None), (T.INT, None), (T.INT, None), (T.DOUBLE, None), (T.FLOAT, None)]
    aggs = [(T.AGG_MAX, T.col(2)), (T.AGG_MIN, T.col(2)), (T.AGG_MAX, T.col(3)), (T.AGG_MAX, T.col(4))]
    for with_pred in (False, True):
        s = g["scalar"]
        st = oracle.AggState(T.make_agg_config(T.AGG_SINGLE_STATE, layout, aggs=aggs,
                                               pred=[(2, T.LT, s["predicate_less_than"])] if with_pred else []))
        st.update(cols)
        _, vals, nulls = st.finalize()
        mx = s["max_with_predicate"] if with_pred else s["max_no_predicate"]
        assert vals[0][0] == mx and vals[1][0] == s["min"] and vals[2][0] == 0.1 * mx and vals[3][0] == np.float32(0.1 * mx)
        assert vals[0].dtype == np.int32 and vals[3].dtype == np.float32 and not any(z[0] for z in nulls)
        e = g["group_by_min_max"]["with_predicate" if with_pred else "without_predicate"]
        for strategy in (T.AGG_COMPACT_KEY, T.AGG_GENERIC):
            st = oracle.AggState(T.make_agg_config(strategy, layout, keys=[0, 1], aggs=aggs, est_groups=20,
                                                   pred=[(2, T.LT, g["group_by_predicate_less_than"])] if with_pred else []))
            for b in range(0, val.size, 10):
                st.update([c[b:b + 10] for c in cols])
            keys, vals, _ = st.finalize()
            order = np.argsort(keys[0] + keys[1] * g["group_by_1_size"])
            assert vals[0][order].tolist() == e["max_int_per_group"] and vals[1][order].tolist() == e["min_int_per_group"]
            assert np.array_equal(vals[2][order], 0.1 * np.array(e["max_int_per_group"]))
    # zero rows: NULL
    st = oracle.AggState(T.make_agg_config(T.AGG_SINGLE_STATE, layout, aggs=aggs, pred=[(2, T.LT, -1)]))
    st.update(cols)
    assert all(z[0] == 1 for z in st.finalize()[2])


def _compressed_cases():
    rng = np.random.default_rng(9)
    yield "int truncated to 1 byte", rng.integers(0, 200, size=5000).astype(np.int32), 1, 1
    yield "long truncated to 2 bytes", rng.integers(0, 60000, size=5000).astype(np.int64), 1, 2
    yield "long truncated to 4 bytes", rng.integers(0, 2**31, size=5000).astype(np.int64), 1, 4
    yield "int dictionary (negative values, few distinct)", rng.choice(np.array([-7, -1, 3, 900, 10**6], dtype=np.int32), size=5000), 2, 1
    yield "double dictionary (TPC-H discount)", rng.integers(0, 11, size=5000) / 100.0, 2, 1
    yield "float dictionary, 2-byte codes", (rng.integers(0, 700, size=20000) * 0.5).astype(np.float32), 2, 2
    yield "int incompressible", rng.integers(-2**31, 2**31 - 1, size=3000).astype(np.int32), 0, 4


def test_compression_choice_follows_the_block_builder(oracle):
    """CompressedBlockBuilder (storage/CompressedBlockBuilder.cpp:508-566, 590-650): truncation by the leading zeros of the
    maximum (non-negative INT/LONG only), dictionary code width by the number of distinct values, the smaller wins."""
    for name, values, kind, width in _compressed_cases():
        col = oracle.CompressedColumn(values)
        assert (col.kind, col.code_width) == (kind, width), name
        assert np.array_equal(col.decode(), values), name
        if kind == 2:
            assert np.array_equal(col.dictionary, np.unique(values)), name
    # LONG maximum == UINT32_MAX must not be truncated (:538-542); one negative value forbids truncation
    assert oracle.CompressedColumn(np.array([5, 2**32 - 1] * 300, dtype=np.int64)).kind == 2
    assert oracle.CompressedColumn(np.arange(-1, 4000, dtype=np.int64)).kind != 1


def test_predicates_on_codes_equal_predicates_on_values(oracle):
    """TransformPredicateOnCompressedAttribute + the code-stripe scans (CompressedStoreUtil.cpp:51-140, 425-616;
    CompressedColumnStoreTupleStorageSubBlock.cpp:420-760): for every comparison and literals below, inside, between and
    above the stored values the matches on codes are the matches of the comparison on the decoded values."""
    rng = np.random.default_rng(10)
    for name, values, kind, width in _compressed_cases():
        if kind == 0:
            continue
        col = oracle.CompressedColumn(values)
        lits = [values.min(), values.max(), np.sort(values)[values.size // 2], values.min() - 1, values.max() + 1]
        if values.dtype.kind == "f":
            lits += [np.sort(np.unique(values))[3] + values.dtype.type(0.001), values.dtype.type(-0.5)]
        else:
            lits += [0, 1, 255, 256, 65535, 65536]
        f = oracle.bitmap_from_bools(rng.random(values.size) < 0.5)
        for lit in lits:
            if values.dtype.kind != "f" and not (np.iinfo(values.dtype).min <= int(lit) <= np.iinfo(values.dtype).max):
                continue
            lit = values.dtype.type(lit)
            for op in range(6):
                want = oracle.select_cmp(values, op, lit)
                assert np.array_equal(col.matches(op, lit), want), (name, op, lit)
                assert np.array_equal(col.matches(op, lit, filter_bitmap=f), oracle.select_cmp(values, op, lit, filter_bitmap=f)), (name, op, lit)


def test_predicate_transformer_result_kinds(oracle):
    """The shapes the transformer produces (CompressedStoreUtil.cpp:548-612): range ends at the code count -> >=,
    starts at 0 -> <, covers everything -> ALL, empty -> NONE; literals absent from the dictionary."""
    d = oracle.CompressedColumn(np.array([10, 20, 30, 40] * 100, dtype=np.int32) - 25)      # dictionary: -15 -5 5 15
    assert d.kind == 2
    t = d.transform(T.EQ, 5)
    assert (t.result, t.comp, t.first) == (oracle.PRED_BASIC, T.CODE_EQ, 2)
    assert d.transform(T.EQ, 6).result == oracle.PRED_NONE and d.transform(T.NE, 6).result == oracle.PRED_ALL
    t = d.transform(T.LT, 5)
    assert (t.result, t.comp, t.first) == (oracle.PRED_BASIC, T.CODE_LT, 2)
    t = d.transform(T.GT, -5)
    assert (t.result, t.comp, t.first) == (oracle.PRED_BASIC, T.CODE_GE, 2)
    assert d.transform(T.GE, -100).result == oracle.PRED_ALL and d.transform(T.GT, 15).result == oracle.PRED_NONE
    tr = oracle.CompressedColumn(np.arange(0, 200, dtype=np.int64))                          # truncated to 1 byte
    assert tr.kind == 1 and tr.code_width == 1
    assert tr.transform(T.LT, 0).result == oracle.PRED_NONE and tr.transform(T.GE, 0).result == oracle.PRED_ALL
    assert tr.transform(T.EQ, 300).result == oracle.PRED_NONE and tr.transform(T.NE, -1).result == oracle.PRED_ALL
    t = tr.transform(T.LE, 7)
    assert (t.result, t.comp, t.first) == (oracle.PRED_BASIC, T.CODE_LT, 8)
    t = tr.transform(T.GT, 7)
    assert (t.result, t.comp, t.first) == (oracle.PRED_BASIC, T.CODE_GE, 8)
    assert tr.transform(T.LE, 255).result == oracle.PRED_ALL                                  # >= the largest 1-byte code


def test_comparison_unittest_dates_and_strings(oracle, golden):
    """Comparison_unittest.cpp's DATE and CHAR samples under all six comparisons: the oracle's DateLit comparison (year,
    month, day; padding bytes ignored) and its strcmpHelper restatement against the expected truth tables."""
    g = golden["comparison_unittest"]
    ops = {"eq": T.EQ, "ne": T.NE, "lt": T.LT, "le": T.LE, "gt": T.GT, "ge": T.GE}
    dates = g["dates"]
    col = np.array([T.date_raw(y, m, d, padding=0xBEEF) for y, m, d in dates], dtype=np.int64)   # garbage in the padding bytes
    for name, op in ops.items():
        for j, (y, m, d) in enumerate(dates):
            bm = oracle.select_cmp(col, op, T.date_raw(y, m, d), qt=T.DATE)
            got = [bool((int(bm[0]) >> (63 - i)) & 1) for i in range(len(dates))]
            assert got == [row[j] for row in g["date_tables"][name]], (name, j)
        # attribute OP attribute, every pair at once
        lhs = np.repeat(col, len(dates))
        rhs = np.tile(np.array([T.date_raw(y, m, d) for y, m, d in dates], dtype=np.int64), len(dates))
        bm = oracle.select_cmp_columns(lhs, rhs, op, qt=T.DATE)
        want = [v for row in g["date_tables"][name] for v in row]
        assert [bool((int(bm[i >> 6]) >> (63 - (i & 63))) & 1) for i in range(len(want))] == want
    strings = g["strings"]
    for name, op in ops.items():
        for i, (ltext, lwidth) in enumerate(strings):
            stripe = np.zeros((1, lwidth), dtype=np.uint8)
            stripe[0, :len(ltext)] = np.frombuffer(ltext.encode(), dtype=np.uint8)
            for j, (rtext, rwidth) in enumerate(strings):
                if rwidth > 64:
                    continue                      # (the device ABI takes literals of up to 64 bytes; the long sample is a left side only)
                literal = rtext.encode() + b"\0" * (rwidth - len(rtext))
                bm = oracle.select_cmp_char(stripe, op, literal)
                assert bool(int(bm[0]) >> 63) == g["string_tables"][name][i][j], (name, i, j)
