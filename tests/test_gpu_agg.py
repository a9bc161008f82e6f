"""GPU parity: K6/K7/K8 aggregation (+ fused K11 expressions, state predicate,
K10 finalize, merge/export) through the C ABI against the oracle.

Bar (BASELINE.json north_star): COUNT and SUM over INT/LONG bit-exact; SUM/AVG
over FLOAT/DOUBLE within 1e-6 relative (summation order differs).  Group rows
are compared as sorted sets: output order is the bucket order of whichever
table survived and is unspecified in the reference."""
import numpy as np
import pytest
import torch

from quickstep_amd import types as T
from helpers import bitmap_dev, to_dev

pytestmark = pytest.mark.gpu

FP_RTOL = 1e-6   # tolerance stated by north_star for FLOAT/DOUBLE SUM/AVG


def run_hip(capi, dev, cfg, cols, blocks=1, filter_bitmap=None, partition=0, num_partitions=1, state=None):
    st = capi.AggState(cfg) if state is None else state
    n = cols[0].size if cols else 0
    dcols = [to_dev(c, dev) for c in cols]
    if blocks == 1:
        st.update(dcols, n, filter_bitmap=None if filter_bitmap is None else bitmap_dev(filter_bitmap, dev))
    else:
        assert filter_bitmap is None
        edges = np.linspace(0, n, blocks + 1).astype(np.int64)
        for b in range(blocks):                            # one AggregationWorkOrder per block
            st.update([c[edges[b]:edges[b + 1]] for c in dcols], int(edges[b + 1] - edges[b]))
    return st


def finalize_np(st, dev, partition=0, num_partitions=1):
    keys, vals, nulls, groups = st.finalize(dev, partition, num_partitions)
    g = int(groups.item())
    if g == T.GROUPS_HASH_COLLISION:
        raise RuntimeError("QSX_GROUPS_HASH_COLLISION")
    return [k.cpu().numpy()[:g] for k in keys], [v.cpu().numpy()[:g] for v in vals], [z.cpu().numpy()[:g] for z in nulls]


def assert_same_groups(got, ref):
    gk, gv, gn = got
    rk, rv, rn = ref
    assert (gk[0].size if gk else gv[0].size) == (rk[0].size if rk else rv[0].size)
    go = np.lexsort([np.asarray(k) for k in gk[::-1]]) if gk else np.arange(gv[0].size)
    ro = np.lexsort([np.asarray(k) for k in rk[::-1]]) if rk else np.arange(rv[0].size)
    for a, b in zip(gk, rk):
        assert np.array_equal(a[go], b[ro])
    for a, b, na, nb in zip(gv, rv, gn, rn):
        assert np.array_equal(na[go], nb[ro])
        if a.dtype == np.int64:
            assert np.array_equal(a[go], b[ro])            # integer results: bit-exact
        else:
            assert np.allclose(a[go], b[ro], rtol=FP_RTOL, atol=0.0)


def agg_rows(g):
    val = np.arange(g["num_tuples"])
    gid = val % g["group_by_width"]
    return dict(gb0=(gid % g["group_by_1_size"]).astype(np.int32), gb1=(gid // g["group_by_1_size"]).astype(np.int32),
                i=val.astype(np.int32), l=val.astype(np.int64), f=(0.1 * val).astype(np.float32), d=0.1 * val)


def test_golden_scalar_aggregates(capi, dev, golden):
    g = golden["agg_unittest"]
    s = g["scalar"]
    r = agg_rows(g)
    cols = [r["i"], r["l"], r["f"], r["d"]]
    layout = [(T.INT, None), (T.LONG, None), (T.FLOAT, None), (T.DOUBLE, None)]
    aggs = [(T.AGG_SUM, T.col(0)), (T.AGG_SUM, T.col(1)), (T.AGG_SUM, T.col(3)), (T.AGG_AVG, T.col(0)),
            (T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(2))]
    for pred, exp_sum, exp_cnt in ((None, s["sum_int_no_predicate"], s["count_no_predicate"]),
                                   (s["predicate_less_than"], s["sum_int_with_predicate"], s["count_with_predicate"])):
        cfg = T.make_agg_config(T.AGG_SINGLE_STATE, layout, aggs=aggs, pred=[] if pred is None else [(0, T.LT, pred)])
        st = run_hip(capi, dev, cfg, cols, blocks=30)
        _, vals, nulls = finalize_np(st, dev)
        assert vals[0].tolist() == [exp_sum] and vals[1].tolist() == [exp_sum] and vals[4].tolist() == [exp_cnt]
        assert vals[2][0] == pytest.approx(0.1 * exp_sum, rel=g["float_rel_tol"])
        assert vals[5][0] == pytest.approx(0.1 * exp_sum, rel=g["float_rel_tol"])
        assert vals[3][0] == pytest.approx(exp_sum / exp_cnt, rel=1e-12)
        assert not any(z[0] for z in nulls)
    cfg = T.make_agg_config(T.AGG_SINGLE_STATE, layout, aggs=aggs, pred=[(0, T.LT, s["zero_rows_predicate_less_than"])])
    st = run_hip(capi, dev, cfg, cols)
    _, vals, nulls = finalize_np(st, dev)
    assert vals[4].tolist() == [0] and [int(z[0]) for z in nulls] == [1, 1, 1, 1, 0, 1]


@pytest.mark.parametrize("strategy", [T.AGG_COMPACT_KEY, T.AGG_GENERIC])
@pytest.mark.parametrize("with_pred", [False, True])
def test_golden_group_by(capi, dev, golden, strategy, with_pred):
    g = golden["agg_unittest"]
    r = agg_rows(g)
    e = g["group_by"]["with_predicate" if with_pred else "without_predicate"]
    cols = [r["gb0"], r["gb1"], r["i"], r["d"], r["f"]]
    cfg = T.make_agg_config(
        strategy, [(T.INT, None), (T.INT, None), (T.INT, None), (T.DOUBLE, None), (T.FLOAT, None)], keys=[0, 1],
        aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.col(3)), (T.AGG_AVG, T.col(2)), (T.AGG_COUNT_STAR, None),
              (T.AGG_SUM, T.col(4))],
        pred=[(2, T.LT, g["group_by_predicate_less_than"])] if with_pred else [], est_groups=20)
    st = run_hip(capi, dev, cfg, cols, blocks=30)
    keys, vals, _ = finalize_np(st, dev)
    gid = keys[0] + keys[1] * g["group_by_1_size"]
    order = np.argsort(gid)
    assert gid[order].tolist() == list(range(20))
    assert vals[0][order].tolist() == e["sum_int_per_group"]
    assert vals[3][order].tolist() == e["count_per_group"]
    assert np.allclose(vals[1][order], e["sum_float_per_group"], rtol=g["float_rel_tol"])
    assert np.allclose(vals[4][order], e["sum_float_per_group"], rtol=g["float_rel_tol"])
    assert np.allclose(vals[2][order], e["avg_int_per_group"], rtol=g["float_rel_tol"])


def q1_config(with_date_pred=False, est_groups=6):
    cols = [(T.CHAR, 1), (T.CHAR, 1), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None)]
    pred = []
    if with_date_pred:
        cols.append((T.LONG, None))                       # stand-in for the 8-byte DateLit l_shipdate
        pred = [(6, T.LE, 19980902)]
    return T.make_agg_config(
        T.AGG_COMPACT_KEY, cols, keys=[0, 1],
        # t0 = 1 - disc ; t1 = price * t0 ; t2 = 1 + tax ; t3 = t1 * t2   (benchmarks/tpch/queries/01.sql)
        instrs=[(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0)),
                (T.EX_ADD, 2, T.const(0), T.col(5)), (T.EX_MUL, 3, T.temp(1), T.temp(2))],
        consts=[1.0],
        aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.col(3)), (T.AGG_SUM, T.temp(1)), (T.AGG_SUM, T.temp(3)),
              (T.AGG_AVG, T.col(2)), (T.AGG_AVG, T.col(3)), (T.AGG_AVG, T.col(4)), (T.AGG_COUNT_STAR, None)],
        pred=pred, est_groups=est_groups)


def q1_columns(rng, n, with_date=False):
    combo = rng.choice(4, size=n, p=[0.2466, 0.0065, 0.5005, 0.2464])     # (A,F) (N,F) (N,O) (R,F)
    k1 = np.frombuffer(b"ANNR", dtype=np.uint8)[combo]
    k2 = np.frombuffer(b"FFOF", dtype=np.uint8)[combo]
    cols = [k1, k2, rng.integers(1, 51, size=n).astype(np.float64), np.round(rng.uniform(900, 105000, size=n), 2),
            rng.integers(0, 11, size=n) / 100.0, rng.integers(0, 9, size=n) / 100.0]
    if with_date:
        cols.append(rng.integers(19920101, 19981231, size=n).astype(np.int64))
    return cols


@pytest.mark.parametrize("n", [0, 1, 255, 70_001, 2_000_000])
@pytest.mark.parametrize("with_date", [False, True])
def test_q1_shape_matches_oracle(capi, oracle, dev, n, with_date):
    rng = np.random.default_rng(n + 4)
    cols = q1_columns(rng, n, with_date)
    cfg = q1_config(with_date)
    st = run_hip(capi, dev, cfg, cols, blocks=3 if n > 1000 else 1)
    o = oracle.AggState(cfg)
    o.update(cols, n)
    assert_same_groups(finalize_np(st, dev), o.finalize())


@pytest.mark.parametrize("strategy,groups", [(T.AGG_COMPACT_KEY, 7), (T.AGG_COMPACT_KEY, 300), (T.AGG_COMPACT_KEY, 9000),
                                             (T.AGG_GENERIC, 50_000)])
def test_many_groups_two_int_keys(capi, oracle, dev, strategy, groups):
    """BASELINE config 3 minimal variant: two INT32 keys + one DOUBLE value (16 B/row); group
    counts beyond the register groups, beyond the LDS table and beyond LDS altogether."""
    rng = np.random.default_rng(groups)
    n = 400_000
    side = int(np.ceil(np.sqrt(groups)))
    k1 = rng.integers(-3, side - 3, size=n).astype(np.int32)            # negative key components too
    k2 = rng.integers(0, side, size=n).astype(np.int32)
    v = rng.normal(1000.0, 50.0, size=n)
    w = rng.integers(-10**9, 10**9, size=n).astype(np.int64)
    cfg = T.make_agg_config(strategy, [(T.INT, None), (T.INT, None), (T.DOUBLE, None), (T.LONG, None)], keys=[0, 1],
                            aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None), (T.AGG_AVG, T.col(2)),
                                  (T.AGG_SUM, T.col(3)), (T.AGG_AVG, T.col(3))], est_groups=groups)
    filt = oracle.bitmap_from_bools(rng.random(n) < 0.9)
    st = run_hip(capi, dev, cfg, [k1, k2, v, w], filter_bitmap=filt)
    o = oracle.AggState(cfg)
    o.update([k1, k2, v, w], n, filter_bitmap=filt)
    assert st.num_groups() >= o.num_groups()
    assert_same_groups(finalize_np(st, dev), o.finalize())
    if strategy == T.AGG_GENERIC:
        # partitioned finalize: membership of every partition = reference's HashCompositeKey % P
        P = 41                                               # --num_aggregation_partitions default
        seen = 0
        for p in range(0, P, 10):
            assert_same_groups(finalize_np(st, dev, p, P), o.finalize(p, P))
        for p in range(P):
            seen += finalize_np(st, dev, p, P)[0][0].size
        assert seen == o.num_groups()


def test_sentinel_and_wide_keys(capi, oracle, dev):
    """8-byte LONG key including -1 (bit pattern of the table's empty marker), INT64 extremes, and a
    CHAR(2)+CHAR(4)+CHAR(2) = 8-byte compact key."""
    rng = np.random.default_rng(8)
    n = 50_000
    k = rng.choice(np.array([-1, 0, 1, 2**63 - 1, -2**63, 123456789012], dtype=np.int64), size=n)
    v = rng.integers(0, 100, size=n).astype(np.int32)
    for strategy in (T.AGG_COMPACT_KEY, T.AGG_GENERIC):
        cfg = T.make_agg_config(strategy, [(T.LONG, None), (T.INT, None)], keys=[0],
                                aggs=[(T.AGG_SUM, T.col(1)), (T.AGG_COUNT_STAR, None)], est_groups=8)
        st = run_hip(capi, dev, cfg, [k, v], blocks=4)
        o = oracle.AggState(cfg)
        o.update([k, v])
        assert_same_groups(finalize_np(st, dev), o.finalize())
    c2 = rng.integers(0, 3, size=n).astype(np.int16)
    c4 = rng.integers(-2, 2, size=n).astype(np.int32)
    c2b = rng.integers(0, 2, size=n).astype(np.int16) - 1
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, [(T.CHAR, 2), (T.CHAR, 4), (T.CHAR, 2), (T.INT, None)], keys=[0, 1, 2],
                            aggs=[(T.AGG_SUM, T.col(3)), (T.AGG_COUNT_STAR, None)], est_groups=30)
    st = run_hip(capi, dev, cfg, [c2, c4, c2b, v])
    o = oracle.AggState(cfg)
    o.update([c2, c4, c2b, v])
    assert_same_groups(finalize_np(st, dev), o.finalize())


@pytest.mark.parametrize("num_entries,partitions", [(1, 1), (700, 3), (100_000, 7), (1_000_000, 16)])
def test_collision_free_vector(capi, oracle, dev, num_entries, partitions):
    """Q3 group-by path: dense key, SUM(double expr) + COUNT + SUM(int); ascending keys per range partition."""
    rng = np.random.default_rng(num_entries)
    n = 500_000
    key = np.sort(rng.integers(0, num_entries, size=n)).astype(np.int32)      # lineitem is clustered on orderkey
    price = np.round(rng.uniform(900, 105000, size=n), 2)
    disc = rng.integers(0, 11, size=n) / 100.0
    qty = rng.integers(1, 51, size=n).astype(np.int32)
    for aggs in ([(T.AGG_SUM, T.temp(1)), (T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(3))], [(T.AGG_SUM, T.temp(1))]):
        cfg = T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.INT, None)],
                                keys=[0], instrs=[(T.EX_SUB, 0, T.const(0), T.col(2)), (T.EX_MUL, 1, T.col(1), T.temp(0))],
                                consts=[1.0], aggs=aggs, num_entries=num_entries)
        st = run_hip(capi, dev, cfg, [key, price, disc, qty], blocks=5)
        o = oracle.AggState(cfg)
        o.update([key, price, disc, qty])
        assert st.num_groups() == o.num_groups()
        for p in range(partitions):
            gk, gv, gn = finalize_np(st, dev, p, partitions)
            rk, rv, rn = o.finalize(p, partitions)
            assert np.array_equal(gk[0], rk[0])            # same keys in the same (ascending) order
            for a, b in zip(gv, rv):
                if a.dtype == np.int64:
                    assert np.array_equal(a, b)
                else:
                    assert np.allclose(a, b, rtol=FP_RTOL, atol=0.0)


@pytest.mark.parametrize("num_entries", [1, 5, 25, 640, 3000, 16_384, 16_385, 40_000, 65_536, 70_000])
@pytest.mark.parametrize("lds", ["1", "0"])
def test_small_collision_free_states_in_random_key_order(capi, oracle, dev, num_entries, lds, monkeypatch):
    """Dense states of few entries are accumulated in LDS (one workgroup per CU, copies of every entry when there are very
    few, up to eight key-range families of workgroups from 16 Ki entries on) instead of with one global atomic per row and aggregate — same results as the per-row path (QSX_AGG_DENSE_LDS=0)
    and the oracle: random key order, SUM(double expression) + COUNT + SUM(int) + MIN + MAX, one stripe, several calls, a
    run of ragged blocks with per-block filters, interpreter and run-time plan shape, a key outside the range."""
    monkeypatch.setenv("QSX_AGG_DENSE_LDS", lds)
    rng = np.random.default_rng(num_entries)
    n = 300_000
    key = rng.integers(0, num_entries, size=n).astype(np.int32)
    if num_entries > 100:
        key[key % 7 == 3] = 0                                 # some entries stay empty, one is hot
    price = np.round(rng.uniform(900, 105000, size=n), 2)
    disc = rng.integers(0, 11, size=n) / 100.0
    qty = rng.integers(-50, 51, size=n).astype(np.int32)
    layout = [(T.INT, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.INT, None)]
    aggs = [(T.AGG_SUM, T.temp(1)), (T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(3)), (T.AGG_MIN, T.col(1)), (T.AGG_MAX, T.col(3))]
    cfg = T.make_agg_config(T.AGG_COLLISION_FREE, layout, keys=[0], instrs=[(T.EX_SUB, 0, T.const(0), T.col(2)), (T.EX_MUL, 1, T.col(1), T.temp(0))],
                            consts=[1.0], aggs=aggs, num_entries=num_entries)
    cols = [key, price, disc, qty]
    o = oracle.AggState(cfg)
    o.update(cols)
    ref = o.finalize()
    keep = rng.random(n) < 0.6
    of = oracle.AggState(cfg)
    of.update(cols, filter_bitmap=oracle.bitmap_from_bools(keep))
    ref_filtered = of.finalize()
    for jit in (False, True):
        monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", "0" if jit else str(1 << 60))
        for blocks in (1, 4):
            assert_same_groups(finalize_np(run_hip(capi, dev, cfg, cols, blocks=blocks), dev), ref)
        assert_same_groups(finalize_np(run_hip(capi, dev, cfg, cols, filter_bitmap=oracle.bitmap_from_bools(keep)), dev), ref_filtered)
        # a run of ragged blocks (one empty), per-block filters with a gap
        cuts = [0, 1000, 1000, 77_777, 200_001, n]
        dcols = [to_dev(c, dev) for c in cols]
        run = [[c[a:b] for c in dcols] for a, b in zip(cuts[:-1], cuts[1:])]
        st = capi.AggState(cfg)
        st.update_blocks(run)
        assert_same_groups(finalize_np(st, dev), ref)
        filters = [bitmap_dev(oracle.bitmap_from_bools(keep[a:b]), dev) if b > a else None for a, b in zip(cuts[:-1], cuts[1:])]
        st = capi.AggState(cfg)
        st.update_blocks(run, filters=filters)
        assert_same_groups(finalize_np(st, dev), ref_filtered)
    bad = key.copy()
    bad[12345] = num_entries                                  # precondition violated: reported at finalize, as before
    st = run_hip(capi, dev, cfg, [bad, price, disc, qty])
    with pytest.raises(capi.QsxError):
        st.finalize(dev, capacity=num_entries + 1)


def test_merge_and_export_import(capi, oracle, dev):
    """Partial states of two 'GPUs' merged: mergeFrom semantics (ThreadPrivateCompactKeyHashTable.cpp:306-363)."""
    rng = np.random.default_rng(21)
    n = 120_000
    for cfg, cols in (
        (q1_config(), q1_columns(rng, n)),
        (T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.LONG, None)], keys=[0],
                           aggs=[(T.AGG_SUM, T.col(1)), (T.AGG_COUNT_STAR, None)], num_entries=5000),
         [rng.integers(0, 5000, size=n).astype(np.int32), rng.integers(-5, 5, size=n).astype(np.int64)]),
        (T.make_agg_config(T.AGG_SINGLE_STATE, [(T.DOUBLE, None)], aggs=[(T.AGG_AVG, T.col(0)), (T.AGG_COUNT_STAR, None)]),
         [rng.normal(size=n)]),
    ):
        half = n // 2
        a = run_hip(capi, dev, cfg, [c[:half] for c in cols])
        b = run_hip(capi, dev, cfg, [c[half:] for c in cols])
        image = b.export(dev)
        assert image.numel() * 8 == b.export_bytes()
        c = capi.AggState(cfg)
        c.merge(a)
        c.import_merge(image)
        o = oracle.AggState(cfg)
        o.update(cols)
        assert_same_groups(finalize_np(c, dev), o.finalize())


def test_table_grows_past_the_optimizer_estimate(capi, oracle, dev):
    """An estimate orders of magnitude too low must not lose a group (the reference's tables resize:
    PackedPayloadHashTable::resize, ThreadPrivateCompactKeyHashTable::resize): block-at-a-time the table grows between
    update calls; inside one call the groups that find no slot wait in the state's spill log."""
    rng = np.random.default_rng(0)
    n = 400_000
    k = rng.integers(0, 50_000, size=n).astype(np.int32)
    v = rng.integers(-1000, 1000, size=n).astype(np.int64)
    d = rng.normal(size=n)
    layout = [(T.INT, None), (T.LONG, None), (T.DOUBLE, None)]
    aggs = [(T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(1)), (T.AGG_MIN, T.col(2)), (T.AGG_MAX, T.col(1)), (T.AGG_AVG, T.col(2))]
    for strategy in (T.AGG_COMPACT_KEY, T.AGG_GENERIC):
        cfg = T.make_agg_config(strategy, layout, keys=[0], aggs=aggs, est_groups=4)
        o = oracle.AggState(cfg)
        o.update([k, v, d])
        ref = o.finalize()
        for blocks in (1, 40):
            st = run_hip(capi, dev, cfg, [k, v, d], blocks=blocks)
            assert st.num_groups() >= ref[0][0].size
            assert_same_groups(finalize_np(st, dev), ref)
        # a state that has grown keeps working: clear + second run, and its (bigger) image merges into a small table
        st.clear()
        st.update([to_dev(c, dev) for c in (k, v, d)], n)
        assert_same_groups(finalize_np(st, dev), ref)
        image = st.export(dev)
        small = capi.AggState(cfg)
        assert image.numel() * 8 == st.export_bytes() > small.export_bytes()
        small.import_merge(image)
        assert_same_groups(finalize_np(small, dev), ref)
        other = capi.AggState(cfg)
        other.merge(st)
        assert_same_groups(finalize_np(other, dev), ref)


def test_lost_rows_are_reported_not_silently_dropped(capi, dev):
    """One call that spills more groups than the state's log holds (1 Mi records): the error surfaces at
    num_groups / finalize / export, never as a silently smaller result."""
    rng = np.random.default_rng(0)
    k = rng.permutation(3_000_000).astype(np.int32)
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, [(T.INT, None)], keys=[0], aggs=[(T.AGG_COUNT_STAR, None)], est_groups=4)
    st = run_hip(capi, dev, cfg, [k])
    for call in (st.num_groups, lambda: st.finalize(dev, capacity=16), lambda: st.export(dev)):
        with pytest.raises(capi.QsxError) as e:
            call()
        assert e.value.status == T.ERR_TOO_MANY_GROUPS
    cfg = T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None)], keys=[0], aggs=[(T.AGG_COUNT_STAR, None)], num_entries=10)
    st = run_hip(capi, dev, cfg, [k])                       # keys outside [0, num_entries): precondition violated
    with pytest.raises(capi.QsxError):
        st.finalize(dev, capacity=16)


def test_q1_at_scale_properties(capi, dev):
    """C3 shape at 60 M rows (the full 600 M runs in bench.py): COUNT per group equals a bincount,
    SUM(qty) is an integer-valued double and therefore exact, SUM(price*(1-disc)) within 1e-6 of a
    torch float64 reduction over the same device data."""
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    n = 60_000_000
    combo = torch.multinomial(torch.tensor([0.2466, 0.0065, 0.5005, 0.2464], device=dev), n, replacement=True, generator=g)
    k1 = torch.tensor(list(b"ANNR"), dtype=torch.uint8, device=dev)[combo]
    k2 = torch.tensor(list(b"FFOF"), dtype=torch.uint8, device=dev)[combo]
    qty = torch.randint(1, 51, (n,), device=dev, generator=g).double()
    price = (torch.rand(n, device=dev, generator=g, dtype=torch.float64) * 104100 + 900).mul(100).round().div(100)
    disc = torch.randint(0, 11, (n,), device=dev, generator=g).double() / 100
    tax = torch.randint(0, 9, (n,), device=dev, generator=g).double() / 100
    st = capi.AggState(q1_config())
    st.update([k1, k2, qty, price, disc, tax], n)
    keys, vals, nulls, groups = st.finalize(dev)
    gcount = int(groups.item())
    assert gcount == 4
    counts = torch.bincount(combo, minlength=4)
    for row in range(gcount):
        c = [i for i in range(4) if b"ANNR"[i] == int(keys[0][row]) and b"FFOF"[i] == int(keys[1][row])][0]
        sel = combo == c
        assert int(vals[7][row]) == int(counts[c])
        assert float(vals[0][row]) == float(qty[sel].sum())
        ref = (price[sel] * (1 - disc[sel])).sum().item()
        assert abs(float(vals[2][row]) - ref) <= FP_RTOL * abs(ref)
        assert abs(float(vals[6][row]) - disc[sel].mean().item()) <= FP_RTOL


def test_plan_shape_and_interpreter_agree(capi, oracle, dev):
    """The AOT plan-shape kernels (csrc/agg_shapes.hpp) and the interpreter kernel run the same body;
    QSX_AGG_NO_SPECIALIZE=1 (read at state creation) forces the interpreter.  Both must match the oracle
    and each other (integers bit-exact)."""
    import os
    rng = np.random.default_rng(77)
    n = 500_003
    for cfg, cols in ((q1_config(), q1_columns(rng, n)),
                      (T.make_agg_config(T.AGG_COMPACT_KEY, [(T.INT, None), (T.INT, None), (T.DOUBLE, None)], keys=[0, 1],
                                         aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None), (T.AGG_AVG, T.col(2))], est_groups=200),
                       [rng.integers(0, 20, size=n).astype(np.int32), rng.integers(-5, 5, size=n).astype(np.int32),
                        rng.normal(size=n)])):
        results = []
        for no_spec in ("0", "1"):
            os.environ["QSX_AGG_NO_SPECIALIZE"] = no_spec
            try:
                st = run_hip(capi, dev, cfg, cols, blocks=2)
            finally:
                os.environ.pop("QSX_AGG_NO_SPECIALIZE", None)
            results.append(finalize_np(st, dev))
        o = oracle.AggState(cfg)
        o.update(cols, n)
        ref = o.finalize()
        assert_same_groups(results[0], ref)
        assert_same_groups(results[1], ref)
        assert_same_groups(results[0], results[1])


def _family_launches(capi):
    import ctypes
    capi.lib.qsx_debug_agg_family_launches.restype = ctypes.c_longlong
    return capi.lib.qsx_debug_agg_family_launches()


FAMILY_KEYS = {"char": (T.CHAR, 1, np.uint8), "int": (T.INT, None, np.int32), "long": (T.LONG, None, np.int64)}


@pytest.mark.parametrize("keys", [("char",), ("int",), ("long",), ("char", "char"), ("char", "int"), ("int", "char"), ("int", "int")])
@pytest.mark.parametrize("num_sums", [1, 3, 6])
@pytest.mark.parametrize("strategy", [T.AGG_COMPACT_KEY, T.AGG_GENERIC])
def test_plans_of_the_aot_family_never_meet_the_interpreter(capi, oracle, dev, keys, num_sums, strategy, monkeypatch):
    """csrc/agg_family.hpp: a GROUP BY of one or two CHAR(1) / INT / LONG keys (packed into 8 bytes) with one to six SUM / AVG over plain
    DOUBLE columns and COUNT(*) — whatever the order of its columns and aggregates — is served by an ahead-of-time kernel of the
    family on its FIRST update, in a process that may not compile anything (QSX_AGG_JIT=0) and holds no recording of the plan:
    the reference's update loop is one template instantiation whatever the plan (storage/AggregationOperationState.cpp:428-474).
    The plans here put their summed columns IN FRONT of the keys, interleave an unused column, repeat a column under SUM and
    AVG and ask for COUNT(*) twice; results against the oracle, and against the interpreter (QSX_AGG_FAMILY=0)."""
    if strategy == T.AGG_GENERIC and "char" in keys:
        pytest.skip("CHAR group-by keys under the GENERIC strategy are out of scope (DESIGN.md §8)")
    monkeypatch.setenv("QSX_AGG_JIT", "0")
    rng = np.random.default_rng(910 + num_sums + len(keys))
    n = 300_007
    columns, cols, key_idx = [], [], []
    value_idx = []
    for j in range(num_sums):                      # the DOUBLE columns first
        columns.append((T.DOUBLE, None))
        cols.append(rng.normal(size=n) * (j + 1))
        value_idx.append(len(columns) - 1)
    columns.append((T.INT, None))                 # a column the plan never reads
    cols.append(rng.integers(0, 9, size=n).astype(np.int32))
    for k in keys[::-1]:                           # the keys last, in reverse
        ty, width, dtype = FAMILY_KEYS[k]
        columns.append((ty, width))
        if k == "char":
            cols.append(rng.choice(np.frombuffer(b"ABCDEFG", dtype=np.uint8), size=n))
        elif k == "int":
            cols.append(rng.integers(-40, 40, size=n).astype(np.int32))
        else:
            cols.append((rng.integers(-20, 20, size=n) * 1_000_000_007).astype(np.int64))
        key_idx.insert(0, len(columns) - 1)
    aggs = [(T.AGG_COUNT_STAR, None)]
    for j in reversed(range(num_sums)):            # accumulators in an order of their own
        aggs.append((T.AGG_SUM if j % 2 == 0 else T.AGG_AVG, T.col(value_idx[j])))
    aggs.append((T.AGG_AVG, T.col(value_idx[0])))  # the same column again: one accumulator (ReuseAggregateExpressions)
    aggs.append((T.AGG_COUNT_STAR, None))
    if len(aggs) > 8:
        aggs = aggs[:8]
    cfg = T.make_agg_config(strategy, columns, keys=key_idx, aggs=aggs, est_groups=64)
    before = _family_launches(capi)
    st = run_hip(capi, dev, cfg, cols, blocks=2)
    assert _family_launches(capi) == before + 2, "the plan did not reach a kernel of the family"
    got = finalize_np(st, dev)
    o = oracle.AggState(cfg)
    o.update(cols, n)
    assert_same_groups(got, o.finalize())
    monkeypatch.setenv("QSX_AGG_FAMILY", "0")     # (read at state creation)
    st2 = run_hip(capi, dev, cfg, cols, blocks=2)
    assert _family_launches(capi) == before + 2
    assert_same_groups(finalize_np(st2, dev), got)


@pytest.mark.parametrize("form", ["filter", "state_predicate", "filter_and_predicate", "run_of_blocks", "run_of_blocks_with_filters"])
def test_the_aot_family_under_filters_predicates_and_runs_of_blocks(capi, oracle, dev, form, monkeypatch):
    """The family's kernels come with and without a filter bitmap, over one stripe per column and over a run of blocks (the
    operators' work orders: every block its own stripes, the run table numbered canonically); a state's own predicate becomes the
    call's filter by a K1 pass per term (csrc/aggregate.hip family_filter).  No compiler (QSX_AGG_JIT=0); against the oracle."""
    monkeypatch.setenv("QSX_AGG_JIT", "0")
    rng = np.random.default_rng(31)
    n = 400_003
    x, y = rng.normal(size=n), rng.uniform(0, 100, size=n)
    unused = rng.integers(0, 5, size=n).astype(np.int64)
    k1 = rng.integers(-9, 9, size=n).astype(np.int32)
    k0 = rng.choice(np.frombuffer(b"NRA", dtype=np.uint8), size=n)
    d = rng.integers(19920101, 19981231, size=n).astype(np.int32)
    columns = [(T.DOUBLE, None), (T.LONG, None), (T.INT, None), (T.DOUBLE, None), (T.CHAR, 1), (T.INT, None)]
    cols = [x, unused, k1, y, k0, d]
    pred = [(5, T.LE, 19980902), (3, T.GT, 7.5)] if "predicate" in form else []
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, columns, keys=[4, 2], aggs=[(T.AGG_AVG, T.col(3)), (T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(0))],
                            pred=pred, est_groups=128)
    filt = oracle.bitmap_from_bools(rng.random(n) < 0.6) if "filter" in form else None
    before = _family_launches(capi)
    st = capi.AggState(cfg)
    dcols = [to_dev(c, dev) for c in cols]
    o = oracle.AggState(cfg)
    if form.startswith("run_of_blocks"):
        cuts = [0, 1, 4096, 4096, 70_001, 300_000, n]          # a one-row block, an empty block, blocks that end inside a tile
        blocks = [[c[a:b] for c in dcols] for a, b in zip(cuts[:-1], cuts[1:])]
        filters = None
        if form.endswith("filters"):
            masks = [rng.random(b - a) < 0.5 if i % 2 == 0 else None for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))]
            filters = [None if m is None or m.size == 0 else bitmap_dev(oracle.bitmap_from_bools(m), dev) for m in masks]
            keep = np.concatenate([np.ones(b - a, dtype=bool) if m is None else m for m, (a, b) in zip(masks, zip(cuts[:-1], cuts[1:]))])
            o.update(cols, n, filter_bitmap=oracle.bitmap_from_bools(keep))
        else:
            o.update(cols, n)
        st.update_blocks(blocks, filters)
    else:
        st.update(dcols, n, filter_bitmap=None if filt is None else bitmap_dev(filt, dev))
        o.update(cols, n, filter_bitmap=filt)
    assert _family_launches(capi) == before + 1, "the call did not reach a kernel of the family"
    assert_same_groups(finalize_np(st, dev), o.finalize())


def _two_level_updates(capi):
    import ctypes
    capi.lib.qsx_debug_agg_two_level_updates.restype = ctypes.c_longlong
    return capi.lib.qsx_debug_agg_two_level_updates()


@pytest.mark.parametrize("groups,keys_kind", [(1_000_000, "int"), (300_000, "int_and_char"), (1_500_000, "int_clustered"), (8_000_000, "int_underestimated")])
def test_two_level_partitioned_aggregation_for_large_group_counts(capi, oracle, dev, groups, keys_kind, monkeypatch):
    """csrc/agg_pieces.hpp: more groups than one partition pass brings into LDS (est_groups >= 100 K): two K9 passes on
    digits of the mixing hash order the rows by its top 12 bits, then 4096 pieces of disjoint groups go through workgroup-private
    LDS tables (the partitioned aggregation of storage/AggregationOperationState.cpp:548-614 with partition = piece).  COUNT(*),
    SUM over a DOUBLE and a LONG column, AVG, MIN over the LONG, MAX over the DOUBLE, SUM and MAX over an expression (its values
    become a stripe in front of the passes), MIN over the INT column under the state's predicate further down; groups compared one by one with numpy (COUNT and the integer sums exact);
    the same plan through the one-pass path (QSX_AGG_TWO_LEVEL_MIN_GROUPS=0) and through the oracle gives the same groups;
    two update calls, the second one over rows of groups the first has not seen."""
    rng = np.random.default_rng(groups % 1000 + 7)
    n = 17_000_017                                                              # two calls of 8.5 M rows: each one beyond the path's row threshold
    if keys_kind == "int_clustered":
        gid = (np.arange(n, dtype=np.int64) * groups // n)                     # sorted: long runs inside every piece
    else:
        gid = rng.integers(0, groups, size=n)
    a = rng.integers(-1000, 1000, size=n) / 8.0                                 # multiples of 1/8: sums are exact in any order
    b = rng.integers(-2**40, 2**40, size=n).astype(np.int64)
    c = rng.integers(-100, 100, size=n).astype(np.int32)
    if keys_kind == "int_and_char":                                             # (a 40-bit packed code: a LONG next to it would make the key a wide one)
        k0 = ((gid // 5) * 1_003 - 17).astype(np.int32)
        k1 = np.frombuffer(b"VWXYZ", dtype=np.uint8)[gid % 5]
        columns = [(T.INT, None), (T.CHAR, 1), (T.DOUBLE, None), (T.LONG, None), (T.INT, None)]
        cols, key_idx, strategy = [k0, k1, a, b, c], [0, 1], T.AGG_COMPACT_KEY
    else:
        k0 = (gid * 7 - 3).astype(np.int32)
        columns = [(T.INT, None), (T.DOUBLE, None), (T.LONG, None), (T.INT, None)]
        cols, key_idx, strategy = [k0, a, b, c], [0], T.AGG_GENERIC
    v = len(key_idx)
    cfg = T.make_agg_config(strategy, columns, keys=key_idx,
                            aggs=[(T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(v)), (T.AGG_SUM, T.col(v + 1)), (T.AGG_AVG, T.col(v)),
                                  (T.AGG_MIN, T.col(v + 1)), (T.AGG_MAX, T.col(v)),
                                  (T.AGG_SUM, T.temp(1)), (T.AGG_MAX, T.temp(1))],                # over the expression (1 - a) * a
                            instrs=[(T.EX_SUB, 0, T.const(0), T.col(v)), (T.EX_MUL, 1, T.temp(0), T.col(v))], consts=[1.0],
                            est_groups=groups if keys_kind != "int_underestimated" else 3_000_000)   # (1953 groups a piece for tables of 2048 slots: the probe bound sends rows down the global path)
    dcols = [to_dev(x, dev) for x in cols]
    half = n // 2 + 3
    results = []
    for two_level in (True, False):
        monkeypatch.setenv("QSX_AGG_TWO_LEVEL_MIN_GROUPS", "100000" if two_level else "0")
        before = _two_level_updates(capi)
        st = capi.AggState(cfg)
        st.update([x[:half] for x in dcols], half)
        st.update([x[half:] for x in dcols], n - half)
        # (clustered keys: the state samples its leading key and keeps the one-pass path, whose LDS table combines the runs)
        assert _two_level_updates(capi) == before + (2 if two_level and keys_kind != "int_clustered" else 0)
        results.append(finalize_np(st, dev))
        st.close()
    gk, gv, gn = results[0]
    # numpy, by group number
    cnt = np.bincount(gid, minlength=groups)
    present = np.nonzero(cnt)[0]
    sum_a = np.bincount(gid, weights=a, minlength=groups)
    sum_b = np.zeros(groups, dtype=np.int64)
    np.add.at(sum_b, gid, b)
    if keys_kind == "int_and_char":
        got_gid = (gk[0].astype(np.int64) + 17) // 1_003 * 5 + (gk[1].reshape(gk[1].shape[0], -1)[:, 0].astype(np.int64) - ord("V"))
    else:
        got_gid = (gk[0].astype(np.int64) + 3) // 7
    assert got_gid.size == present.size and np.array_equal(np.sort(got_gid), present)
    assert np.array_equal(gv[0], cnt[got_gid])
    assert np.array_equal(gv[1], sum_a[got_gid])
    assert np.array_equal(gv[2], sum_b[got_gid])
    assert np.allclose(gv[3], sum_a[got_gid] / cnt[got_gid], rtol=1e-12, atol=0.0)
    min_b = np.full(groups, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(min_b, gid, b)
    max_a = np.full(groups, -np.inf)
    np.maximum.at(max_a, gid, a)
    assert np.array_equal(gv[4], min_b[got_gid]) and np.array_equal(gv[5], max_a[got_gid])
    expr = (1.0 - a) * a                                                        # (multiples of 1/64: exact, and their sums too)
    sum_e = np.bincount(gid, weights=expr, minlength=groups)
    max_e = np.full(groups, -np.inf)
    np.maximum.at(max_e, gid, expr)
    assert np.array_equal(gv[6], sum_e[got_gid]) and np.array_equal(gv[7], max_e[got_gid])
    assert_same_groups(results[0], results[1])
    # the operators' form: a run of 1 M-row blocks (ragged at the end, an empty one inside) in one call — laid end to end in scratch
    # of the call and through the same partition passes (aggregate.hip update_run_end_to_end)
    import ctypes
    capi.lib.qsx_debug_agg_run_concats.restype = ctypes.c_longlong
    monkeypatch.setenv("QSX_AGG_TWO_LEVEL_MIN_GROUPS", "100000")
    before, concats = _two_level_updates(capi), capi.lib.qsx_debug_agg_run_concats()
    st = capi.AggState(cfg)
    edges = list(range(0, n, 1_000_000)) + [n]
    edges.insert(5, edges[5])
    st.update_blocks([[x[a:b] for x in dcols] for a, b in zip(edges[:-1], edges[1:])])
    assert capi.lib.qsx_debug_agg_run_concats() == concats + 1
    assert _two_level_updates(capi) == before + (1 if keys_kind != "int_clustered" else 0)
    assert_same_groups(finalize_np(st, dev), results[0])
    st.close()
    if keys_kind in ("int", "int_and_char"):
        # under a filter bitmap (a predicate's / LIP filter's TupleIdSequence): many rows survive -> the used columns are compacted
        # under the filter and the survivors take the partition passes (aggregate.hip update_filtered_end_to_end); against the oracle
        capi.lib.qsx_debug_agg_filtered_compactions.restype = ctypes.c_longlong
        keep = rng.random(n) < 0.7
        fbits = oracle.bitmap_from_bools(keep)
        before, compactions = _two_level_updates(capi), capi.lib.qsx_debug_agg_filtered_compactions()
        st = capi.AggState(cfg)
        st.update(dcols, n, filter_bitmap=bitmap_dev(fbits, dev))
        assert capi.lib.qsx_debug_agg_filtered_compactions() == compactions + 1
        assert _two_level_updates(capi) == before + 1
        of = oracle.AggState(cfg)
        of.update(cols, n, filter_bitmap=fbits)
        assert_same_groups(finalize_np(st, dev), of.finalize())
        st.close()
        # the same filter over a run of blocks (an AggregationWorkOrder over a run under its predicate): K2 over the run, then the stripe form
        st = capi.AggState(cfg)
        fedges = list(range(0, n, 2_000_000)) + [n]
        st.update_blocks([[x[a:b] for x in dcols] for a, b in zip(fedges[:-1], fedges[1:])],
                         filters=[bitmap_dev(oracle.bitmap_from_bools(keep[a:b]), dev) for a, b in zip(fedges[:-1], fedges[1:])])
        assert capi.lib.qsx_debug_agg_filtered_compactions() == compactions + 2
        assert_same_groups(finalize_np(st, dev), of.finalize())
        st.close()
        compactions += 1
        # the STATE's own predicate (a conjunction on the DOUBLE and the INT column, with the call's filter on top): a K1 pass per term
        # makes the bitmap, the survivors are compacted and — past their predicate — take the two partition passes
        pcfg = T.make_agg_config(strategy, columns, keys=key_idx,
                                 aggs=[(T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(v)), (T.AGG_SUM, T.col(v + 1)), (T.AGG_MIN, T.col(v + 2))],
                                 pred=[(v, T.GT, -100.0), (v + 2, T.LT, 60)], est_groups=groups)
        for filt in (None, fbits):
            before, comps = _two_level_updates(capi), capi.lib.qsx_debug_agg_filtered_compactions()
            st = capi.AggState(pcfg)
            st.update(dcols, n, filter_bitmap=None if filt is None else bitmap_dev(filt, dev))
            assert capi.lib.qsx_debug_agg_filtered_compactions() == comps + 1 and _two_level_updates(capi) == before + 1
            op = oracle.AggState(pcfg)
            op.update(cols, n, filter_bitmap=filt)
            assert_same_groups(finalize_np(st, dev), op.finalize())
            st.close()
        compactions = capi.lib.qsx_debug_agg_filtered_compactions() - 1
        few = oracle.bitmap_from_bools(rng.random(n) < 0.01)                      # few survivors: compacted too, then the tile kernels on what is left
        st = capi.AggState(cfg)
        st.update(dcols, n, filter_bitmap=bitmap_dev(few, dev))
        assert capi.lib.qsx_debug_agg_filtered_compactions() == compactions + 2
        of = oracle.AggState(cfg)
        of.update(cols, n, filter_bitmap=few)
        assert_same_groups(finalize_np(st, dev), of.finalize())
        st.close()
    o = oracle.AggState(cfg)                                                    # (the oracle's hash table: seconds at these sizes)
    # (a call whose columns would not fit 8 GiB of scratch goes slice by slice; forced here: a child process, the variable is read once)
    if keys_kind == "int":
        import subprocess, sys, os
        code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); import quickstep_amd.capi as capi; from quickstep_amd import types as T;"
                "n, groups = 17_000_017, 1_000_000; g = torch.Generator(device='cuda:0'); g.manual_seed(3);"
                "k = torch.randint(0, groups, (n,), device='cuda:0', generator=g, dtype=torch.int32); v = torch.randint(0, 1000, (n,), device='cuda:0', generator=g).double();"
                "cfg = T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.DOUBLE, None)], keys=[0], aggs=[(T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(1))], est_groups=groups);"
                "st = capi.AggState(cfg); st.update([k, v], n); keys, vals, _, found = st.finalize(torch.device('cuda:0')); f = int(found.item());"
                "wc = torch.bincount(k.long(), minlength=groups); ws = torch.zeros(groups, dtype=torch.float64, device='cuda:0').index_add_(0, k.long(), v);"
                "gk = keys[0][:f].long(); assert f == int((wc > 0).sum()); assert bool((vals[0][:f] == wc[gk]).all()) and bool((vals[1][:f] == ws[gk]).all());"
                "capi.lib.qsx_debug_agg_two_level_updates.restype = __import__('ctypes').c_longlong; assert capi.lib.qsx_debug_agg_two_level_updates() == 4; print('sliced ok')"
                % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, QSX_AGG_TWO_LEVEL_SLICE_ROWS="5000000", QSX_AGG_TWO_LEVEL_MIN_GROUPS="100000"))
        assert r.returncode == 0 and "sliced ok" in r.stdout, r.stderr[-2000:]
    o.update([x[:half] for x in cols], half)
    o.update([x[half:] for x in cols], n - half)
    assert_same_groups(results[0], o.finalize())


def test_plans_outside_the_aot_family_keep_their_kernels(capi, oracle, dev, monkeypatch):
    """An expression under an aggregate, an INT sum, MIN over two keys: not of the family (agg_family.hpp)."""
    monkeypatch.setenv("QSX_AGG_JIT", "0")
    rng = np.random.default_rng(5)
    n = 100_000
    k = rng.integers(0, 30, size=n).astype(np.int32)
    x, y = rng.normal(size=n), rng.normal(size=n)
    z = rng.integers(0, 100, size=n).astype(np.int32)
    plans = [
        T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.INT, None)], keys=[0],
                          instrs=[(T.EX_MUL, 0, T.col(1), T.col(2))], aggs=[(T.AGG_SUM, T.temp(0))], est_groups=64),
        T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.INT, None)], keys=[0],
                          aggs=[(T.AGG_SUM, T.col(3)), (T.AGG_SUM, T.col(1))], est_groups=64),
        T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.INT, None)], keys=[0, 3],
                          aggs=[(T.AGG_MIN, T.col(1))], est_groups=64),
    ]
    before = _family_launches(capi)
    for cfg in plans:
        st = run_hip(capi, dev, cfg, [k, x, y, z])
        o = oracle.AggState(cfg)
        o.update([k, x, y, z], n)
        assert_same_groups(finalize_np(st, dev), o.finalize())
    assert _family_launches(capi) == before


# ---- MIN / MAX (AggregationHandleMin/Max; AggregationOperator_unittest.cpp:675-712, :1550-1680) -----------------
@pytest.mark.parametrize("with_pred", [False, True])
def test_golden_scalar_min_max(capi, dev, golden, with_pred):
    g = golden["agg_unittest"]
    r = agg_rows(g)
    e = g["scalar"]
    cols = [r["i"], r["l"], r["f"], r["d"]]
    aggs = []
    for c in range(4):
        aggs += [(T.AGG_MAX, T.col(c)), (T.AGG_MIN, T.col(c))]
    cfg = T.make_agg_config(T.AGG_SINGLE_STATE, [(T.INT, None), (T.LONG, None), (T.FLOAT, None), (T.DOUBLE, None)], aggs=aggs,
                            pred=[(0, T.LT, e["predicate_less_than"])] if with_pred else [])
    st = run_hip(capi, dev, cfg, cols, blocks=30)
    _, vals, nulls = finalize_np(st, dev)
    mx = e["max_with_predicate"] if with_pred else e["max_no_predicate"]
    assert [v.dtype for v in vals] == [np.int32, np.int32, np.int64, np.int64, np.float32, np.float32, np.float64, np.float64]
    assert vals[0][0] == mx and vals[2][0] == mx and vals[1][0] == e["min"] and vals[3][0] == e["min"]
    assert vals[4][0] == np.float32(0.1 * mx) and vals[6][0] == 0.1 * mx          # exactly the stored values
    assert vals[5][0] == 0.0 and vals[7][0] == 0.0
    assert not any(z[0] for z in nulls)


def test_golden_scalar_min_max_of_expressions_and_zero_rows(capi, dev, golden):
    g = golden["agg_unittest"]
    r = agg_rows(g)
    e = g["scalar"]
    cfg = T.make_agg_config(T.AGG_SINGLE_STATE, [(T.INT, None)],
                            instrs=[(T.EX_ADD, 0, T.col(0), T.col(0)), (T.EX_MUL, 1, T.col(0), T.col(0))],
                            aggs=[(T.AGG_MAX, T.temp(0)), (T.AGG_MAX, T.temp(1)), (T.AGG_MIN, T.temp(0))])
    _, vals, nulls = finalize_np(run_hip(capi, dev, cfg, [r["i"]], blocks=7), dev)
    assert vals[0][0] == e["max_expr_add_no_predicate"] and vals[1][0] == e["max_expr_mul_no_predicate"] and vals[2][0] == 0.0
    # no row passes: MIN / MAX are NULL like SUM (AggregationHandleMin.cpp:100-120)
    cfg = T.make_agg_config(T.AGG_SINGLE_STATE, [(T.INT, None)], aggs=[(T.AGG_MAX, T.col(0)), (T.AGG_MIN, T.col(0)), (T.AGG_COUNT_STAR, None)],
                            pred=[(0, T.LT, e["zero_rows_predicate_less_than"])])
    _, vals, nulls = finalize_np(run_hip(capi, dev, cfg, [r["i"]]), dev)
    assert nulls[0][0] == 1 and nulls[1][0] == 1 and vals[2][0] == 0


@pytest.mark.parametrize("strategy", [T.AGG_COMPACT_KEY, T.AGG_GENERIC])
@pytest.mark.parametrize("with_pred", [False, True])
def test_golden_group_by_min_max(capi, dev, golden, strategy, with_pred):
    g = golden["agg_unittest"]
    r = agg_rows(g)
    e = g["group_by_min_max"]["with_predicate" if with_pred else "without_predicate"]
    cfg = T.make_agg_config(
        strategy, [(T.INT, None), (T.INT, None), (T.INT, None), (T.DOUBLE, None), (T.FLOAT, None), (T.LONG, None)], keys=[0, 1],
        aggs=[(T.AGG_MAX, T.col(2)), (T.AGG_MIN, T.col(2)), (T.AGG_MAX, T.col(3)), (T.AGG_MIN, T.col(3)),
              (T.AGG_MAX, T.col(4)), (T.AGG_MIN, T.col(5)), (T.AGG_SUM, T.col(2))],
        pred=[(2, T.LT, g["group_by_predicate_less_than"])] if with_pred else [], est_groups=20)
    st = run_hip(capi, dev, cfg, [r["gb0"], r["gb1"], r["i"], r["d"], r["f"], r["l"]], blocks=30)
    keys, vals, _ = finalize_np(st, dev)
    order = np.argsort(keys[0] + keys[1] * g["group_by_1_size"])
    mx, mn = np.array(e["max_int_per_group"]), np.array(e["min_int_per_group"])
    assert vals[0][order].tolist() == mx.tolist() and vals[1][order].tolist() == mn.tolist()
    assert np.array_equal(vals[2][order], 0.1 * mx) and np.array_equal(vals[3][order], 0.1 * mn)
    assert np.array_equal(vals[4][order], (0.1 * mx).astype(np.float32)) and vals[5][order].tolist() == mn.tolist()


@pytest.mark.parametrize("strategy,n_groups", [(T.AGG_COMPACT_KEY, 7), (T.AGG_GENERIC, 3000), (T.AGG_COLLISION_FREE, 40_000),
                                                (T.AGG_SINGLE_STATE, 1)])
def test_random_min_max_matches_oracle(capi, oracle, dev, strategy, n_groups):
    """Negative values, -0.0 / +0.0, huge magnitudes, filters, merge of two partial states: MIN/MAX are
    exact for every type (no rounding is involved), so everything compares bit for bit."""
    rng = np.random.default_rng(n_groups)
    n = 300_000
    key = (np.sort(rng.integers(0, n_groups, size=n)) if strategy == T.AGG_COLLISION_FREE else rng.integers(0, n_groups, size=n)).astype(np.int32)
    iv = rng.integers(-2**31, 2**31 - 1, size=n).astype(np.int32)
    lv = rng.integers(-2**62, 2**62, size=n).astype(np.int64)
    fv = rng.normal(size=n).astype(np.float32) * np.float32(1e20)
    dv = rng.normal(size=n) * 1e-300
    dv[::97] = 0.0
    keys = [] if strategy == T.AGG_SINGLE_STATE else [0]
    cfg = T.make_agg_config(strategy, [(T.INT, None), (T.INT, None), (T.LONG, None), (T.FLOAT, None), (T.DOUBLE, None)], keys=keys,
                            instrs=[(T.EX_MUL, 0, T.col(4), T.col(3))],
                            aggs=[(T.AGG_MIN, T.col(1)), (T.AGG_MAX, T.col(1)), (T.AGG_MIN, T.col(2)), (T.AGG_MAX, T.col(2)),
                                  (T.AGG_MIN, T.col(3)), (T.AGG_MAX, T.col(4)), (T.AGG_MAX, T.temp(0)), (T.AGG_SUM, T.col(1))],
                            est_groups=n_groups, num_entries=n_groups)
    cols = [key, iv, lv, fv, dv]
    f = oracle.bitmap_from_bools(rng.random(n) < 0.7)
    half = n // 2
    a = run_hip(capi, dev, cfg, [c[:half] for c in cols], filter_bitmap=oracle.bitmap_from_bools(oracle.bools_from_bitmap(f, n)[:half]))
    b = run_hip(capi, dev, cfg, [c[half:] for c in cols], filter_bitmap=oracle.bitmap_from_bools(oracle.bools_from_bitmap(f, n)[half:]))
    a.import_merge(b.export(dev))
    o = oracle.AggState(cfg)
    o.update(cols, filter_bitmap=f)
    gk, gv, gn = finalize_np(a, dev)
    rk, rv, rn = o.finalize()
    go = np.argsort(gk[0], kind="stable") if gk else np.arange(1)
    ro = np.argsort(rk[0], kind="stable") if rk else np.arange(1)
    if gk:
        assert np.array_equal(gk[0][go], rk[0][ro])
    for x, y in zip(gv, rv):
        assert x.dtype == y.dtype
        assert np.array_equal(x[go], y[ro])


@pytest.mark.parametrize("strategy", [T.AGG_COMPACT_KEY, T.AGG_GENERIC])
@pytest.mark.parametrize("n_groups", [3_000, 40_000])
def test_partitioned_aggregation_for_mid_size_group_counts(capi, oracle, dev, monkeypatch, strategy, n_groups):
    """More groups than a workgroup-private LDS table holds: inputs above QSX_AGG_PARTITION_MIN_ROWS are
    hash-partitioned on the key code first (csrc/aggregate.hip, update_partitioned) and aggregated piece by piece.
    Forced here on 600 k rows; several update calls accumulate into the same state, with MIN/MAX and a predicate."""
    monkeypatch.setenv("QSX_AGG_PARTITION_MIN_ROWS", "0")
    rng = np.random.default_rng(n_groups)
    n = 600_000
    k1 = rng.integers(0, 200, size=n).astype(np.int32)
    k2 = rng.integers(0, n_groups // 200, size=n).astype(np.int32)
    val = rng.normal(size=n) * 1000
    iv = rng.integers(-5, 5, size=n).astype(np.int32)
    cfg = T.make_agg_config(strategy, [(T.INT, None), (T.INT, None), (T.DOUBLE, None), (T.INT, None)], keys=[0, 1],
                            instrs=[(T.EX_MUL, 0, T.col(2), T.col(3))],
                            aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None), (T.AGG_AVG, T.temp(0)), (T.AGG_MIN, T.col(2)),
                                  (T.AGG_MAX, T.col(3)), (T.AGG_SUM, T.col(3))],
                            pred=[(3, T.GE, -4)], est_groups=n_groups)
    cols = [k1, k2, val, iv]
    st = run_hip(capi, dev, cfg, cols, blocks=3)
    o = oracle.AggState(cfg)
    o.update(cols)
    assert_same_groups(finalize_np(st, dev), o.finalize())
    # the same rows without partitioning (hash-range families / global table) agree as well
    monkeypatch.setenv("QSX_AGG_PARTITION_MIN_ROWS", str(1 << 60))
    assert_same_groups(finalize_np(run_hip(capi, dev, cfg, cols, blocks=3), dev), o.finalize())


@pytest.mark.parametrize("key_dtype", [np.int32, np.int64])
def test_existence_map_then_aggregate_q13_shape(capi, oracle, dev, key_dtype):
    """CrossReferenceCoalesceAggregate (ExecutionGenerator.cpp:2054-2210): BuildAggregationExistenceMap over the left
    relation's unique keys, then the right relation aggregated into the same collision-free table.  Left keys
    without right rows finalize with COUNT 0 / SUM 0 / AVG, MIN, MAX NULL; the Q13 histogram follows."""
    rng = np.random.default_rng(13)
    customers, n_orders = 150_000, 1_500_000
    c_custkey = rng.permutation(customers).astype(key_dtype)
    o_custkey = rng.integers(0, customers * 2 // 3, size=n_orders).astype(key_dtype)     # a third has no orders
    o_total = rng.integers(1, 500_000, size=n_orders).astype(np.int64)
    o_price = np.round(rng.uniform(1, 1000, size=n_orders), 2)
    kt = T.INT if key_dtype == np.int32 else T.LONG
    cfg = T.make_agg_config(T.AGG_COLLISION_FREE, [(kt, None), (T.LONG, None), (T.DOUBLE, None)], keys=[0],
                            aggs=[(T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(1)), (T.AGG_AVG, T.col(2)), (T.AGG_MIN, T.col(1)),
                                  (T.AGG_MAX, T.col(2))],
                            pred=[(1, T.GE, 1000)], num_entries=customers)
    keep = rng.random(customers) < 0.9                        # existence map built under a filter, in 3 blocks
    filt = oracle.bitmap_from_bools(keep)
    st = capi.AggState(cfg)
    o = oracle.AggState(cfg)
    st.mark_existence(to_dev(c_custkey, dev), to_dev(filt, dev))
    o.mark_existence(c_custkey, filt)
    assert st.num_groups() == o.num_groups() == int(keep.sum())
    for lo in range(0, n_orders, n_orders // 3):
        hi = min(lo + n_orders // 3, n_orders)
        st.update([to_dev(c[lo:hi], dev) for c in (o_custkey, o_total, o_price)], hi - lo)
    o.update([o_custkey, o_total, o_price])
    got, ref = finalize_np(st, dev), o.finalize()
    assert np.array_equal(got[0][0], ref[0][0])                                   # ascending keys, same set
    assert_same_groups(got, ref)
    counts = got[1][0]
    assert (counts == 0).sum() > 0 and got[2][3][counts == 0].all() and not got[2][3][counts > 0].any()
    # Q13's outer query: customers per order count
    assert np.array_equal(np.bincount(counts), np.bincount(ref[1][0]))


def test_existence_map_rejects_hash_strategies(capi, dev):
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, [(T.INT, None)], keys=[0], aggs=[(T.AGG_COUNT_STAR, None)], est_groups=8)
    with pytest.raises(capi.QsxError) as e:
        capi.AggState(cfg).mark_existence(torch.zeros(4, dtype=torch.int32, device=dev))
    assert e.value.status == T.ERR_UNSUPPORTED


# ---- nullable inputs -----------------------------------------------------------------------------------------------
def test_sql_golden_scalar_aggregates_skip_nulls(capi, oracle, dev, golden):
    """Select.test:609-623 on the device: int_col / double_col are NULL in rows 0, 10, 20 (TestDatabaseLoader.cpp:118-170)."""
    from test_oracle_pins import check_select_scalar_with_nulls, nullable_test_table, select_scalar_with_nulls_config
    cols, nulls = nullable_test_table(oracle, golden)
    st = capi.AggState(select_scalar_with_nulls_config())
    st.update_nullable([to_dev(c, dev) for c in cols], [None if b is None else bitmap_dev(b, dev) for b in nulls])
    _, vals, flags = finalize_np(st, dev)
    check_select_scalar_with_nulls(vals, flags, golden["sql_golden"]["select"]["scalar_with_nulls"])
    all_null = bitmap_dev(oracle.bitmap_from_bools(np.ones(25, dtype=bool)), dev)
    st = capi.AggState(select_scalar_with_nulls_config())
    st.update_nullable([to_dev(c, dev) for c in cols], [all_null, None, None, all_null])
    _, vals, flags = finalize_np(st, dev)
    assert int(vals[0][0]) == 25 and int(vals[5][0]) == 0 and [int(z[0]) for z in flags] == [0, 1, 1, 1, 0, 0]


@pytest.mark.parametrize("kernel", ["interpreter", "plan_shape"])
@pytest.mark.parametrize("strategy", [T.AGG_SINGLE_STATE, T.AGG_COMPACT_KEY, T.AGG_GENERIC, T.AGG_COLLISION_FREE])
def test_nullable_columns_match_the_oracle(capi, oracle, dev, strategy, kernel, monkeypatch):
    """NULL group-by keys drop the tuple, NULL predicate operands fail the predicate, every aggregate skips the tuples whose
    argument (or an operand of its expression) is NULL, COUNT(*) does not; groups whose argument was always NULL finalize
    as NULL.  Blocks with and without bitmaps, with and without a filter — through the interpreter kernel and through the
    state's run-time plan shape (the null bitmaps of a call behind a pointer; compiled at the first update here)."""
    monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", "0" if kernel == "plan_shape" else str(1 << 60))
    rng = np.random.default_rng(77 + strategy)
    n = 150_000
    key = rng.integers(0, 300, size=n).astype(np.int32)
    key[40_064:100_032][key[40_064:100_032] == 7] = 8      # (the block that comes without bitmaps has no row of group 7)
    x = rng.integers(-50, 50, size=n).astype(np.int32)
    y = rng.normal(size=n)
    z = rng.integers(0, 1000, size=n).astype(np.int64)
    p = rng.uniform(size=n).astype(np.float32)
    # group 7 never has a non-NULL x; about a fifth of every nullable column is NULL
    null_key = rng.uniform(size=n) < 0.1
    null_x = (rng.uniform(size=n) < 0.2) | (key == 7)
    null_y = rng.uniform(size=n) < 0.2
    null_p = rng.uniform(size=n) < 0.05
    layout = [(T.INT, None), (T.INT, None), (T.DOUBLE, None), (T.LONG, None), (T.FLOAT, None)]
    keys = [] if strategy == T.AGG_SINGLE_STATE else [0]
    cfg = T.make_agg_config(
        strategy, layout, keys=keys,
        instrs=[(T.EX_MUL, 0, T.col(1), T.col(2)), (T.EX_ADD, 1, T.temp(0), T.col(3))],     # x * y + z
        aggs=[(T.AGG_COUNT_STAR, None), (T.AGG_COUNT, T.col(1)), (T.AGG_SUM, T.col(1)), (T.AGG_AVG, T.col(2)),
              (T.AGG_MIN, T.temp(1)), (T.AGG_MAX, T.col(1)), (T.AGG_SUM, T.col(3))],
        pred=[(4, T.LT, 0.9)], est_groups=512, num_entries=300, nullable=[0, 1, 2, 4])
    cols = [key, x, y, z, p]
    null_bools = [null_key, null_x, null_y, None, null_p]
    keep = rng.uniform(size=n) < 0.7
    edges = [0, 40_000, 40_064, 100_032, n]      # block boundaries at multiples of 64 (bitmaps are sliced by word)
    for with_filter in (False, True):
        st = capi.AggState(cfg)
        o = oracle.AggState(cfg)
        for b in range(len(edges) - 1):
            lo, hi = edges[b], edges[b + 1]
            part = [c[lo:hi] for c in cols]
            # the third block comes without bitmaps (an attribute may hold no NULL in a block)
            bitmaps = [None if (nb is None or b == 2) else oracle.bitmap_from_bools(nb[lo:hi]) for nb in null_bools]
            filt = oracle.bitmap_from_bools(keep[lo:hi]) if with_filter else None
            o.update_nullable(part, bitmaps, filter_bitmap=filt)
            st.update_nullable([to_dev(c, dev) for c in part], [None if m is None else bitmap_dev(m, dev) for m in bitmaps],
                               filter_bitmap=None if filt is None else bitmap_dev(filt, dev))
        got, ref = finalize_np(st, dev), o.finalize()
        assert_same_groups(got, ref)
        # which kernel served the state: 1 = its run-time plan shape, -2 = never asked for one
        assert capi.lib.qsx_debug_agg_jit_state(st._h, 1 if with_filter else 0) == (1 if kernel == "plan_shape" else -2)
        if strategy != T.AGG_SINGLE_STATE:
            row = int(np.nonzero(got[0][0] == 7)[0][0])
            assert [int(z_[row]) for z_ in got[2]] == [0, 0, 1, 0, 1, 1, 0] and int(got[1][1][row]) == 0


def test_bitmap_gather_segmented(capi, oracle, dev):
    """Null bits follow gathered values: bit i = null bit of row tids[i] in its segment, 1 for the outer join's padding."""
    rng = np.random.default_rng(5)
    sizes = [1000, 64, 3001]
    first = [0, 1000, 1064]
    bools = [rng.uniform(size=s) < 0.3 for s in sizes]
    segs = [bitmap_dev(oracle.bitmap_from_bools(bools[0]), dev), None, bitmap_dev(oracle.bitmap_from_bools(bools[2]), dev)]
    whole = np.concatenate([bools[0], np.zeros(64, dtype=bool), bools[2]])
    for count in (0, 1, 63, 64, 65, 5000):
        tids = rng.integers(-1, sum(sizes), size=count).astype(np.int32)
        out = capi.bitmap_gather_segmented(segs, first, to_dev(tids, dev))
        want = oracle.bitmap_from_bools(np.where(tids < 0, True, whole[np.maximum(tids, 0)]))
        assert np.array_equal(out.cpu().numpy().view(np.uint64), want[:(count + 63) // 64])


# ---- mid-size group counts: the group directory ------------------------------------------------------------------------
@pytest.mark.parametrize("directory", ["1", "0"])
def test_thousands_of_groups_directory_and_partitioned_paths(capi, oracle, dev, directory, monkeypatch):
    """Thousands of groups: one accumulator per group in LDS behind the key -> group-number directory (default), or the
    partition pass (QSX_AGG_DIRECTORY=0).  AOT shape (two INT keys SUM/COUNT/AVG), the interpreter with MIN/MAX, a state
    predicate and a filter, and an estimate that the real group count overruns (rows without an accumulator take the
    global path; the table grows)."""
    monkeypatch.setenv("QSX_AGG_DIRECTORY", directory)
    monkeypatch.setenv("QSX_AGG_PARTITION_MIN_ROWS", "100000")
    rng = np.random.default_rng(31)
    n = 2_000_000
    k1 = rng.integers(0, 95, size=n).astype(np.int32)
    k2 = rng.integers(-50, 50, size=n).astype(np.int32)
    val = rng.normal(size=n)
    ival = rng.integers(-1000, 1000, size=n).astype(np.int64)
    layout = [(T.INT, None), (T.INT, None), (T.DOUBLE, None), (T.LONG, None)]
    shape_cfg = T.make_agg_config(T.AGG_COMPACT_KEY, layout[:3], keys=[0, 1],
                                  aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None), (T.AGG_AVG, T.col(2))], est_groups=10_000)
    o = oracle.AggState(shape_cfg)
    o.update([k1, k2, val])
    ref = o.finalize()
    for blocks in (1, 7):
        assert_same_groups(finalize_np(run_hip(capi, dev, shape_cfg, [k1, k2, val], blocks=blocks), dev), ref)
    keep = rng.uniform(size=n) < 0.6
    filt = oracle.bitmap_from_bools(keep)
    for est in (10_000, 2_000):
        cfg = T.make_agg_config(T.AGG_GENERIC, layout, keys=[0, 1],
                                instrs=[(T.EX_MUL, 0, T.col(2), T.col(3))],
                                aggs=[(T.AGG_MIN, T.col(2)), (T.AGG_MAX, T.col(3)), (T.AGG_SUM, T.temp(0)), (T.AGG_COUNT_STAR, None),
                                      (T.AGG_SUM, T.col(3))],
                                pred=[(3, T.GE, -900)], est_groups=est)
        for filter_bitmap in (None, filt):
            o = oracle.AggState(cfg)
            o.update([k1, k2, val, ival], filter_bitmap=filter_bitmap)
            st = run_hip(capi, dev, cfg, [k1, k2, val, ival], filter_bitmap=filter_bitmap)
            st.update([to_dev(c[:1000], dev) for c in (k1, k2, val, ival)], 1000)      # a second, small call on the same state
            o.update([c[:1000] for c in (k1, k2, val, ival)])
            assert_same_groups(finalize_np(st, dev), o.finalize())
            st.clear()                                                                    # the directory starts over with the state
            st.update([to_dev(c, dev) for c in (k1, k2, val, ival)], n)
            o2 = oracle.AggState(cfg)
            o2.update([k1, k2, val, ival])
            assert_same_groups(finalize_np(st, dev), o2.finalize())


@pytest.mark.parametrize("run_time_shape", [False, True])
@pytest.mark.parametrize("layout_kind", ["box", "sparse", "clustered_sample"])
def test_group_directory_numbering_modes(capi, oracle, dev, layout_kind, run_time_shape, monkeypatch):
    """The three ways a row of a mid-size group-by finds its LDS accumulator: position in the key box of the build pass
    (small key ranges), directory lookup (sparse keys: the box has too many cells), and — the build pass only samples a
    large input — neither: groups and keys the sample missed are aggregated through the global table."""
    monkeypatch.setenv("QSX_AGG_DIRECTORY", "1")
    monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", "0" if run_time_shape else str(1 << 60))   # hipRTC shape of the directory kernel / interpreter
    rng = np.random.default_rng(47)
    n = 1_500_000
    k1 = rng.integers(0, 90, size=n).astype(np.int32)
    k2 = rng.integers(-40, 40, size=n).astype(np.int32)
    if layout_kind == "sparse":
        k1 = (k1 * 100_003).astype(np.int32)
        k2 = (k2 * 7_919 - 5).astype(np.int32)
    if layout_kind == "clustered_sample":
        monkeypatch.setenv("QSX_AGG_DIR_SAMPLE_ROWS", "40000")    # every 37th tile of 1024 rows
        order = np.argsort(k1, kind="stable")                      # clustered on k1: the sampled tiles miss whole key ranges
        k1, k2 = k1[order], k2[order]
        k1[-5:] = 5_000_000                                        # and keys far outside any sampled box
    val = rng.normal(size=n)
    layout = [(T.INT, None), (T.INT, None), (T.DOUBLE, None)]
    for strategy, aggs in ((T.AGG_COMPACT_KEY, [(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None), (T.AGG_AVG, T.col(2))]),
                           (T.AGG_GENERIC, [(T.AGG_MIN, T.col(2)), (T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None)])):
        cfg = T.make_agg_config(strategy, layout, keys=[0, 1], aggs=aggs, est_groups=8_000)
        o = oracle.AggState(cfg)
        o.update([k1, k2, val])
        ref = o.finalize()
        for blocks in (1, 3):
            assert_same_groups(finalize_np(run_hip(capi, dev, cfg, [k1, k2, val], blocks=blocks), dev), ref)
        # a run of ragged blocks in one launch pair (qsx_agg_update_blocks: the AOT shape of the COMPACT_KEY configuration,
        # run-time shape or interpreter of the GENERIC one), then more runs into the same state
        dcols = [to_dev(c, dev) for c in (k1, k2, val)]
        cuts = [0, 200_000, 200_000, 777_777, n]
        st = capi.AggState(cfg)
        st.update_blocks([[c[a:b] for c in dcols] for a, b in zip(cuts[:-1], cuts[1:])])
        assert_same_groups(finalize_np(st, dev), ref)
        st = capi.AggState(cfg)
        st.update_blocks([[c[:600_000] for c in dcols]])
        st.update_blocks([[c[600_000:900_000] for c in dcols], [c[900_000:] for c in dcols]])
        assert_same_groups(finalize_np(st, dev), ref)


# ---- group-by keys wider than 8 bytes ---------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", ["long_long", "int_long_int", "double_long_thousands"])
def test_wide_group_keys_match_oracle(capi, oracle, dev, shape, monkeypatch):
    """PackedPayloadHashTable takes any composite key (storage/PackedPayloadHashTable.hpp:499-521); the device packs a key
    of up to 24 bytes into words, groups by a 64-bit hash of them and proves the grouping with MIN / MAX of every word
    (DevConfig::wide_words).  Checked against the oracle's component-wise keys: several blocks, a filter, a predicate, an
    estimate the group count overruns (growth), partitioned finalize, thousands of groups (group directory), the
    interpreter and the run-time plan shape."""
    rng = np.random.default_rng(53)
    n = 400_000
    monkeypatch.setenv("QSX_AGG_PARTITION_MIN_ROWS", "100000")     # thousands of groups, too many accumulators for the group
    if shape == "long_long":                                       # directory: the partition pass takes the single-block updates
        layout = [(T.LONG, None), (T.LONG, None), (T.DOUBLE, None), (T.INT, None)]
        cols = [rng.integers(-3, 4, size=n).astype(np.int64) * (2**40 + 17), rng.integers(0, 9, size=n).astype(np.int64) - 2**62,
                rng.normal(size=n), rng.integers(-100, 100, size=n).astype(np.int32)]
        keys, est = [0, 1], 8
    elif shape == "int_long_int":
        layout = [(T.INT, None), (T.LONG, None), (T.INT, None), (T.DOUBLE, None)]
        cols = [rng.integers(-5, 5, size=n).astype(np.int32), rng.integers(0, 6, size=n).astype(np.int64) * 2**33,
                rng.integers(2**30, 2**30 + 4, size=n).astype(np.int32), rng.normal(size=n)]
        keys, est = [0, 1, 2], 300
    else:
        layout = [(T.DOUBLE, None), (T.LONG, None), (T.DOUBLE, None), (T.INT, None)]
        cols = [rng.integers(0, 70, size=n) * 0.25 - 3.0, rng.integers(0, 60, size=n).astype(np.int64) * 1_000_003,
                rng.normal(size=n), rng.integers(-100, 100, size=n).astype(np.int32)]
        keys, est = [0, 1], 4200
    val = 3 if shape == "int_long_int" else 2
    aggs = [(T.AGG_SUM, T.col(val)), (T.AGG_COUNT_STAR, None), (T.AGG_AVG, T.col(val)), (T.AGG_MIN, T.col(val))]
    keep = oracle.bitmap_from_bools(rng.uniform(size=n) < 0.7)
    for jit in (False, True):
        monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", "0" if jit else str(1 << 60))
        for pred in ([], [(val, T.GT, -1.0)]):
            cfg = T.make_agg_config(T.AGG_GENERIC, layout, keys=keys, aggs=aggs, pred=pred, est_groups=est)
            o = oracle.AggState(cfg)
            o.update(cols)
            for blocks in (1, 3):
                st = run_hip(capi, dev, cfg, cols, blocks=blocks)
                for P in (1, 3):
                    for part in range(P):
                        assert_same_groups(finalize_np(st, dev, part, P), o.finalize(part, P))
            of = oracle.AggState(cfg)
            of.update(cols, filter_bitmap=keep)
            assert_same_groups(finalize_np(run_hip(capi, dev, cfg, cols, filter_bitmap=keep), dev), of.finalize())
    # partial states of two "ranks" merged through their exported images
    cfg = T.make_agg_config(T.AGG_GENERIC, layout, keys=keys, aggs=aggs, est_groups=est)
    a = run_hip(capi, dev, cfg, [c[:n // 2] for c in cols])
    b = run_hip(capi, dev, cfg, [c[n // 2:] for c in cols])
    a.import_merge(b.export(dev))
    o = oracle.AggState(cfg)
    o.update(cols)
    assert_same_groups(finalize_np(a, dev), o.finalize())


def test_wide_group_key_hash_collision_is_reported_not_returned(capi, dev, monkeypatch):
    """Two different wide keys under one hash: finalize must say so (QSX_GROUPS_HASH_COLLISION) instead of returning merged
    groups.  A 5-bit hash (test hook) over 40 keys forces it; with the full hash the same input finalizes normally."""
    rng = np.random.default_rng(59)
    n = 50_000
    cols = [rng.integers(0, 5, size=n).astype(np.int64), rng.integers(0, 8, size=n).astype(np.int64), rng.normal(size=n)]
    cfg = T.make_agg_config(T.AGG_GENERIC, [(T.LONG, None), (T.LONG, None), (T.DOUBLE, None)], keys=[0, 1],
                            aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None)], est_groups=64)
    monkeypatch.setenv("QSX_AGG_WIDE_HASH_BITS", "5")
    with pytest.raises(RuntimeError, match="QSX_GROUPS_HASH_COLLISION"):
        finalize_np(run_hip(capi, dev, cfg, cols), dev)
    monkeypatch.delenv("QSX_AGG_WIDE_HASH_BITS")
    keys, vals, nulls = finalize_np(run_hip(capi, dev, cfg, cols), dev)
    assert keys[0].size == 40


@pytest.mark.parametrize("shape", ["key_box", "looked_up", "three_words"])
def test_wide_group_keys_through_the_group_directory(capi, oracle, dev, shape, monkeypatch):
    """Thousands of groups under a key wider than 8 bytes: the group directory carries the key words in its entries, a row
    gets a group number only when all its key words match, and only the aggregates' own accumulators live in LDS
    (DirView::wide_words).  key_box: component ranges small enough to number the groups by position; looked_up: components
    spread (the bench_ops shape); three_words: a 24-byte key.  A sample that misses groups (global path next to the
    directory), a filter, several blocks, both interpreter and run-time shape; a short hash (collisions) must be reported."""
    rng = np.random.default_rng(61)
    n = 600_000
    k1 = rng.integers(0, 100, size=n).astype(np.int32)
    k2 = rng.integers(0, 90, size=n).astype(np.int64)
    if shape == "key_box":
        layout = [(T.INT, None), (T.LONG, None), (T.INT, None), (T.DOUBLE, None)]
        cols = [k1 - 50, k2 + 2**40, (k1 & 1).astype(np.int32), rng.normal(size=n)]
        keys, val = [0, 1, 2], 3
    elif shape == "looked_up":
        layout = [(T.INT, None), (T.LONG, None), (T.INT, None), (T.DOUBLE, None)]
        cols = [k1, k2 << 33, (k1 & 1).astype(np.int32), rng.normal(size=n)]
        keys, val = [0, 1, 2], 3
    else:
        layout = [(T.LONG, None), (T.LONG, None), (T.LONG, None), (T.DOUBLE, None)]
        cols = [k1.astype(np.int64) * (2**41 + 3), k2 * -(2**35 + 11), (k1 % 3).astype(np.int64) - 2**62, rng.normal(size=n)]
        keys, val = [0, 1, 2], 3
    aggs = [(T.AGG_SUM, T.col(val)), (T.AGG_COUNT_STAR, None), (T.AGG_AVG, T.col(val))]
    keep = oracle.bitmap_from_bools(rng.uniform(size=n) < 0.6)
    for jit in (False, True):
        monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", "0" if jit else str(1 << 60))
        for sample in (None, "20000"):
            if sample is None:
                monkeypatch.delenv("QSX_AGG_DIR_SAMPLE_ROWS", raising=False)
            else:
                monkeypatch.setenv("QSX_AGG_DIR_SAMPLE_ROWS", sample)
            cfg = T.make_agg_config(T.AGG_GENERIC, layout, keys=keys, aggs=aggs, est_groups=9_500)
            o = oracle.AggState(cfg)
            o.update(cols)
            ref = o.finalize()
            for blocks in (1, 3):
                assert_same_groups(finalize_np(run_hip(capi, dev, cfg, cols, blocks=blocks), dev), ref)
            of = oracle.AggState(cfg)
            of.update(cols, filter_bitmap=keep)
            assert_same_groups(finalize_np(run_hip(capi, dev, cfg, cols, filter_bitmap=keep), dev), of.finalize())
            # the same through ONE launch pair over a run of ragged blocks (the directory's two passes walk the run's tiles),
            # with and without per-block filters
            cuts = [0, 70_000, 70_000, 333_333, n]
            dcols = [to_dev(c, dev) for c in cols]
            run = [[c[a:b] for c in dcols] for a, b in zip(cuts[:-1], cuts[1:])]
            st = capi.AggState(cfg)
            st.update_blocks(run)
            assert_same_groups(finalize_np(st, dev), ref)
            keep_bools = oracle.bools_from_bitmap(keep, n) if hasattr(oracle, "bools_from_bitmap") else None
            if keep_bools is not None:
                filters = [bitmap_dev(oracle.bitmap_from_bools(keep_bools[a:b]), dev) if b > a else None for a, b in zip(cuts[:-1], cuts[1:])]
                st = capi.AggState(cfg)
                st.update_blocks(run, filters=filters)
                assert_same_groups(finalize_np(st, dev), of.finalize())
    monkeypatch.delenv("QSX_AGG_DIR_SAMPLE_ROWS", raising=False)
    if shape != "key_box":   # (the key box never consults the hash)
        monkeypatch.setenv("QSX_AGG_WIDE_HASH_BITS", "9")
        cfg = T.make_agg_config(T.AGG_GENERIC, layout, keys=keys, aggs=aggs, est_groups=9_500)
        with pytest.raises(RuntimeError, match="QSX_GROUPS_HASH_COLLISION"):
            finalize_np(run_hip(capi, dev, cfg, cols), dev)


def test_wide_group_key_limits(capi):
    """Every key word costs two accumulators of the state: what does not fit is refused at creation."""
    layout = [(T.LONG, None)] * 4 + [(T.DOUBLE, None)]
    with pytest.raises(capi.QsxError):     # 32 bytes: four words
        capi.AggState(T.make_agg_config(T.AGG_GENERIC, layout, keys=[0, 1, 2, 3], aggs=[(T.AGG_COUNT_STAR, None)], est_groups=8))
    with pytest.raises(capi.QsxError):     # three words + three more accumulators > 8
        capi.AggState(T.make_agg_config(T.AGG_GENERIC, layout, keys=[0, 1, 2],
                                        aggs=[(T.AGG_SUM, T.col(4)), (T.AGG_MIN, T.col(4)), (T.AGG_MAX, T.col(4))], est_groups=8))
    capi.AggState(T.make_agg_config(T.AGG_GENERIC, layout, keys=[0, 1, 2], aggs=[(T.AGG_SUM, T.col(4)), (T.AGG_COUNT_STAR, None)],
                                    est_groups=8)).close()


# ---- DATE columns: predicate and group-by key ---------------------------------------------------------------------------
def test_date_predicate_and_date_group_keys(capi, oracle, dev, monkeypatch):
    """TPC-H Q1's `l_shipdate <= DATE` inside the aggregation state and Q3's GROUP BY l_orderkey, o_orderdate,
    o_shippriority (INT + DATE + INT = 16 bytes: a wide key): DateLit columns with garbage in their padding bytes; the
    padding takes no part in the comparison, the key or the output."""
    rng = np.random.default_rng(61)
    n = 300_000
    years, months, days = rng.integers(1992, 1999, size=n), rng.integers(1, 13, size=n), rng.integers(1, 29, size=n)
    pad = rng.integers(0, 1 << 16, size=n)
    dates = ((years.astype(np.int64) & 0xFFFFFFFF) | (months.astype(np.int64) << 32) | (days.astype(np.int64) << 40) |
             (pad.astype(np.int64) << 48)).astype(np.int64)
    flag = rng.choice(np.frombuffer(b"ANR", dtype=np.uint8), size=n)
    qty = rng.integers(1, 51, size=n).astype(np.float64)
    cutoff = T.date_raw(1998, 9, 2)
    for jit in (False, True):
        monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", "0" if jit else str(1 << 60))
        # Q1 shape: CHAR(1) key, DATE predicate
        cfg = T.make_agg_config(T.AGG_COMPACT_KEY, [(T.CHAR, 1), (T.DOUBLE, None), (T.DATE, None)], keys=[0],
                                aggs=[(T.AGG_SUM, T.col(1)), (T.AGG_COUNT_STAR, None)], pred=[(2, T.LE, cutoff)], est_groups=4)
        o = oracle.AggState(cfg)
        o.update([flag, qty, dates])
        assert_same_groups(finalize_np(run_hip(capi, dev, cfg, [flag, qty, dates], blocks=2), dev), o.finalize())
        # GROUP BY a DATE (month granularity keeps the group count moderate): narrow key (8 bytes) and the Q3-shaped wide key
        month_dates = ((years.astype(np.int64) & 0xFFFFFFFF) | (months.astype(np.int64) << 32) | (np.int64(1) << 40) |
                       (pad.astype(np.int64) << 48)).astype(np.int64)
        okey = rng.integers(0, 40, size=n).astype(np.int32)
        prio = rng.integers(0, 2, size=n).astype(np.int32)
        for keys, layout, cols in (([0], [(T.DATE, None), (T.DOUBLE, None)], [month_dates, qty]),
                                   ([0, 1, 2], [(T.INT, None), (T.DATE, None), (T.INT, None), (T.DOUBLE, None)], [okey, month_dates, prio, qty])):
            cfg = T.make_agg_config(T.AGG_GENERIC, layout, keys=keys, aggs=[(T.AGG_SUM, T.col(len(layout) - 1)), (T.AGG_COUNT_STAR, None)],
                                    est_groups=128)
            o = oracle.AggState(cfg)
            o.update(cols)
            ref = o.finalize()
            got = finalize_np(run_hip(capi, dev, cfg, cols, blocks=3), dev)
            assert_same_groups(got, ref)
            date_key = got[0][keys.index(0) if len(keys) == 1 else 1]
            assert int((date_key.view(np.uint64) >> np.uint64(48)).max()) == 0     # output dates carry zero padding


def test_expression_projection_is_bit_equal_to_the_oracle(capi, oracle, dev):
    """qsx_eval_expression (ScalarBinaryExpression::getAllValues on its own): Q1's charge expression and a mixed-type one;
    every node is rounded on its own in IEEE double on both sides, so the columns are bit-equal."""
    rng = np.random.default_rng(83)
    for n in (1, 63, 100_003):
        price = np.round(rng.uniform(900, 105000, size=n), 2)
        disc = rng.integers(0, 11, size=n) / 100.0
        tax = rng.integers(0, 9, size=n) / 100.0
        qty = rng.integers(1, 51, size=n).astype(np.int32)
        big = rng.integers(-2**40, 2**40, size=n).astype(np.int64)
        f32 = rng.normal(size=n).astype(np.float32)
        cols = [price, disc, tax, qty, big, f32]
        programs = [
            ([(T.EX_SUB, 0, T.const(0), T.col(1)), (T.EX_MUL, 1, T.col(0), T.temp(0)), (T.EX_ADD, 2, T.const(0), T.col(2)),
              (T.EX_MUL, 3, T.temp(1), T.temp(2))], [1.0], T.temp(3)),
            ([(T.EX_DIV, 0, T.col(4), T.col(3)), (T.EX_ADD, 1, T.temp(0), T.col(5)), (T.EX_MUL, 0, T.temp(1), T.const(1))], [0.0, -0.75], T.temp(0)),
            ([], [2.5], T.col(3)),            # a bare attribute: its conversion to DOUBLE
            ([], [2.5], T.const(0)),          # a literal
        ]
        for instrs, consts, result in programs:
            got = capi.eval_expression([to_dev(c, dev) for c in cols], instrs, consts + [0.0] * (T.MAX_CONSTS - len(consts)), result).cpu().numpy()
            want = oracle.eval_expression(cols, instrs, consts + [0.0] * (T.MAX_CONSTS - len(consts)), result)
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


@pytest.mark.parametrize("order", ["clustered", "random", "clustered_then_random"])
def test_lds_table_flushes_under_pressure(capi, oracle, dev, order, monkeypatch):
    """More groups than the workgroup's LDS table holds (est 100 k: 512 slots): the table is written out and restarted when
    rows start missing it — clustered keys (lineitem on l_orderkey: a flushed group never returns) — and the mechanism
    switches itself off where a flush absorbs nothing (random order).  Every accumulator kind, a predicate, a filter, the
    interpreter and the run-time plan shape; multiset of groups against the oracle."""
    rng = np.random.default_rng(97)
    n, groups = 1_500_000, 150_000
    if order == "clustered":
        key = np.sort(rng.integers(0, groups, size=n)).astype(np.int32)
    elif order == "random":
        key = rng.integers(0, groups, size=n).astype(np.int32)
    else:
        key = np.concatenate([np.sort(rng.integers(0, groups, size=n // 2)), rng.integers(0, groups, size=n - n // 2)]).astype(np.int32)
    val = rng.normal(size=n)
    ival = rng.integers(-1000, 1000, size=n).astype(np.int64)
    layout = [(T.INT, None), (T.DOUBLE, None), (T.LONG, None)]
    keep = oracle.bitmap_from_bools(rng.uniform(size=n) < 0.8)
    for jit in (False, True):
        monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", "0" if jit else str(1 << 60))
        cfg = T.make_agg_config(T.AGG_GENERIC, layout, keys=[0],
                                aggs=[(T.AGG_SUM, T.col(1)), (T.AGG_SUM, T.col(2)), (T.AGG_MIN, T.col(1)), (T.AGG_MAX, T.col(2)),
                                      (T.AGG_AVG, T.col(1)), (T.AGG_COUNT_STAR, None)],
                                pred=[(2, T.GE, -900)], est_groups=100_000)
        for filt in (None, keep):
            o = oracle.AggState(cfg)
            o.update([key, val, ival], filter_bitmap=filt)
            assert_same_groups(finalize_np(run_hip(capi, dev, cfg, [key, val, ival], filter_bitmap=filt), dev), o.finalize())
        o = oracle.AggState(cfg)
        o.update([key, val, ival])
        assert_same_groups(finalize_np(run_hip(capi, dev, cfg, [key, val, ival], blocks=4), dev), o.finalize())


# ---- a run of blocks in one launch -------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["q1_aot", "generic_interpreter", "generic_jit", "dense", "hash_ranges"])
def test_update_blocks_equals_one_update_per_block(capi, oracle, dev, kind, monkeypatch):
    """qsx_agg_update_blocks: ONE launch over a run of blocks, every block with its own stripes (the reference's 2-4 MB
    blocks are ~120 K rows: a launch per block is launch-bound by 20x).  Ragged block sizes — empty blocks, single rows,
    sizes that are no multiple of a tile — per-block filters with gaps; AOT shape, interpreter, run-time shape, dense state,
    hash-range families.  Must equal the oracle fed block by block."""
    rng = np.random.default_rng(101)
    sizes = [0, 1, 1023, 1024, 1025, 70_000, 0, 333, 120_000, 512, 5, 48_321]
    n = sum(sizes)
    monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", "0" if kind in ("generic_jit", "dense", "hash_ranges") else str(1 << 60))
    if kind == "q1_aot":
        import bench
        cfg = bench.q1_config()
        cols = bench.gen_q1_columns_cpu(n, 5)
    elif kind == "dense":
        cols = [rng.integers(0, 5000, size=n).astype(np.int32), rng.normal(size=n), rng.integers(-9, 9, size=n).astype(np.int64)]
        cfg = T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.DOUBLE, None), (T.LONG, None)], keys=[0],
                                aggs=[(T.AGG_SUM, T.col(1)), (T.AGG_COUNT_STAR, None), (T.AGG_MAX, T.col(2))], num_entries=5000)
    else:
        groups = 40 if kind != "hash_ranges" else 3000
        cols = [rng.integers(0, groups, size=n).astype(np.int32), rng.normal(size=n), rng.integers(-9, 9, size=n).astype(np.int64)]
        cfg = T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.DOUBLE, None), (T.LONG, None)], keys=[0],
                                instrs=[(T.EX_MUL, 0, T.col(1), T.col(2))],
                                aggs=[(T.AGG_SUM, T.temp(0)), (T.AGG_MIN, T.col(1)), (T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None)],
                                pred=[(2, T.GE, -8)], est_groups=groups)
    edges = np.concatenate([[0], np.cumsum(sizes)])
    host_blocks = [[c[edges[b]:edges[b + 1]] for c in cols] for b in range(len(sizes))]
    dev_blocks = [[to_dev(c, dev) for c in blk] for blk in host_blocks]      # separate allocations per block and column
    o = oracle.AggState(cfg)
    for blk in host_blocks:
        if blk[0].size:
            o.update(blk)
    st = capi.AggState(cfg)
    st.update_blocks(dev_blocks)
    st.update_blocks([])                                                      # an empty run is a no-op
    assert_same_groups(finalize_np(st, dev), o.finalize())
    if kind == "q1_aot":
        return                                                                # (the AOT shape takes no filter)
    # per-block filters; blocks 3 and 8 have none (= every row)
    masks = [rng.uniform(size=s) < 0.6 if s else np.zeros(0, dtype=bool) for s in sizes]
    filters = [None if (b in (3, 8) or sizes[b] == 0) else bitmap_dev(oracle.bitmap_from_bools(masks[b]), dev) for b in range(len(sizes))]
    o = oracle.AggState(cfg)
    for b, blk in enumerate(host_blocks):
        if blk[0].size:
            o.update(blk, filter_bitmap=None if filters[b] is None else oracle.bitmap_from_bools(masks[b]))
    st = capi.AggState(cfg)
    st.update_blocks(dev_blocks, filters=filters)
    st.update_blocks(dev_blocks[:2], filters=filters[:2])                     # a second, tiny run on the same state
    for blk, f, m in list(zip(host_blocks, filters, masks))[:2]:
        if blk[0].size:
            o.update(blk, filter_bitmap=None if f is None else oracle.bitmap_from_bools(m))
    assert_same_groups(finalize_np(st, dev), o.finalize())


def test_partitions_of_a_dense_state_are_finalized_concurrently(capi, oracle, dev):
    """FinalizeAggregationOperator makes one work order per partition of a state and Workers run them at the same time,
    each on its own stream: qsx_agg_num_groups + qsx_agg_finalize(partition p of P) from P host threads must give what one
    thread gives partition by partition (the state used to keep the scratch of these calls: two callers got each other's
    counts)."""
    import threading
    rng = np.random.default_rng(91)
    entries, n, P = 300_000, 400_000, 4
    keys = rng.integers(0, entries, size=n).astype(np.int32)
    vals = rng.integers(0, 1000, size=n).astype(np.int64)
    cfg = T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.LONG, None)], keys=[0], num_entries=entries,
                            aggs=[(T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(1))])
    st = capi.AggState(cfg)
    st.update([to_dev(keys, dev), to_dev(vals, dev)], n)
    torch.cuda.synchronize()
    want_groups = int(np.unique(keys).size)
    want_sum = int(vals.sum())
    streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
    for _ in range(25):
        results, errors = [None] * P, []

        def work(p):
            try:
                g = st.num_groups(stream=streams[p])
                k, v, _, groups = st.finalize(dev, partition=p, num_partitions=P, capacity=g, stream=streams[p])
                streams[p].synchronize()
                m = int(groups.item())
                results[p] = (g, m, k[0][:m].cpu().numpy(), v[0][:m].cpu().numpy(), v[1][:m].cpu().numpy())
            except Exception as exc:  # noqa: BLE001
                errors.append(exc)

        threads = [threading.Thread(target=work, args=(p,)) for p in range(P)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert all(r[0] == want_groups for r in results), [r[0] for r in results]       # every caller sees the whole state's count
        all_keys = np.concatenate([r[2] for r in results])
        assert all_keys.size == want_groups and np.unique(all_keys).size == want_groups
        assert int(sum(r[3].sum() for r in results)) == n and int(sum(r[4].sum() for r in results)) == want_sum


def test_runs_of_blocks_at_scale_equal_one_stripe(capi, dev):
    """C3 shape at 120 M rows cut into 1000 ragged blocks: qsx_agg_update_blocks (one launch), with and without per-block
    filter bitmaps, gives the groups of qsx_agg_update over the same rows as one stripe — COUNT and the integer-valued SUM
    exactly, the others within the floating-point tolerance; select + probe over the same cut agree with their stripe forms."""
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    n = 120_000_000
    combo = torch.multinomial(torch.tensor([0.2466, 0.0065, 0.5005, 0.2464], device=dev), n, replacement=True, generator=g)
    k1 = torch.tensor(list(b"ANNR"), dtype=torch.uint8, device=dev)[combo]
    k2 = torch.tensor(list(b"FFOF"), dtype=torch.uint8, device=dev)[combo]
    qty = torch.randint(1, 51, (n,), device=dev, generator=g).double()
    price = (torch.rand(n, device=dev, generator=g, dtype=torch.float64) * 104100 + 900).mul(100).round().div(100)
    disc = torch.randint(0, 11, (n,), device=dev, generator=g).double() / 100
    tax = torch.randint(0, 9, (n,), device=dev, generator=g).double() / 100
    cols = [k1, k2, qty, price, disc, tax]
    cuts = sorted(set([0, n] + [int(x) // 64 * 64 for x in torch.randint(0, n, (999,), generator=torch.Generator().manual_seed(1)).tolist()]))
    blocks = [[c[a:b] for c in cols] for a, b in zip(cuts[:-1], cuts[1:])]
    bm, cnt = capi.select_cmp(qty, T.LE, 40.0)
    filters = [bm[a // 64:(b + 63) // 64] for a, b in zip(cuts[:-1], cuts[1:])]

    def groups_of(state):
        keys, vals, _, groups = state.finalize(dev, capacity=16)
        k = int(groups.item())
        order = torch.argsort(keys[0][:k].to(torch.int64) * 256 + keys[1][:k].to(torch.int64))
        return [v[:k][order] for v in vals]

    for use_filter in (False, True):
        one, run = capi.AggState(q1_config()), capi.AggState(q1_config())
        one.update(cols, n, filter_bitmap=bm if use_filter else None)
        run.update_blocks(blocks, filters if use_filter else None)
        a, b = groups_of(one), groups_of(run)
        assert torch.equal(a[7], b[7]) and torch.equal(a[0], b[0])            # COUNT(*), SUM(qty): exact
        assert int(a[7].sum().item()) == (int(cnt.item()) if use_filter else n)
        for x, y in zip(a[1:7], b[1:7]):
            assert torch.allclose(x, y, rtol=FP_RTOL, atol=0.0)
    # K1 and the probe over the same cut
    outs, counts = capi.select_cmp_blocks([blk[2] for blk in blocks], T.LE, 40.0)
    assert int(counts.sum().item()) == int(cnt.item())
    assert torch.equal(torch.cat([o[:(blk[2].numel() + 63) // 64] for o, blk in zip(outs, blocks)]), bm[:(n + 63) // 64])
    keys32 = torch.randint(0, 2_000_000, (n,), device=dev, generator=g, dtype=torch.int32)
    table = capi.JoinTable(T.INT, 1_000_000, key_range=(0, 999_999))
    table.build(torch.randperm(1_000_000, device=dev, dtype=torch.int32))
    total = int(table.probe_count(keys32).item())
    assert int(table.probe_count_blocks([keys32[a:b] for a, b in zip(cuts[:-1], cuts[1:])]).item()) == total == int((keys32 < 1_000_000).sum().item())
    p, b_, c = table.probe_blocks([keys32[a:b] for a, b in zip(cuts[:-1], cuts[1:])], capacity=total)
    assert int(c.item()) == total
    assert bool((keys32[p[:total].long()] < 1_000_000).all())                 # run-global probe row numbers
    assert int(torch.unique(p[:total]).numel()) == total                      # every matching row exactly once (unique build keys)
