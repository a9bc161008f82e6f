"""GPU parity for the whole hot path composed as TPC-H Q3 (BASELINE config 5, one GPU's slice):
Select(customer) -> BuildHash + LIP -> Select(orders) + LIP probe + semi probe -> BuildHash + LIP ->
Select(lineitem) + LIP probe -> inner probe -> gather -> CollisionFreeVector GROUP BY l_orderkey
SUM(l_extendedprice * (1 - l_discount)) -> finalize -> join back o_orderdate / o_shippriority.

This is the physical plan the reference's optimizer produces for queries/03.sql (LIP filters by
AttachLIPFilters, group-by reduced to l_orderkey by ReduceGroupByAttributes, CollisionFreeVector by
StarSchemaSimpleCostModel.cpp:614-776).  Every step goes through the C ABI; the expected result is
computed twice: by the oracle's operators composed the same way and by plain numpy set logic."""
import numpy as np
import pytest
import torch

from quickstep_amd import types as T
from helpers import bitmap_dev, bitmap_np, to_dev

pytestmark = pytest.mark.gpu

FP_RTOL = 1e-6          # BASELINE.json north_star: FLOAT/DOUBLE SUM within 1e-6 relative
DATE = 19950315         # dates carried as yyyymmdd integers (DateLit compares the same way)
BUILDING = 1


def make_q3_tables(scale_orders, seed):
    rng = np.random.default_rng(seed)
    n_c, n_o = scale_orders // 10, scale_orders
    c_custkey = rng.permutation(n_c).astype(np.int32) + 1
    c_mktsegment = rng.integers(0, 5, size=n_c).astype(np.int32)
    o_orderkey = (rng.permutation(n_o).astype(np.int32) + 1)
    o_custkey = rng.integers(1, n_c + 1, size=n_o).astype(np.int32)
    o_orderdate = rng.choice(np.array([19950101, 19950210, 19950314, 19950315, 19950401, 19960101]), size=n_o).astype(np.int32)
    o_shippriority = rng.integers(0, 3, size=n_o).astype(np.int32)
    lines = rng.integers(1, 8, size=n_o)
    l_orderkey = np.repeat(np.sort(o_orderkey), lines[np.argsort(o_orderkey)]).astype(np.int32)   # clustered on orderkey
    n_l = l_orderkey.size
    l_extendedprice = np.round(rng.uniform(900, 105000, size=n_l), 2)
    l_discount = rng.integers(0, 11, size=n_l) / 100.0
    l_shipdate = rng.choice(np.array([19950101, 19950315, 19950316, 19950620]), size=n_l).astype(np.int32)
    return dict(c_custkey=c_custkey, c_mktsegment=c_mktsegment, o_orderkey=o_orderkey, o_custkey=o_custkey,
                o_orderdate=o_orderdate, o_shippriority=o_shippriority, l_orderkey=l_orderkey,
                l_extendedprice=l_extendedprice, l_discount=l_discount, l_shipdate=l_shipdate)


def q3_agg_config(num_entries):
    # columns: l_orderkey, l_extendedprice, l_discount ; temp1 = price * (1 - disc)
    return T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.DOUBLE, None), (T.DOUBLE, None)], keys=[0],
                             instrs=[(T.EX_SUB, 0, T.const(0), T.col(2)), (T.EX_MUL, 1, T.col(1), T.temp(0))],
                             consts=[1.0], aggs=[(T.AGG_SUM, T.temp(1))], num_entries=num_entries)


def q3_on_gpu(capi, dev, t, dense_tables):
    d = {k: to_dev(v, dev) for k, v in t.items()}
    n_c, n_o, n_l = t["c_custkey"].size, t["o_orderkey"].size, t["l_orderkey"].size
    # customer: c_mktsegment = 'BUILDING' -> hash table + exact LIP filter on c_custkey
    c_sel, _ = capi.select_cmp(d["c_mktsegment"], T.EQ, BUILDING)
    t_c = capi.JoinTable(T.INT, n_c, key_range=(1, n_c) if dense_tables else None)
    t_c.build(d["c_custkey"], filter_bitmap=c_sel)
    lip_c = capi.LipFilter(T.LIP_BITVECTOR_EXACT, n_c, 1)
    lip_c.build(d["c_custkey"], filter_bitmap=c_sel)
    # orders: o_orderdate < DATE, LIP(c_custkey), customer must exist (no customer column is projected)
    o_sel, _ = capi.select_cmp(d["o_orderdate"], T.LT, DATE)
    o_lip, _ = lip_c.probe(d["o_custkey"], in_bitmap=o_sel)
    o_ok, o_cnt = t_c.probe_exists(d["o_custkey"], filter_bitmap=o_lip)
    t_o = capi.JoinTable(T.INT, int(o_cnt.item()), key_range=(1, n_o) if dense_tables else None)
    t_o.build(d["o_orderkey"], filter_bitmap=o_ok)
    lip_o = capi.LipFilter(T.LIP_BITVECTOR_EXACT, n_o, 1)
    lip_o.build(d["o_orderkey"], filter_bitmap=o_ok)
    # lineitem: l_shipdate > DATE, LIP(o_orderkey), inner probe
    l_sel, _ = capi.select_cmp(d["l_shipdate"], T.GT, DATE)
    l_lip, _ = lip_o.probe(d["l_orderkey"], in_bitmap=l_sel)
    total = int(t_o.probe_count(d["l_orderkey"], filter_bitmap=l_lip).item())
    p, b, cnt = t_o.probe(d["l_orderkey"], capacity=total, filter_bitmap=l_lip)
    assert int(cnt.item()) == total
    p, b = p[:total], b[:total]
    key = capi.gather(d["l_orderkey"], p)
    price = capi.gather(d["l_extendedprice"], p)
    disc = capi.gather(d["l_discount"], p)
    # group by l_orderkey (dense, max key + 1 entries)
    st = capi.AggState(q3_agg_config(n_o + 1))
    st.update([key, price, disc], total)
    keys, vals, nulls, groups = st.finalize(dev)
    g = int(groups.item())
    out_key, revenue = keys[0][:g], vals[0][:g]
    # o_orderdate, o_shippriority of every group: look the order up again
    gp, gb, gcnt = t_o.probe(out_key, capacity=g)
    assert int(gcnt.item()) == g
    order = torch.argsort(gp[:g])                                   # pair order is unspecified
    rows = gb[:g][order]
    odate = capi.gather(d["o_orderdate"], rows)
    oprio = capi.gather(d["o_shippriority"], rows)
    return (out_key.cpu().numpy(), revenue.cpu().numpy(), odate.cpu().numpy(), oprio.cpu().numpy(),
            dict(customers=int(capi.bitmap_count(c_sel, n_c).item()), orders=int(o_cnt.item()), pairs=total))


def q3_with_oracle(oracle, t):
    n_c, n_o = t["c_custkey"].size, t["o_orderkey"].size
    c_sel = oracle.select_cmp(t["c_mktsegment"], T.EQ, BUILDING)
    t_c = oracle.JoinTable(T.INT, n_c)
    t_c.build(t["c_custkey"], filter_bitmap=c_sel)
    lip_c = oracle.LipFilter(T.LIP_BITVECTOR_EXACT, n_c, 1)
    lip_c.build(t["c_custkey"], filter_bitmap=c_sel)
    o_sel = oracle.select_cmp(t["o_orderdate"], T.LT, DATE)
    o_ok = t_c.probe_exists(t["o_custkey"], filter_bitmap=lip_c.probe(t["o_custkey"], in_bitmap=o_sel))
    t_o = oracle.JoinTable(T.INT, n_o)
    t_o.build(t["o_orderkey"], filter_bitmap=o_ok)
    lip_o = oracle.LipFilter(T.LIP_BITVECTOR_EXACT, n_o, 1)
    lip_o.build(t["o_orderkey"], filter_bitmap=o_ok)
    l_sel = oracle.select_cmp(t["l_shipdate"], T.GT, DATE)
    p, b = t_o.probe(t["l_orderkey"], filter_bitmap=lip_o.probe(t["l_orderkey"], in_bitmap=l_sel))
    st = oracle.AggState(q3_agg_config(n_o + 1))
    st.update([oracle.gather(t["l_orderkey"], p), oracle.gather(t["l_extendedprice"], p), oracle.gather(t["l_discount"], p)])
    keys, vals, _ = st.finalize()
    return keys[0], vals[0], o_ok, p.size


@pytest.mark.parametrize("dense_tables", [False, True])
@pytest.mark.parametrize("scale_orders,seed", [(2_000, 1), (150_000, 2)])
def test_q3_pipeline_matches_oracle_and_numpy(capi, oracle, dev, scale_orders, seed, dense_tables):
    t = make_q3_tables(scale_orders, seed)
    key, revenue, odate, oprio, stats = q3_on_gpu(capi, dev, t, dense_tables)
    rkey, rrev, o_ok, rpairs = q3_with_oracle(oracle, t)
    assert stats["pairs"] == rpairs and stats["orders"] == oracle.bitmap_count(o_ok, t["o_orderkey"].size)
    assert np.array_equal(key, rkey)                                  # ascending l_orderkey, bit-exact
    assert np.allclose(revenue, rrev, rtol=FP_RTOL, atol=0.0)
    # independent statement of the query in numpy
    cust_ok = np.zeros(t["c_custkey"].size + 1, dtype=bool)
    cust_ok[t["c_custkey"][t["c_mktsegment"] == BUILDING]] = True
    ord_ok = (t["o_orderdate"] < DATE) & cust_ok[t["o_custkey"]]
    by_key = np.zeros(t["o_orderkey"].size + 1, dtype=np.int64) - 1
    by_key[t["o_orderkey"][ord_ok]] = np.nonzero(ord_ok)[0]
    li_ok = (t["l_shipdate"] > DATE) & (by_key[t["l_orderkey"]] >= 0)
    want = np.zeros(t["o_orderkey"].size + 1)
    np.add.at(want, t["l_orderkey"][li_ok], t["l_extendedprice"][li_ok] * (1.0 - t["l_discount"][li_ok]))
    want_keys = np.unique(t["l_orderkey"][li_ok])
    assert np.array_equal(key, want_keys)
    assert np.allclose(revenue, want[want_keys], rtol=FP_RTOL, atol=0.0)
    assert np.array_equal(odate, t["o_orderdate"][by_key[want_keys]])
    assert np.array_equal(oprio, t["o_shippriority"][by_key[want_keys]])
    assert stats["customers"] == int((t["c_mktsegment"] == BUILDING).sum())
