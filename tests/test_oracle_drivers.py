"""The CPU baseline drivers of the oracle (oracle/qsx_oracle.cpp: qso_bench_agg_coded, qso_bench_partitioned_join,
qso_bench_q3 — what bench.py's `secondary.*.cpu_baseline` times) against numpy restatements of the same queries.
Test infrastructure checking test infrastructure: no GPU, no product code."""
import numpy as np
import pytest

from oracle import pyoracle as O
from quickstep_amd import types as T


def _c4_inputs(n_o, seed):
    rng = np.random.default_rng(seed)
    o_key = (rng.permutation(n_o) + 1).astype(np.int32)
    lines = rng.integers(1, 8, size=n_o)
    l_key = np.repeat(np.arange(1, n_o + 1, dtype=np.int32), lines)
    line_no = np.arange(l_key.size) - np.repeat(np.cumsum(lines) - lines, lines)
    return o_key, o_key.astype(np.int64) * 3 + 1, l_key, l_key.astype(np.int64) * 5 + line_no


@pytest.mark.parametrize("partitions,threads,block_rows", [(8, 4, 1000), (4, 1, 257), (1, 3, 100_000), (3, 2, 64)])
def test_partitioned_join_driver_produces_every_joined_row_once(partitions, threads, block_rows):
    o_key, o_pay, l_key, l_pay = _c4_inputs(20_000, 5)
    r = O.bench_partitioned_join(o_key, o_pay, l_key, l_pay, partitions, block_rows, threads)
    assert r["violations"] == 0
    assert r["output_rows"] == l_key.size            # every lineitem has exactly one order
    # order-independent checksum over (o_payload, l_payload, key) of the rows the join must produce
    want = (l_key.astype(np.uint64) * np.uint64(3) + np.uint64(1)) * np.uint64(1000003) + l_pay.astype(np.uint64) + l_key.astype(np.uint64)
    assert r["checksum"] == int(want.sum(dtype=np.uint64))


def test_partitioned_join_driver_drops_lineitems_without_an_order():
    o_key, o_pay, l_key, l_pay = _c4_inputs(5_000, 6)
    keep = o_key % 3 != 0
    r = O.bench_partitioned_join(o_key[keep], o_pay[keep], l_key, l_pay, 8, 512, 4)
    assert r["violations"] == 0 and r["output_rows"] == int((l_key % 3 != 0).sum())


def _q3_inputs(sf, seed):
    rng = np.random.default_rng(seed)
    n_c, n_o = int(150_000 * sf), int(1_500_000 * sf)
    lines = rng.integers(1, 8, size=n_o)
    l_orderkey = np.repeat(np.arange(1, n_o + 1, dtype=np.int32), lines)
    n_l = l_orderkey.size
    return {"c_custkey": (rng.permutation(n_c) + 1).astype(np.int32), "c_mktsegment": rng.integers(0, 5, size=n_c).astype(np.int32),
            "o_orderkey": (rng.permutation(n_o) + 1).astype(np.int32), "o_custkey": rng.integers(1, n_c + 1, size=n_o).astype(np.int32),
            "o_orderdate": rng.integers(19920101, 19981231, size=n_o).astype(np.int32),
            "l_orderkey": l_orderkey, "l_extendedprice": rng.uniform(900, 105000, size=n_l),
            "l_discount": rng.integers(0, 11, size=n_l) / 100.0, "l_shipdate": rng.integers(19920101, 19981231, size=n_l).astype(np.int32),
            "customers_total": n_c, "orders_total": n_o}


@pytest.mark.parametrize("threads,block_rows", [(1, 4096), (4, 1000), (3, 50_000)])
def test_q3_driver_matches_numpy(threads, block_rows):
    from helpers import q3_reference_numpy
    inp = _q3_inputs(0.02, 11)
    keys, sums, pairs = q3_reference_numpy([inp])
    r = O.bench_q3(inp, 1, 19950315, block_rows, threads)
    assert r["pairs"] == pairs and r["groups"] == keys.size
    order = np.argsort(-sums, kind="stable")[:10]
    np.testing.assert_allclose(r["top_revenue"][:order.size], sums[order], rtol=1e-9)
    assert r["top_orderkey"][:order.size] == keys[order].tolist()


def test_coded_aggregation_driver_equals_the_plain_one():
    rng = np.random.default_rng(3)
    n = 50_000
    k1 = rng.choice(np.frombuffer(b"ANR", dtype=np.uint8), size=n)
    k2 = rng.choice(np.frombuffer(b"FO", dtype=np.uint8), size=n)
    qty_c, disc_c, tax_c = (rng.integers(0, m, size=n).astype(np.uint8) for m in (50, 11, 9))
    price = np.round(rng.uniform(900, 105000, size=n), 2)
    qty_d, disc_d, tax_d = np.arange(1, 51, dtype=np.float64), np.arange(11) / 100.0, np.arange(9) / 100.0

    def config(coded):
        cfg = T.make_agg_config(
            T.AGG_COMPACT_KEY,
            columns=[(T.CHAR, 1), (T.CHAR, 1), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None)], keys=[0, 1],
            instrs=[(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0)),
                    (T.EX_ADD, 2, T.const(0), T.col(5)), (T.EX_MUL, 3, T.temp(1), T.temp(2))], consts=[1.0],
            aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.temp(1)), (T.AGG_SUM, T.temp(3)), (T.AGG_AVG, T.col(4)), (T.AGG_COUNT_STAR, None)],
            est_groups=6)
        if coded:
            for c in (2, 4, 5):
                cfg.column_code_width[c] = 1
        return cfg
    _, coded = O.bench_agg_coded(config(True), [k1, k2, qty_c, price, disc_c, tax_c], [None, None, qty_d, None, disc_d, tax_d], n, 4096, 3)
    _, plain = O.bench_agg(config(False), [k1, k2, qty_d[qty_c], price, disc_d[disc_c], tax_d[tax_c]], n, 4096, 3)
    ck, cv, _ = coded.finalize()
    pk, pv, _ = plain.finalize()
    oc, op = np.lexsort([ck[1], ck[0]]), np.lexsort([pk[1], pk[0]])
    assert np.array_equal(ck[0][oc], pk[0][op]) and np.array_equal(ck[1][oc], pk[1][op])
    for a in range(5):
        if cv[a].dtype == np.int64:
            assert np.array_equal(cv[a][oc], pv[a][op])
        else:
            np.testing.assert_allclose(cv[a][oc], pv[a][op], rtol=1e-9)
