"""GPU parity for ORDER BY: qsx_sort_permutation against the oracle's stable comparator sort
(SortConfiguration semantics), single and composite keys, ASC/DESC, every key type, ties, Q1/Q3-shaped outputs."""
import numpy as np
import pytest
import torch

from helpers import to_dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 63, 2048, 2049, 100_003])
def test_single_key_every_type_and_direction(capi, oracle, dev, n):
    rng = np.random.default_rng(n)
    cols = {
        "int32": rng.integers(-50, 50, size=n).astype(np.int32),
        "int32 wide": rng.integers(-2**31, 2**31 - 1, size=n).astype(np.int32),
        "int64": rng.integers(-2**62, 2**62, size=n).astype(np.int64),
        "float32": (rng.normal(size=n) * 1e3).astype(np.float32),
        "float64": np.round(rng.normal(size=n), 2),            # many ties
    }
    for name, col in cols.items():
        for desc in (False, True):
            got = capi.sort_permutation([to_dev(col, dev)], [desc]).cpu().numpy()
            want = oracle.sort_permutation([col], [desc])
            assert np.array_equal(got, want), (name, desc)       # stable: identical permutation, not just sorted values


def test_composite_keys_mixed_directions(capi, oracle, dev):
    rng = np.random.default_rng(3)
    n = 250_000
    a = rng.integers(0, 7, size=n).astype(np.int32)
    b = np.round(rng.normal(size=n), 1)
    c = rng.integers(-3, 3, size=n).astype(np.int64)
    d = rng.choice(np.array([0.0, -0.0, 1.5, -1.5], dtype=np.float32), size=n)      # -0.0 and +0.0 tie
    for keys, desc in (([a, b], [False, True]), ([b, a, c], [True, True, False]), ([d, c, a, b], [False, True, False, False])):
        got = capi.sort_permutation([to_dev(k, dev) for k in keys], desc).cpu().numpy()
        want = oracle.sort_permutation(keys, desc)
        assert np.array_equal(got, want)


def test_q3_order_by_revenue_desc_orderdate_limit_10(capi, oracle, dev):
    """ORDER BY revenue DESC, o_orderdate LIMIT 10 over the groups of Q3 (benchmarks/tpch/queries/03.sql): sort, take
    the first 10 row numbers, gather the output columns."""
    rng = np.random.default_rng(4)
    groups = 1_200_000
    revenue = np.round(rng.uniform(1000, 500000, size=groups), 4)
    orderdate = rng.integers(19920101, 19950315, size=groups).astype(np.int32)
    orderkey = rng.permutation(groups).astype(np.int32)
    perm = capi.sort_permutation([to_dev(revenue, dev), to_dev(orderdate, dev)], [True, False])
    top = perm[:10]
    got_key = capi.gather(to_dev(orderkey, dev), top).cpu().numpy()
    got_rev = capi.gather(to_dev(revenue, dev), top).cpu().numpy()
    want = oracle.sort_permutation([revenue, orderdate], [True, False])[:10]
    assert np.array_equal(got_key, orderkey[want]) and np.array_equal(got_rev, revenue[want])
    assert (np.diff(got_rev) <= 0).all()


@pytest.mark.parametrize("n,k", [(10, 3), (10, 50), (70_000, 10), (70_000, 1), (1_500_000, 10), (1_500_000, 20_000), (300_000, 0)])
def test_top_k_is_the_head_of_the_full_sort(capi, oracle, dev, n, k):
    """qsx_sort_top_k = first k rows of the stable sort (SortMergeRunOperator's top_k), for every key type, including
    skewed keys where one histogram bin holds most rows (fallback to the full sort) and heavy ties at the threshold."""
    rng = np.random.default_rng(n + k)
    cols = {
        "uniform double": np.round(rng.uniform(1000, 500000, size=n), 4),
        "skewed int": np.where(rng.random(n) < 0.9, 7, rng.integers(-1000, 1000, size=n)).astype(np.int32),
        "ties int64": rng.integers(0, 5, size=n).astype(np.int64) * (1 << 40),
        "float32 signed": (rng.normal(size=n) * 1e3).astype(np.float32),
        "doubles in [0, 1) (shared leading bits: refined threshold)": rng.random(n),
        "small int64 (leading 40 bits equal)": rng.integers(0, 1 << 20, size=n).astype(np.int64),
    }
    second = rng.integers(0, 100, size=n).astype(np.int32)
    for name, col in cols.items():
        for desc in (False, True):
            got = capi.sort_top_k([to_dev(col, dev), to_dev(second, dev)], k, [desc, False]).cpu().numpy()
            want = oracle.sort_permutation([col, second], [desc, False])[:k]
            assert np.array_equal(got, want), (name, desc)


def test_char_sort_keys(capi, oracle, dev):
    """1-byte CHAR keys (Q1's l_returnflag, l_linestatus) order as unsigned bytes, ASC and DESC, with a numeric tiebreak."""
    rng = np.random.default_rng(12)
    n = 200_000
    flag = rng.choice(np.frombuffer(b"ANR\xf0", dtype=np.uint8), size=n)
    status = rng.choice(np.frombuffer(b"FO", dtype=np.uint8), size=n)
    qty = rng.integers(1, 51, size=n).astype(np.int32)
    for desc in ([False, False, False], [True, False, True]):
        got = capi.sort_permutation([to_dev(flag, dev), to_dev(status, dev), to_dev(qty, dev)], desc).cpu().numpy()
        assert np.array_equal(got, oracle.sort_permutation([flag, status, qty], desc))


@pytest.mark.parametrize("n", [1, 64, 65, 5000, 400_000])
def test_distinct_rows_matches_the_distinctify_table(capi, oracle, dev, n):
    """qsx_distinct_rows = first occurrence of every distinct (group-by..., argument) tuple, in tuple order, with and
    without a filter; tuples over every key type incl. -0.0 / +0.0 (one value) and CHAR."""
    rng = np.random.default_rng(n)
    g1 = rng.integers(0, 3, size=n).astype(np.int32)
    g2 = rng.choice(np.frombuffer(b"AB", dtype=np.uint8), size=n)
    cases = {
        "int arg": [g1, rng.integers(-20, 20, size=n).astype(np.int32)],
        "long arg": [g1, g2, rng.integers(-2**40, 2**40, size=n).astype(np.int64) // (1 << 37)],
        "double arg": [g2, rng.choice(np.array([0.0, -0.0, 0.5, -7.25, 1e300]), size=n)],
        "float alone": [rng.choice(np.array([0.0, -0.0, 2.5, -2.5], dtype=np.float32), size=n)],
        "all distinct": [np.arange(n, dtype=np.int64)[::-1].copy()],
    }
    keep = rng.random(n) < 0.6
    filt = oracle.bitmap_from_bools(keep)
    for name, cols in cases.items():
        dcols = [to_dev(c, dev) for c in cols]
        got = capi.distinct_rows(dcols).cpu().numpy()
        assert np.array_equal(got, oracle.distinct_rows(cols)), name
        got = capi.distinct_rows(dcols, to_dev(filt, dev)).cpu().numpy()
        assert np.array_equal(got, oracle.distinct_rows(cols, filt)), name
    none = oracle.bitmap_from_bools(np.zeros(n, dtype=bool))
    assert capi.distinct_rows([to_dev(g1, dev)], to_dev(none, dev)).numel() == 0


def test_count_distinct_at_scale_q16_shape(capi, dev):
    """COUNT(DISTINCT ps_suppkey) GROUP BY a 3-attribute key (TPC-H Q16 shape) on 20 M rows: distinct tuples fed to a
    COUNT(*) group-by equal a torch unique on the same device data."""
    g = torch.Generator(device=dev)
    g.manual_seed(16)
    n = 20_000_000
    brand = torch.randint(0, 25, (n,), device=dev, generator=g, dtype=torch.int32)
    size = torch.randint(1, 51, (n,), device=dev, generator=g, dtype=torch.int32)
    supp = torch.randint(0, 1000, (n,), device=dev, generator=g, dtype=torch.int32)
    rows = capi.distinct_rows([brand, size, supp])
    packed = (brand.long() * 64 + size.long()) * 1024 + supp.long()
    uniq = torch.unique(packed)
    assert rows.numel() == uniq.numel()
    assert torch.equal(packed[rows.long()], uniq)                      # tuple order = packed order, one row per tuple
    first = torch.full((int(packed.max()) + 1,), n, dtype=torch.int64, device=dev)
    first.scatter_reduce_(0, packed, torch.arange(n, device=dev), reduce="amin")
    assert torch.equal(first[uniq], rows.long())                        # the representative is the first occurrence


def test_date_sort_keys_and_distinct_dates(capi, oracle, dev):
    """ORDER BY revenue DESC, o_orderdate with o_orderdate a real DateLit column (Q3), top-k of the same, and DISTINCT over a
    DATE: ordered by year, month, day whatever the padding bytes hold."""
    from quickstep_amd import types as T
    rng = np.random.default_rng(71)
    n = 120_000
    years = rng.integers(1992, 1999, size=n)
    years[:3] = [-18017, 99999, -99999]
    months, days, pad = rng.integers(1, 13, size=n), rng.integers(1, 29, size=n), rng.integers(0, 1 << 16, size=n)
    dates = ((years.astype(np.int64) & 0xFFFFFFFF) | (months.astype(np.int64) << 32) | (days.astype(np.int64) << 40) |
             (pad.astype(np.int64) << 48)).astype(np.int64)
    revenue = np.round(rng.uniform(0, 50, size=n), 0)            # many ties: the date decides
    for desc in ([False], [True]):
        got = capi.sort_permutation([to_dev(dates, dev)], desc, types=[T.DATE]).cpu().numpy()
        assert np.array_equal(got, oracle.sort_permutation([dates], desc, types=[T.DATE]))
    keys, types, desc = [revenue, dates], [T.DOUBLE, T.DATE], [True, False]
    want = oracle.sort_permutation(keys, desc, types=types)
    assert np.array_equal(capi.sort_permutation([to_dev(k, dev) for k in keys], desc, types=types).cpu().numpy(), want)
    assert np.array_equal(capi.sort_top_k([to_dev(k, dev) for k in keys], 10, desc, types=types).cpu().numpy(), want[:10])
    month_dates = ((years.astype(np.int64) & 0xFFFFFFFF) | (months.astype(np.int64) << 32) | (np.int64(1) << 40) |
                   (pad.astype(np.int64) << 48)).astype(np.int64)
    got = capi.distinct_rows([to_dev(month_dates, dev)], types=[T.DATE]).cpu().numpy()
    assert np.array_equal(got, oracle.distinct_rows([month_dates], types=[T.DATE]))
