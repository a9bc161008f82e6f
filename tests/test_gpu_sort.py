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
    }
    second = rng.integers(0, 100, size=n).astype(np.int32)
    for name, col in cols.items():
        for desc in (False, True):
            got = capi.sort_top_k([to_dev(col, dev), to_dev(second, dev)], k, [desc, False]).cpu().numpy()
            want = oracle.sort_permutation([col, second], [desc, False])[:k]
            assert np.array_equal(got, want), (name, desc)
