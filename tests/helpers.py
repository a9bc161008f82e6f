"""Shared helpers of the gpu parity tests."""
import numpy as np
import torch


def to_dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def bitmap_np(t):
    return t.cpu().numpy().view(np.uint64)


def bitmap_dev(words, dev):
    return torch.from_numpy(words.view(np.int64).copy()).to(dev)


def sorted_pairs(p, b):
    a = np.stack([np.asarray(p, dtype=np.int64), np.asarray(b, dtype=np.int64)], 1)
    return a[np.lexsort((a[:, 1], a[:, 0]))]


def rows_sorted(cols):
    """Sort rows of a list of equally long 1-D arrays lexicographically (first column major)."""
    order = np.lexsort([np.asarray(c) for c in cols[::-1]])
    return [np.asarray(c)[order] for c in cols]


def q3_reference_numpy(inputs_per_rank):
    """The same query in numpy over the concatenated shares (the checker of the tests; small sizes only)."""
    import numpy as np
    cat = lambda k: np.concatenate([i[k] for i in inputs_per_rank])  # noqa: E731
    c_ok = set(cat("c_custkey")[cat("c_mktsegment") == 1].tolist())
    o_key, o_cust, o_date = cat("o_orderkey"), cat("o_custkey"), cat("o_orderdate")
    o_sel = (o_date < 19950315) & np.isin(o_cust, np.fromiter(c_ok, dtype=np.int64, count=len(c_ok)))
    ok_orders = o_key[o_sel]
    l_key, l_price, l_disc, l_ship = cat("l_orderkey"), cat("l_extendedprice"), cat("l_discount"), cat("l_shipdate")
    l_sel = (l_ship > 19950315) & np.isin(l_key, ok_orders)
    rev = l_price[l_sel] * (1.0 - l_disc[l_sel])
    keys, inverse = np.unique(l_key[l_sel], return_inverse=True)
    sums = np.bincount(inverse, weights=rev, minlength=keys.size)
    return keys, sums, int(l_sel.sum())
