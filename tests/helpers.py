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
