#!/usr/bin/env python3
"""Q1 aggregation under a filter bitmap: one stripe (qsx_agg_update) against the same rows as a run of blocks with per-block
bitmaps (qsx_agg_update_blocks).  usage: python tools/agg_runs_filter_time.py [rows] [blocks]"""
import json
import os
import sys

import torch

os.environ.setdefault("QSX_AGG_JIT_MIN_ROWS", "0")   # run-time plan shapes compiled at first use (a filter variant has no AOT shape)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600_000_000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 150
dev = torch.device("cuda:0")
cols = bench.gen_q1_columns_gpu(n, dev, 4)
bm, cnt = capi.select_cmp(cols[2], T.LE, 49.0)          # ~98 % of the rows, like Q1's shipdate predicate
st = capi.AggState(bench.q1_config())
block = (n // nb + 63) // 64 * 64                        # block boundaries on bitmap words
starts = list(range(0, n, block))
blocks = [[c[s:min(n, s + block)] for c in cols] for s in starts]
filters = [bm[s // 64:(min(n, s + block) + 63) // 64] for s in starts]


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def one():
    st.clear()
    st.update(cols, n, filter_bitmap=bm)


def run():
    st.clear()
    st.update_blocks(blocks, filters)


def run_no_filter():
    st.clear()
    st.update_blocks(blocks)


def one_no_filter():
    st.clear()
    st.update(cols, n)


res = {"rows": n, "blocks": len(starts), "selected": int(cnt.item())}
for name, fn in (("one stripe, no filter", one_no_filter), ("run of blocks, no filter", run_no_filter), ("one stripe under a filter", one),
                 ("run of blocks under per-block filters", run)):
    res[name + ": ms"] = round(timed(fn), 3)
    if "filter" in name:
        keys, vals, _, groups = st.finalize(dev, capacity=16)
        res.setdefault("count(*) by plan", []).append(int(vals[7][:int(groups.item())].sum().item()))
print(json.dumps(res))
