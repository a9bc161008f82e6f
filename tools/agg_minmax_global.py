#!/usr/bin/env python3
"""Aggregation through the GLOBAL table (more groups than LDS tables or the group directory hold), 100 M rows, random keys:
a 16-byte key (its hidden MIN / MAX proof: four accumulators next to COUNT and SUM) and a narrow key with MIN + MAX aggregates,
at 10^5 / 10^6 groups.  usage: python tools/agg_minmax_global.py [rows]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("QSX_AGG_JIT_SYNC", "1")
os.environ.setdefault("QSX_AGG_JIT_MIN_ROWS", "0")
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
val = torch.rand(n, device=dev, generator=g, dtype=torch.float64)


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


for groups in (100_000, 1_000_000):
    k = torch.randint(0, groups, (n,), device=dev, generator=g, dtype=torch.int32)
    shapes = {
        "wide_key_sum_count": ([(T.INT, None), (T.LONG, None), (T.DOUBLE, None)], [k, k.long() << 20, val], [0, 1],
                               [(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None)]),
        "narrow_key_sum_count": ([(T.INT, None), (T.DOUBLE, None)], [k, val], [0], [(T.AGG_SUM, T.col(1)), (T.AGG_COUNT_STAR, None)]),
        "narrow_key_min_max_sum": ([(T.INT, None), (T.DOUBLE, None)], [k, val], [0],
                                   [(T.AGG_MIN, T.col(1)), (T.AGG_MAX, T.col(1)), (T.AGG_SUM, T.col(1))]),
    }
    line = {"rows": n, "groups": groups}
    for name, (layout, cols, keys, aggs) in shapes.items():
        cfg = T.make_agg_config(T.AGG_GENERIC, layout, keys=keys, aggs=aggs, est_groups=groups)
        st = capi.AggState(cfg)
        line[name + "_ms"] = round(timed(lambda: st.update(cols, n)), 3)
        st.close()
    print(json.dumps(line), flush=True)
