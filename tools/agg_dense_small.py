#!/usr/bin/env python3
"""COLLISION_FREE (dense arrays) against GENERIC over the same small key ranges, random keys: 100 M rows, one INT key,
SUM(double) + COUNT(*).  usage: python tools/agg_dense_small.py [rows]"""
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


for groups in (5, 25, 1000, 8000, 16_000, 30_000, 60_000, 100_000):
    keys = torch.randint(0, groups, (n,), device=dev, generator=g, dtype=torch.int32)
    line = {"rows": n, "entries": groups}
    for strategy, name in ((T.AGG_GENERIC, "generic_ms"), (T.AGG_COLLISION_FREE, "collision_free_ms")):
        cfg = T.make_agg_config(strategy, [(T.INT, None), (T.DOUBLE, None)], keys=[0], aggs=[(T.AGG_SUM, T.col(1)), (T.AGG_COUNT_STAR, None)],
                                est_groups=groups, num_entries=groups)
        st = capi.AggState(cfg)

        def run():
            st.clear()
            st.update([keys, val], n)
        line[name] = round(timed(run), 3)
        line[name.replace("_ms", "_groups")] = st.num_groups()
        st.close()
    print(json.dumps(line), flush=True)
