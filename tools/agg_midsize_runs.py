#!/usr/bin/env python3
"""A mid-size group-by (two INT keys, 10 k groups, SUM + COUNT) over a RUN of blocks (qsx_agg_update_blocks, 200 M rows as
1600 blocks of 125 K rows): the group directory's two passes over the run against the hash-range families the run form
used before (QSX_AGG_DIRECTORY=0), and the one-stripe call for reference.  usage: python tools/agg_midsize_runs.py [rows]"""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
k1 = torch.randint(0, 100, (n,), device=dev, generator=g, dtype=torch.int32)
k2 = torch.randint(0, 100, (n,), device=dev, generator=g, dtype=torch.int32)
val = torch.rand(n, device=dev, generator=g, dtype=torch.float64)
rows = 125_000
blocks = [[c[a:a + rows] for c in (k1, k2, val)] for a in range(0, n, rows)]


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps, 3)


line = {"rows": n, "blocks": len(blocks), "groups": 10_000}
for strategy, name in ((T.AGG_GENERIC, "generic"), (T.AGG_COMPACT_KEY, "compact_key_aot_shape")):
    aggs = [(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None)] + ([(T.AGG_AVG, T.col(2))] if strategy == T.AGG_COMPACT_KEY else [])
    cfg = T.make_agg_config(strategy, [(T.INT, None), (T.INT, None), (T.DOUBLE, None)], keys=[0, 1], aggs=aggs, est_groups=10_000)
    for directory in ("1", "0"):
        os.environ["QSX_AGG_DIRECTORY"] = directory
        st = capi.AggState(cfg)
        tag = "directory" if directory == "1" else "hash_range_families"
        line[f"{name}_run_of_blocks_ms_{tag}"] = timed(lambda: st.update_blocks(blocks))
        if directory == "1":
            line[f"{name}_one_stripe_ms"] = timed(lambda: st.update([k1, k2, val], n))
        st.close()
print(json.dumps(line))
