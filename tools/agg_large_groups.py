#!/usr/bin/env python3
"""Hash aggregation with MANY groups (the global-table path: every row costs NS + 1 global atomics): GENERIC, one INT or
LONG key, SUM(double) + COUNT(*), rows x groups.  usage: python tools/agg_large_groups.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("QSX_AGG_JIT_SYNC", "1")
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)
n = 100_000_000
val = torch.rand(n, device=dev, generator=g, dtype=torch.float64)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


only_random_generic = len(sys.argv) > 1 and sys.argv[1] == "random-generic"   # (the profile of the two-level path's kernels)
for groups in (100_000, 300_000, 1_000_000, 3_000_000, 10_000_000) if not only_random_generic else (1_000_000, 10_000_000):
    for label, keys in (("random keys", torch.randint(0, groups, (n,), device=dev, generator=g, dtype=torch.int32)),
                        ("clustered keys (sorted)", (torch.arange(n, device=dev, dtype=torch.int64) * groups // n).to(torch.int32))):
        if only_random_generic and label != "random keys":
            continue
        for strategy, name in ((T.AGG_GENERIC, "GENERIC (hash table)"), (T.AGG_COLLISION_FREE, "COLLISION_FREE (dense arrays)"))[:1 if only_random_generic else 2]:
            cfg = T.make_agg_config(strategy, [(T.INT, None), (T.DOUBLE, None)], keys=[0], aggs=[(T.AGG_SUM, T.col(1)), (T.AGG_COUNT_STAR, None)],
                                    est_groups=groups, num_entries=groups)
            st = capi.AggState(cfg)

            def run():
                st.clear()
                st.update([keys, val], n)
            ms = timed(run)
            print(json.dumps({"rows": n, "groups": groups, "keys": label, "strategy": name, "found": st.num_groups(), "ms": round(ms, 3),
                              "G_rows_per_s": round(n / ms / 1e6, 2)}), flush=True)
            del st
