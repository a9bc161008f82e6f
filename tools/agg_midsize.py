#!/usr/bin/env python3
"""Mid-size group counts (SURVEY §8d minimal variant: two INT keys + one DOUBLE, up to 10 k groups): group directory
vs partition pass, by group count.  usage: python tools/agg_midsize.py [rows]"""
import json
import os
import sys

import torch

os.environ.setdefault("QSX_AGG_JIT_SYNC", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

sparse = "--sparse" in sys.argv    # keys spread over the INT range: the key box has too many cells, the directory is looked up
if sparse:
    sys.argv.remove("--sparse")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


val = torch.rand(n, device=dev, generator=g, dtype=torch.float64)
sides = [int(a) for a in sys.argv[2:]] or [20, 50, 70, 100]
for side in sides:
    k1 = torch.randint(0, side, (n,), device=dev, generator=g, dtype=torch.int32)
    k2 = torch.randint(0, side, (n,), device=dev, generator=g, dtype=torch.int32)
    if sparse:
        k1 = k1 * 1_000_003
        k2 = k2 * 7_919 - 11
    for strategy, label in ((T.AGG_COMPACT_KEY, "AOT shape"), (T.AGG_GENERIC, "run-time shape / interpreter")):
        aggs = [(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None), (T.AGG_AVG, T.col(2))] if strategy == T.AGG_COMPACT_KEY else \
            [(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None)]
        cfg = T.make_agg_config(strategy, [(T.INT, None), (T.INT, None), (T.DOUBLE, None)], keys=[0, 1], aggs=aggs,
                                est_groups=side * side)
        for directory in ("1", "0"):
            os.environ["QSX_AGG_DIRECTORY"] = directory
            st = capi.AggState(cfg)
            ms = timed(lambda: st.update([k1, k2, val], n))
            groups = st.num_groups()
            print(json.dumps({"keys": "sparse" if sparse else "small ranges", "groups": side * side, "found": groups - 1, "config": label, "directory": directory == "1",
                              "rows": n, "ms": round(ms, 3), "G_rows_per_s": round(n / ms / 1e6, 1),
                              "GBps_of_16B_rows": round(16 * n / ms / 1e6, 1)}))
            del st
