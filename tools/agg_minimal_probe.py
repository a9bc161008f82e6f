#!/usr/bin/env python3
"""BASELINE config 3's "minimal variant": two INT keys + one DOUBLE, SUM / COUNT / AVG, a handful of groups (the AOT plan shape
ShapeTwoIntKeysSumCountAvg over a 16-slot table), 600 M rows = 9.6 GB.  usage: agg_minimal_probe.py [rows_millions]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

dev = torch.device("cuda", 0)
n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 600_000_000
g = torch.Generator(device=dev)
g.manual_seed(9)
k1 = torch.randint(0, 3, (n,), device=dev, generator=g, dtype=torch.int32)
k2 = torch.randint(0, 3, (n,), device=dev, generator=g, dtype=torch.int32)
val = torch.rand(n, device=dev, generator=g, dtype=torch.float64)
cfg = T.make_agg_config(T.AGG_COMPACT_KEY, [(T.INT, None), (T.INT, None), (T.DOUBLE, None)], keys=[0, 1],
                        aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None), (T.AGG_AVG, T.col(2))], est_groups=9)
st = capi.AggState(cfg)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


ms = timed(lambda: (st.clear(), st.update([k1, k2, val], n)))
print(json.dumps({"rows": n, "ms": ms, "GBps": 16 * n / ms / 1e6, "groups": st.num_groups(),
                  "blocks_per_cu": os.environ.get("QSX_AGG_BLOCKS_PER_CU", "default")}))
