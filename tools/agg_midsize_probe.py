#!/usr/bin/env python3
"""Mid-size group count aggregation (2 INT keys, 10 k groups, 200 M rows): run under rocprofv3 --kernel-trace for the
split between key codes / K9 / per-piece aggregation."""
import os
import sys

import torch

os.environ.setdefault("QSX_AGG_JIT_SYNC", "1")   # time the run-time plan shape, not the interpreter that covers its compile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
g = torch.Generator(device=dev)
g.manual_seed(1)
k1 = torch.randint(0, 100, (n,), device=dev, generator=g, dtype=torch.int32)
k2 = torch.randint(0, 100, (n,), device=dev, generator=g, dtype=torch.int32)
val = torch.rand(n, device=dev, generator=g, dtype=torch.float64)
cfg = T.make_agg_config(T.AGG_COMPACT_KEY, [(T.INT, None), (T.INT, None), (T.DOUBLE, None)], keys=[0, 1],
                        aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None), (T.AGG_AVG, T.col(2))], est_groups=10_000)
st = capi.AggState(cfg)
for _ in range(2):
    st.update([k1, k2, val], n)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    st.update([k1, k2, val], n)
e1.record()
torch.cuda.synchronize()
print(f"update: {e0.elapsed_time(e1) / 3:.3f} ms")
