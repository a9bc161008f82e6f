#!/usr/bin/env python3
"""K7 on the Q3 group-by shape: COLLISION_FREE state over a clustered dense key (4 rows per key), SUM(price * (1 - disc)),
200 M rows, through the run-time plan shape (QSX_AGG_NO_SPECIALIZE at creation is not needed: no AOT shape exists).  For the
QSX_EXP_* hooks through QSX_JIT_OPTIONS (tools/agg_dense_exp.sh).  usage: agg_dense_probe.py [rows_millions]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("QSX_AGG_JIT_MIN_ROWS", "0")
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

dev = torch.device("cuda", 0)
na = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 200_000_000
g = torch.Generator(device=dev)
g.manual_seed(5)


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


ne = na // 4
okey = (torch.arange(na, device=dev, dtype=torch.int64) // 4).to(torch.int32)
price = torch.rand(na, device=dev, generator=g, dtype=torch.float64) * 1e5
disc = torch.randint(0, 11, (na,), device=dev, generator=g).double() / 100
cfg = T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.DOUBLE, None), (T.DOUBLE, None)], keys=[0],
                        instrs=[(T.EX_SUB, 0, T.const(0), T.col(2)), (T.EX_MUL, 1, T.col(1), T.temp(0))], consts=[1.0],
                        aggs=[(T.AGG_SUM, T.temp(1))], num_entries=ne)
st = capi.AggState(cfg)
res = {"rows": na, "jit_options": os.environ.get("QSX_JIT_OPTIONS", "")}
res["clustered_ms"] = timed(lambda: st.update([okey, price, disc], na))
res["jit_state"] = capi.lib.qsx_debug_agg_jit_state(st._h, 0)
print(json.dumps(res))
