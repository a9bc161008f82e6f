#!/usr/bin/env python3
"""Times qsx_agg_update for a ladder of configurations (keys only -> full Q1) to see where the
aggregation kernel's time goes.  usage: python tools/agg_probe.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402
from bench import gen_q1_columns_gpu  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
dev = torch.device("cuda:0")
cols = gen_q1_columns_gpu(n, dev, 4)
layout = [(T.CHAR, 1), (T.CHAR, 1), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None)]
q1_instrs = [(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0)),
             (T.EX_ADD, 2, T.const(0), T.col(5)), (T.EX_MUL, 3, T.temp(1), T.temp(2))]
ladder = {
    "A keys + COUNT": dict(aggs=[(T.AGG_COUNT_STAR, None)]),
    "B + SUM(qty)": dict(aggs=[(T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(2))]),
    "C 5 plain SUMs": dict(aggs=[(T.AGG_COUNT_STAR, None)] + [(T.AGG_SUM, T.col(c)) for c in (2, 3, 4, 5)] + [(T.AGG_AVG, T.col(2))]),
    "D 4 SUMs + 1 expr SUM": dict(instrs=q1_instrs[:2], consts=[1.0],
                                  aggs=[(T.AGG_COUNT_STAR, None)] + [(T.AGG_SUM, T.col(c)) for c in (2, 3, 4)] + [(T.AGG_SUM, T.temp(1))]),
    "E full Q1": dict(instrs=q1_instrs, consts=[1.0],
                      aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.col(3)), (T.AGG_SUM, T.temp(1)), (T.AGG_SUM, T.temp(3)),
                            (T.AGG_AVG, T.col(2)), (T.AGG_AVG, T.col(3)), (T.AGG_AVG, T.col(4)), (T.AGG_COUNT_STAR, None)]),
    "F no keys, SUM(qty)": dict(strategy=T.AGG_SINGLE_STATE, keys=[], aggs=[(T.AGG_SUM, T.col(2))]),
}
for name, kw in ladder.items():
    strategy = kw.pop("strategy", T.AGG_COMPACT_KEY)
    keys = kw.pop("keys", [0, 1])
    cfg = T.make_agg_config(strategy, layout, keys=keys, est_groups=6, **kw)
    st = capi.AggState(cfg)
    st.update(cols, n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        st.update(cols, n)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{name:28s} {ms:8.3f} ms  {n / ms / 1e6:8.1f} G rows/s")
