#!/usr/bin/env python3
"""TPC-H Q1 composed from the C-ABI entry points on one GPU (synthetic lineitem of SF = argv[1], default 100 -> 600 M rows):
l_shipdate <= DATE (98 % of the rows) -> GROUP BY l_returnflag, l_linestatus with Q1's eight aggregates -> ORDER BY the keys.
Two plans: the predicate inside the aggregation state (AggregationOperationState's own predicate, one pass) and a K1
bitmap handed to the aggregation as its filter (SelectOperator in front)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("QSX_AGG_JIT_MIN_ROWS", "0")
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
SF = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
n = int(6_000_000 * SF)
cols = bench.gen_q1_columns_gpu(n, dev, 4)
g = torch.Generator(device=dev)
g.manual_seed(11)
# l_shipdate as the reference stores it: DATE = 8-byte DateLit (year | month << 32 | day << 40), uniform over 1992-01-01 .. 1998-12-28;
# `l_shipdate <= DATE '1998-09-02'` keeps ~95 % of the rows.  argv[2] = "int": the 4-byte integer stand-in of round 1.
AS_INT = len(sys.argv) > 2 and sys.argv[2] == "int"
if AS_INT:
    shipdate = torch.randint(19920101, 19981201, (n,), device=dev, generator=g, dtype=torch.int32)
    DATE, DATE_TYPE = 19980902, T.INT
else:
    shipdate = (torch.randint(1992, 1999, (n,), device=dev, generator=g, dtype=torch.int64) |
                (torch.randint(1, 13, (n,), device=dev, generator=g, dtype=torch.int64) << 32) |
                (torch.randint(1, 29, (n,), device=dev, generator=g, dtype=torch.int64) << 40))
    DATE, DATE_TYPE = T.date_raw(1998, 9, 2), T.DATE

q1 = bench.q1_config()
with_pred = T.make_agg_config(
    T.AGG_COMPACT_KEY,
    columns=[(T.CHAR, 1), (T.CHAR, 1), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None), (DATE_TYPE, None)],
    keys=[0, 1],
    instrs=[(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0)),
            (T.EX_ADD, 2, T.const(0), T.col(5)), (T.EX_MUL, 3, T.temp(1), T.temp(2))],
    consts=[1.0],
    aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.col(3)), (T.AGG_SUM, T.temp(1)), (T.AGG_SUM, T.temp(3)),
          (T.AGG_AVG, T.col(2)), (T.AGG_AVG, T.col(3)), (T.AGG_AVG, T.col(4)), (T.AGG_COUNT_STAR, None)],
    pred=[(6, T.LE, DATE)], est_groups=6)
s_pred, s_filter = capi.AggState(with_pred), capi.AggState(q1)


def plan_state_predicate():
    s_pred.clear()
    s_pred.update(cols + [shipdate], n)
    keys, vals, _, groups = s_pred.finalize(dev, capacity=16)
    k = int(groups.item())
    order = capi.sort_permutation([keys[0][:k], keys[1][:k]])
    return [capi.gather(v[:k], order) for v in vals], k


def plan_select_then_aggregate():
    bm, _ = capi.select_cmp(shipdate, T.LE, DATE, qtype=DATE_TYPE)
    s_filter.clear()
    s_filter.update(cols, n, filter_bitmap=bm)
    keys, vals, _, groups = s_filter.finalize(dev, capacity=16)
    k = int(groups.item())
    order = capi.sort_permutation([keys[0][:k], keys[1][:k]])
    return [capi.gather(v[:k], order) for v in vals], k


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


ms_a, (va, ka) = timed(plan_state_predicate)
ms_b, (vb, kb) = timed(plan_select_then_aggregate)
same = ka == kb and all(torch.allclose(x.double(), y.double(), rtol=1e-9) for x, y in zip(va, vb))
print(json.dumps({"query": "TPC-H Q1 (synthetic, 1 GPU), l_shipdate " + ("INT stand-in" if AS_INT else "DATE (8-byte DateLit)"), "SF": SF, "lineitem": n, "groups": ka,
                  "predicate in the aggregation state: ms": ms_a, "rows_per_s": n / ms_a * 1e3,
                  "select (K1) then aggregate under the bitmap: ms": ms_b, "rows_per_s (select + aggregate)": n / ms_b * 1e3,
                  "plans agree": bool(same), "count_order": [int(c) for c in va[7].tolist()]}))
