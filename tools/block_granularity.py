#!/usr/bin/env python3
"""What work-order granularity costs: the Q1 aggregation over N rows issued as one qsx_agg_update per block of B rows
(the reference issues one AggregationWorkOrder per 4 MB storage block, relational_operators/AggregationOperator.cpp:38-79;
~120 K Q1 rows), and K1 + the dense probe the same way.  usage: python tools/block_granularity.py [rows]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120_000_000
dev = torch.device("cuda:0")
cols = bench.gen_q1_columns_gpu(n, dev, 4)
keys = torch.randint(0, 1_000_000, (n,), device=dev, dtype=torch.int32)
table = capi.JoinTable(T.INT, 1_000_000, key_range=(0, 999_999))
table.build(torch.randperm(1_000_000, device=dev, dtype=torch.int32))
out = (torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))
st = capi.AggState(bench.q1_config())


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for block in (n, 10_000_000, 1_000_000, 120_000):
    def agg():
        st.clear()
        for s in range(0, n, block):
            e = min(n, s + block)
            st.update([c[s:e] for c in cols], e - s)

    def probe():
        for s in range(0, n, block):
            e = min(n, s + block)
            table.probe(keys[s:e], capacity=e - s, out=(out[0][s:e], out[1][s:e], out[2]), probe_base_tid=s)

    def select():
        for s in range(0, n, block):
            e = min(n, s + block)
            capi.select_cmp(cols[2][s:e], T.LT, 24.0)

    def agg_run():                       # the same blocks as ONE work order: qsx_agg_update_blocks
        st.clear()
        st.update_blocks(run_blocks)

    def select_run():                    # qsx_select_cmp_blocks
        capi.select_cmp_blocks(select_blocks, T.LT, 24.0, out_bitmaps=select_outs)

    def probe_run():                     # qsx_join_probe_blocks
        table.probe_blocks(key_blocks, out=out)

    key_blocks = [keys[s0:min(n, s0 + block)] for s0 in range(0, n, block)]
    select_blocks = [cols[2][s0:min(n, s0 + block)] for s0 in range(0, n, block)]
    select_outs = [capi.new_bitmap(c.numel(), dev) for c in select_blocks]
    run_blocks = [[c[s0:min(n, s0 + block)] for c in cols] for s0 in range(0, n, block)]
    print(json.dumps({"rows": n, "block_rows": block, "calls": (n + block - 1) // block, "aggregate_ms": round(timed(agg), 3),
                      "aggregate_as_one_run_of_blocks_ms": round(timed(agg_run), 3),
                      "probe_ms": round(timed(probe), 3), "probe_as_one_run_of_blocks_ms": round(timed(probe_run), 3), "select_ms": round(timed(select), 3),
                      "select_as_one_run_of_blocks_ms": round(timed(select_run), 3)}), flush=True)
