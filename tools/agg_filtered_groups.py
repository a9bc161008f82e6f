#!/usr/bin/env python3
"""A filter bitmap in front of a group-by with many groups (a predicate's or LIP filter's TupleIdSequence): the survivors compacted
for the partition passes (aggregate.hip update_filtered_end_to_end) against the tile kernels under the filter
(QSX_AGG_FILTER_COMPACT=0).  usage: python tools/agg_filtered_groups.py   (run it once per setting: the switch is read once)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(5)
n, groups = 100_000_000, 1_000_000
keys = torch.randint(0, groups, (n,), device=dev, generator=g, dtype=torch.int32)
val = torch.rand(n, device=dev, generator=g, dtype=torch.float64)
cfg = T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.DOUBLE, None)], keys=[0], aggs=[(T.AGG_SUM, T.col(1)), (T.AGG_COUNT_STAR, None)], est_groups=groups)
for sigma in (0.9, 0.5, 0.1, 0.01):
    keep = torch.rand(n, device=dev, generator=g) < sigma
    words = torch.zeros((n + 63) // 64 * 64, dtype=torch.bool, device=dev)
    words[:n] = keep
    # MSB-first words (storage/TupleIdSequence): bit i of the sequence = bit 63 - (i & 63) of word i >> 6
    w = (words.view(-1, 64).to(torch.int64) << torch.arange(63, -1, -1, device=dev, dtype=torch.int64)).sum(dim=1)
    st = capi.AggState(cfg)

    def run():
        st.clear()
        st.update([keys, val], n, filter_bitmap=w)
    run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run()
    b.record()
    torch.cuda.synchronize()
    print(json.dumps({"rows": n, "groups": groups, "selectivity": sigma, "filter_compact": os.environ.get("QSX_AGG_FILTER_COMPACT", "1"),
                      "ms": round(a.elapsed_time(b) / 5, 3)}), flush=True)
    del st, keep, words, w
