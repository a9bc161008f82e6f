#!/usr/bin/env python3
"""Hashed join build (keys without a dense domain): clear and build timed apart, by build size and by how full the table gets
(estimate = rows: load 0.8; estimate = 4 x rows: load 0.2).  usage: build_split_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for nb in (100_000, 300_000, 1_000_000, 3_000_000, 10_000_000):
    keys = (torch.randperm(nb, device=dev, generator=g, dtype=torch.int64) * 7 + 3).to(torch.int32)
    for slack in (1, 4):
        t = capi.JoinTable(T.INT, nb * slack)
        c = timed(lambda: t.clear())
        cb = timed(lambda: (t.clear(), t.build(keys)))
        print(json.dumps({"build_rows": nb, "estimate": nb * slack, "clear_ms": round(c, 4), "build_ms": round(cb - c, 4)}))
