#!/usr/bin/env python3
"""K9 by partition count: 100 M rows of (INT key, DOUBLE value) scattered into P partitions (the reference's modulo partition
function on random keys).  usage: python tools/partition_by_p.py [rows]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
keys = torch.randint(0, 1 << 30, (n,), device=dev, generator=g, dtype=torch.int32)
val = torch.rand(n, device=dev, generator=g, dtype=torch.float64)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


line = {"rows": n, "bytes_per_row": 12}
for P in (8, 64, 256, 1024, 4096):
    try:
        line[f"P{P}_ms"] = round(timed(lambda: capi.partition_scatter(keys, P, [keys, val])), 3)
    except Exception as e:  # noqa: BLE001
        line[f"P{P}_ms"] = str(e)[:60]
print(json.dumps(line))
