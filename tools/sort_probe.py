#!/usr/bin/env python3
"""Times qsx_sort_permutation / qsx_sort_top_k / qsx_distinct_rows.  usage: sort_probe.py [rows_millions]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402

dev = torch.device("cuda", 0)
n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 100_000_000
g = torch.Generator(device=dev)
g.manual_seed(9)
k32 = torch.randint(-2**31, 2**31 - 1, (n,), device=dev, generator=g, dtype=torch.int32)
f64 = torch.rand(n, device=dev, generator=g, dtype=torch.float64)
small = torch.randint(0, 1000, (n,), device=dev, generator=g, dtype=torch.int32)


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


res = {"rows": n}
res["sort INT key"] = timed(lambda: capi.sort_permutation([k32]))
res["sort DOUBLE key"] = timed(lambda: capi.sort_permutation([f64]))
res["sort (INT, DOUBLE) DESC/ASC"] = timed(lambda: capi.sort_permutation([small, f64], [True, False]))
res["top 10 of DOUBLE"] = timed(lambda: capi.sort_top_k([f64], 10, [True]))
res["distinct (INT 1000 values, INT 1000 values)"] = timed(lambda: capi.distinct_rows([small, (k32 & 1023)]))
res["torch.sort INT (reference point)"] = timed(lambda: torch.sort(k32))
print(json.dumps(res))
