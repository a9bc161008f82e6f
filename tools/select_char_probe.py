#!/usr/bin/env python3
"""K1 on CHAR(n) stripes (qsx_select_cmp_char): ms and share of the HBM peak per width.  usage: select_char_probe.py [rows_millions]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

dev = torch.device("cuda", 0)
n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 25_000_000
g = torch.Generator(device=dev)
g.manual_seed(5)


def timed(fn, reps=10):
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


res = {"rows": n}
for width in (1, 2, 7, 10, 15, 16, 25):
    words = torch.zeros((5, width), dtype=torch.uint8, device=dev)
    for i, w in enumerate([b"AUTOMOBILE", b"BUILDING", b"FURNITURE", b"HOUSEHOLD", b"MACHINERY"]):
        raw = list(w[:width])
        words[i, :len(raw)] = torch.tensor(raw, dtype=torch.uint8, device=dev)
    col = words[torch.randint(0, 5, (n,), device=dev, generator=g)].contiguous()
    lit = b"BUILDING"[:width]
    for op, name in ((T.EQ, "eq"), (T.LT, "lt")):
        ms = timed(lambda: capi.select_cmp_char(col, op, lit))
        res[f"char{width}_{name}_ms"] = round(ms, 4)
        res[f"char{width}_{name}_frac"] = round((width + 0.125) * n / (ms * 1e-3) / 8e12, 3)
    bm, cnt = capi.select_cmp_char(col, T.EQ, lit)
    want = int((col == words[1]).all(dim=1).sum().item())
    res[f"char{width}_eq_count_ok"] = int(cnt.item()) == want
print(json.dumps(res))
