#!/usr/bin/env python3
"""Dense probe of tables beyond one XCD's L2: plain kernel vs the XCD-sliced kernel (join_dense.hpp), 100 M probe keys,
pairs and count-only.  usage: python tools/probe_sliced.py [probe_rows]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


out = (torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev),
       torch.zeros(1, dtype=torch.int64, device=dev))
for nb in (1_000_000, 2_000_000, 4_000_000, 8_000_000, 16_000_000, 64_000_000):
    b = torch.randperm(nb, device=dev, generator=g, dtype=torch.int32)
    p = torch.randint(0, nb, (n,), device=dev, generator=g, dtype=torch.int32)
    t = capi.JoinTable(T.INT, nb, key_range=(0, nb - 1))
    t.build(b)
    line = {"build_keys": nb, "head_MiB": nb * 4 / 2**20, "probe_rows": n}
    for sliced in ("0", "1"):
        os.environ["QSX_JOIN_SLICED"] = sliced
        line["pairs_ms_sliced" if sliced == "1" else "pairs_ms_plain"] = round(timed(lambda: t.probe(p, capacity=n, out=out)), 3)
        line["count_ms_sliced" if sliced == "1" else "count_ms_plain"] = round(timed(lambda: t.probe_count(p)), 3)
    print(json.dumps(line), flush=True)
    t.close()
